"""Per-kernel register / spill / occupancy table of one .hip source (hipcc -Rpass-analysis=kernel-resource-usage)."""
import re
import subprocess
import sys

src = sys.argv[1]
out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-I/root/repo/include",
                      "-Rpass-analysis=kernel-resource-usage", "--cuda-device-only", "-S", src, "-o", "/tmp/kres.s"],
                     capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = subprocess.run(["c++filt", t.split(":", 1)[1].strip()], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r"\(.*", "", cur)
        rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1)
        rows[cur][k.strip()] = v.strip()
for k, v in rows.items():
    print(f"{k[:90]:90s} vgpr {v.get('VGPRs','?'):>4} agpr {v.get('AGPRs','?'):>4} sgpr {v.get('TotalSGPRs','?'):>4} "
          f"vspill {v.get('VGPRs Spill','?'):>3} sspill {v.get('SGPRs Spill','?'):>4} occ {v.get('Occupancy [waves/SIMD]','?')} scratch {v.get('ScratchSize [bytes/lane]','?')}")
