"""Times the split-over-the-batch weight gradient for the layer shapes of the FNN towers:
python tools/ts_probe.py [library.so ...]"""
import subprocess
import sys
import torch
libs = sys.argv[1:]
if len(libs) > 1:
    for lib in libs:
        subprocess.run([sys.executable, __file__, lib], check=True)
    sys.exit(0)
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from mdnvae_amd import _lib   # noqa: E402
if libs:
    _lib.LIB_PATH = libs[0]
from mdnvae_amd import ops    # noqa: E402
out = []
for (B, N, M) in ((16384, 128, 1024), (16384, 68, 128), (256, 128, 1024), (256, 68, 128)):
    g, x = torch.randn(B, N, device="cuda"), torch.randn(B, M, device="cuda")
    for _ in range(3):
        ops._linear_wgrad(g, x)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops._linear_wgrad(g, x)
    e1.record()
    torch.cuda.synchronize()
    out.append(f"[{B},{N},{M}] {e0.elapsed_time(e1) / 20 * 1e3:6.1f} us")
print(libs[0] if libs else "in-tree", "  ".join(out))
