"""Times the decoder panel weight gradient (20 panels): python tools/pw_probe.py [library.so ...]"""
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
out = []
RL = 20
for B in (16384, 4096, 256):
    hin, dp = torch.randn(RL, B, 96, device="cuda"), torch.randn(RL, B, 96, device="cuda")
    dwt = torch.empty(RL, 96, 96, device="cuda")
    ws = torch.empty(_lib.workspace_bytes_f("mdnvae_decoder_wgrad_workspace_floats", B, RL), device="cuda")
    run = lambda: _lib.call("mdnvae_decoder_wgrad", _lib.ptr(hin), _lib.ptr(dp), B, RL, _lib.ptr(ws), _lib.ptr(dwt), _lib.stream())
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    out.append(f"B={B} {e0.elapsed_time(e1) / 20 * 1e3:6.1f} us")
print(libs[0] if libs else "in-tree", "  ".join(out))
