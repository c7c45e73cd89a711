"""Times mdnvae_tri_pack_absmax / mdnvae_tri_fold at the bench shapes.  usage: [SG_LIB=...] python tools/tri_probe.py"""
import os
import sys
import torch
sys.path.insert(0, ".")
import mdnvae_amd as M  # noqa: E402,F401
from mdnvae_amd import _lib  # noqa: E402
if os.environ.get("SG_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["SG_LIB"])
from mdnvae_amd._lib import call, ptr, stream  # noqa: E402

DEV = "cuda:0"
B, N, T = 16384, 87, 3840
x = torch.rand(B, N * N, device=DEV)
x = ((x.view(B, N, N) + x.view(B, N, N).transpose(1, 2)) * 0.5).reshape(B, N * N).contiguous()
w = torch.randn(1024, N * N, device=DEV)
xt = torch.empty(B, T, device=DEV)
wt = torch.empty(1024, T, device=DEV)
asym = torch.zeros(1, dtype=torch.int32, device=DEV)
mx = torch.zeros(64, dtype=torch.int32, device=DEV)


def timed(f, n=20):
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


print("pack %.4f ms" % timed(lambda: call("mdnvae_tri_pack", ptr(x), B, N, T, ptr(xt), asym.data_ptr(), stream())))
print("pack+absmax %.4f ms" % timed(lambda: call("mdnvae_tri_pack_absmax", ptr(x), B, N, T, ptr(xt), asym.data_ptr(), 1, mx.data_ptr(), stream())))
print("pack+absmax, no clear %.4f ms" % timed(lambda: call("mdnvae_tri_pack_absmax", ptr(x), B, N, T, ptr(xt), asym.data_ptr(), 0, mx.data_ptr(), stream())))
print("fold %.4f ms" % timed(lambda: call("mdnvae_tri_fold", ptr(w), 1024, N, T, ptr(wt), stream())))
assert int(asym.item()) == 0
