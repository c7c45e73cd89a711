"""Time the instance-norm passes of the residual level (192 channels, n = 87) at B = 1024; optional scale argument =
experiment switch of a development build."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import mdnvae_amd as M
from mdnvae_amd import ops
DEV = "cuda:0"
B, C = 1024, 192
G = ops._CBGeom(87)
H = G.h3
Qy = ops._ru(H * G.pr, 8)
y16 = torch.randn(B, C, Qy, device=DEV).half()
st = torch.empty(B * C, 2, device=DEV)
Qg = ops._ru((H + 2) * G.pr, 8)
g = torch.randn(B, C, Qg, device=DEV)
base = torch.randn(B, C, H, H, device=DEV)
skip = torch.randn(B, C, H, H, device=DEV)


def timed(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


fwd_bytes = B * C * (Qy * 2 + G.pp_r * 2)
t = timed(lambda: ops._cb_norm_fwd(y16, None, H, H, G.pr, True, 0, G.pp_r))
print(f"norm fwd (fp16 in, bf16 out)            {t*1e3:7.1f} us  {fwd_bytes/t/1e9:6.2f} TB/s")
for sc in (2.0, 3.0, 4.0):
    t = timed(lambda: ops._cb_norm_fwd(y16, None, H, H, G.pr, True, 0, G.pp_r, scale=sc))
    print(f"   experiment {sc}: {t*1e3:7.1f} us")
fb = fwd_bytes + B * C * H * H * 8
t = timed(lambda: ops._cb_norm_fwd(y16, None, H, H, G.pr, False, 0, G.pp_r, skip=skip, want_stream=True))
print(f"norm fwd + skip + stream                {t*1e3:7.1f} us  {fb/t/1e9:6.2f} TB/s")
_, _, _, st = ops._cb_norm_fwd(y16, None, H, H, G.pr, True, 0, G.pp_r)
bb = B * C * (Qy * 2 + Qg * 4 + G.pp_dr * 2)
t = timed(lambda: ops._cb_norm_bwd(g, 1, G.pr, y16, st, G.pr, True, C, H, H, 0, G.pp_dr))
print(f"norm bwd (fold)                         {t*1e3:7.1f} us  {bb/t/1e9:6.2f} TB/s")
t = timed(lambda: ops._cb_norm_bwd(g, 1, G.pr, y16, st, G.pr, False, C, H, H, 0, G.pp_dr, base=base, want_sum=True))
print(f"norm bwd (fold) + base + sum            {t*1e3:7.1f} us  {(bb + B*C*H*H*8)/t/1e9:6.2f} TB/s")
