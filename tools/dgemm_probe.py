"""Time the float64 matrix-core product against the library's at the validation size."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import mdnvae_amd as M
from mdnvae_amd import fid
d = int(sys.argv[1]) if len(sys.argv) > 1 else 4624
g = torch.Generator(device="cuda").manual_seed(1)
q = torch.randn(d, d, dtype=torch.float64, device="cuda", generator=g)
b = (q + q.T) / (2 * d ** 0.5)
a = b @ b
out = torch.empty_like(a)


def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


fl = 2.0 * d ** 3
t = timed(lambda: torch.matmul(a, b, out=out)); print(f"library dgemm      {t:7.2f} ms  {fl/t/1e9:6.1f} TFLOP/s")
t = timed(lambda: fid._dgemm(a, b, out, False)); print(f"own, all tiles     {t:7.2f} ms  {fl/t/1e9:6.1f} TFLOP/s")
t = timed(lambda: fid._dgemm(a, b, out, True)); print(f"own, upper tiles   {t:7.2f} ms  {fl/t/1e9:6.1f} TFLOP/s of the full product's flops")
