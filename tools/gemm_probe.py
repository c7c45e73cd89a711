"""Times the library fp32 GEMM layouts for the FNN tower shapes (which layout rocBLAS/hipBLASLt runs fastest)."""
import os, torch, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
if os.environ.get("BLAS_LIB"):
    torch.backends.cuda.preferred_blas_library(os.environ["BLAS_LIB"])
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
D = int(sys.argv[2]) if len(sys.argv) > 2 else 7569
H = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
x = torch.randn(B, D, device="cuda"); w = torch.randn(H, D, device="cuda") * 0.01; wt = w.t().contiguous()
g = torch.randn(B, H, device="cuda"); xt = x.t().contiguous()
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
fl = 2.0 * B * D * H
for name, fn in [("fwd NT x@w.t()", lambda: torch.matmul(x, w.t())), ("fwd NN x@wt", lambda: torch.matmul(x, wt)),
                 ("fwd via (w@xt).t", lambda: torch.matmul(w, xt)),
                 ("wgrad TN g.t()@x", lambda: torch.matmul(g.t(), x)), ("wgrad as (x.t()@g)", lambda: torch.matmul(x.t(), g)),
                 ("wgrad NN xt@g", lambda: torch.matmul(xt, g)),
                 ("hand wgrad g^T.x", (lambda: __import__("mdnvae_amd").ops._linear_wgrad(g, x)) if (D % 4 == 0 and H % 4 == 0) else (lambda: None)),
                 ("dgrad NN g@w", lambda: torch.matmul(g, w)), ("dgrad via (wt@g.t()).t", lambda: torch.matmul(wt, g.t()))]:
    ms = timeit(fn)
    print(f"{name:24s} {ms:8.3f} ms  {fl / ms / 1e9:7.1f} TFLOP/s")
