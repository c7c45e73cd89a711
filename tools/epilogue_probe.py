"""Library GEMM with bias / ReLU epilogue against GEMM + the bias_act kernel, first-layer shape."""
import sys
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import mdnvae_amd as M          # noqa: E402
from mdnvae_amd import _lib     # noqa: E402

dev = "cuda:0"
for (B, K, H) in ((16384, 3840, 1024), (16384, 1024, 128), (16384, 128, 68)):
    x = torch.randn(B, K, device=dev)
    w = torch.randn(H, K, device=dev) * 0.02
    b = torch.randn(H, device=dev)

    def timeit(fn, iters=10):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e3

    def plain():
        y = torch.mm(x, w.t())
        _lib.call("mdnvae_bias_act", _lib.ptr(y), _lib.ptr(b), B, H, 3, _lib.stream())
        return y

    def fused():
        return torch._addmm_activation(b, x, w.t(), use_gelu=False)

    def addmm():
        return torch.addmm(b, x, w.t())

    ref = plain()
    got = fused()
    print(f"[{B}x{K}]@[{K}x{H}]: mm+bias_act {timeit(plain):8.1f} us, _addmm_activation {timeit(fused):8.1f} us, "
          f"addmm (bias only) {timeit(addmm):8.1f} us, max diff {float((ref - got).abs().max()):.3e}")
