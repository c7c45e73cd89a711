"""Library bf16 GEMM with fp32 output at the shapes of a 3-way split first layer (6 products stacked along K)."""
import torch
dev = "cuda:0"
B, T, H = 16384, 3840, 1024


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


for terms in (1, 3, 6):
    K = T * terms
    a = torch.randn(B, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(H, K, device=dev, dtype=torch.bfloat16)
    try:
        us = timeit(lambda: torch.mm(a, w.t(), out_dtype=torch.float32))
        print(f"fwd  [{B}x{K}]@[{K}x{H}] bf16->f32: {us:8.1f} us {2.0 * B * K * H / us / 1e6:8.1f} TFLOP/s")
    except Exception as e:      # noqa: BLE001
        print("out_dtype fwd failed:", type(e).__name__, str(e)[:200])
        us = timeit(lambda: torch.mm(a, w.t()))
        print(f"fwd  [{B}x{K}]@[{K}x{H}] bf16->bf16: {us:8.1f} us {2.0 * B * K * H / us / 1e6:8.1f} TFLOP/s")
    # weight gradient: [T x (B*terms)] @ [(B*terms) x H]
    xa = torch.randn(B * terms, T, device=dev, dtype=torch.bfloat16)
    g = torch.randn(B * terms, H, device=dev, dtype=torch.bfloat16)
    try:
        us = timeit(lambda: torch.mm(xa.t(), g, out_dtype=torch.float32))
        print(f"wgrad [{T}x{B * terms}]@[{B * terms}x{H}] bf16->f32: {us:8.1f} us {2.0 * B * terms * T * H / us / 1e6:8.1f} TFLOP/s")
    except Exception as e:      # noqa: BLE001
        print("out_dtype wgrad failed:", type(e).__name__, str(e)[:200])
