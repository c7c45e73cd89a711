"""Library float64 GEMM rate at the Newton-Schulz size (what bounds fid_from_stats at large d)."""
import sys, torch
d = int(sys.argv[1]) if len(sys.argv) > 1 else 4624
a = torch.randn(d, d, device="cuda", dtype=torch.float64); b = torch.randn(d, d, device="cuda", dtype=torch.float64)
for _ in range(2): torch.mm(a, b)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): torch.mm(a, b)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print(f"dgemm d={d}: {ms:.2f} ms  {2.0 * d ** 3 / ms / 1e9:.1f} TFLOP/s")
a32, b32 = a.float(), b.float()
for _ in range(2): torch.mm(a32, b32)
e0.record()
for _ in range(5): torch.mm(a32, b32)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print(f"sgemm d={d}: {ms:.2f} ms  {2.0 * d ** 3 / ms / 1e9:.1f} TFLOP/s")
