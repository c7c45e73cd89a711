"""Times the first-layer forward product (B x 3840 . 3840 x 2048) as six bf16 products and as three fp16
products, with their operand splits.  usage: python tools/sg16_probe.py [B]"""
import sys
import torch
sys.path.insert(0, ".")
import mdnvae_amd as M  # noqa: E402,F401
from mdnvae_amd import _lib  # noqa: E402
from mdnvae_amd._lib import call, ptr, stream  # noqa: E402

import os  # noqa: E402
if os.environ.get("SG_LIB"):                       # a tuning build of the library
    _lib.LIB_PATH = os.path.abspath(os.environ["SG_LIB"])
DEV = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
T, N = 3840, 2048
x = torch.rand(B, T, device=DEV) * 5
w = torch.randn(N, T, device=DEV) * 0.02
b1 = torch.randn(N // 2, device=DEV)
y1 = torch.empty(B, N // 2, device=DEV)
y2 = torch.empty(B, N // 2, device=DEV)
op = lambda r, k: torch.empty(int(_lib.load().mdnvae_split_operand_bytes(r, k)), dtype=torch.uint8, device=DEV)
xa, wb = op(B, T), op(N, T)
mx, mw = (torch.zeros(64, dtype=torch.int32, device=DEV) for _ in range(2))


def timed(f, n=10):
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


steps = {
    "split_rows x (bf16 x3)": lambda: call("mdnvae_split_rows", ptr(x), B, T, T, 0, xa.data_ptr(), stream()),
    "split_rows w (bf16 x3)": lambda: call("mdnvae_split_rows", ptr(w), N, T, T, 0, wb.data_ptr(), stream()),
    "gemm 6 bf16 terms": lambda: call("mdnvae_split_gemm", xa.data_ptr(), wb.data_ptr(), B, N, T, B, N // 2, ptr(b1), ptr(b1), 1, 0,
                                      ptr(y1), ptr(y2), N // 2, N // 2, 256, 6, stream()),
    "gemm 3 bf16 terms": lambda: call("mdnvae_split_gemm", xa.data_ptr(), wb.data_ptr(), B, N, T, B, N // 2, ptr(b1), ptr(b1), 1, 0,
                                      ptr(y1), ptr(y2), N // 2, N // 2, 256, 3, stream()),
    "absmax x": lambda: call("mdnvae_absmax", ptr(x), B, T, T, 0.0, 1, mx.data_ptr(), stream()),
    "absmax w": lambda: call("mdnvae_absmax", ptr(w), N, T, T, 0.0, 1, mw.data_ptr(), stream()),
    "split_rows x (fp16 x2)": lambda: call("mdnvae_split_rows_f16", ptr(x), B, T, T, 0, mx.data_ptr(), xa.data_ptr(), stream()),
    "split_rows w (fp16 x2)": lambda: call("mdnvae_split_rows_f16", ptr(w), N, T, T, 0, mw.data_ptr(), wb.data_ptr(), stream()),
    "gemm 3 fp16 terms": lambda: call("mdnvae_split_gemm_f16", xa.data_ptr(), wb.data_ptr(), mx.data_ptr(), mw.data_ptr(), B, N, T, B,
                                      N // 2, ptr(b1), ptr(b1), 1, 0, ptr(y1), ptr(y2), N // 2, N // 2, 256, stream()),
    "gemm 3 fp16 terms, 128-row tiles": lambda: call("mdnvae_split_gemm_f16", xa.data_ptr(), wb.data_ptr(), mx.data_ptr(), mw.data_ptr(),
                                                     B, N, T, B, N // 2, ptr(b1), ptr(b1), 1, 0, ptr(y1), ptr(y2), N // 2, N // 2, 128, stream()),
}
for name, f in steps.items():
    print(f"{name:36s} {timed(f):.4f} ms", flush=True)
want = torch.relu(x[:256].double() @ w.double().t() + torch.cat([b1, b1]).double())
got = torch.cat([y1[:256], y2[:256]], 1).double()
print("fp16 product, max error / max entry:", float((got - want).abs().max() / want.abs().max()))
