"""Times the rank-one likelihood pass through the C ABI (general and symmetric-target forms):
python tools/outer_probe.py [B] [N] [library.so ...]   (default: the in-tree library)"""
import subprocess
import sys
import torch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
N = int(sys.argv[2]) if len(sys.argv) > 2 else 87
libs = sys.argv[3:]
if len(libs) > 1:                       # one process per library: the binding loads once
    for lib in libs:
        subprocess.run([sys.executable, __file__, str(B), str(N), lib], check=True)
    sys.exit(0)
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from mdnvae_amd import _lib   # noqa: E402
if libs:
    _lib.LIB_PATH = libs[0]
import mdnvae_amd as M        # noqa: E402

K, L, dev = 5, 68, "cuda:0"
torch.manual_seed(0)
nodes = torch.zeros(B, K, M.ops.DEC_NP, device=dev)
nodes[:, :, :N] = torch.rand(B, K, N, device=dev)
x = torch.rand(B, N, N, device=dev)
x = (x + x.transpose(1, 2)).reshape(B, N * N).contiguous()
logits, ls = torch.randn(B, K, device=dev), torch.full((K,), 1.5, device=dev)
mu, lv = torch.randn(B, L, device=dev) * 0.1, torch.randn(B, L, device=dev) * 0.1
out = torch.empty(3, device=dev)
resp, glog, gls = (torch.empty(B, K, device=dev) for _ in range(3))
gn, gmu, glv = torch.empty_like(nodes), torch.empty_like(mu), torch.empty_like(mu)
ws = torch.empty(_lib.load().mdnvae_mdn_workspace_bytes(B), dtype=torch.uint8, device=dev)
p = _lib.ptr


def run(sym):
    _lib.call("mdnvae_mdn_outer_fwdbwd", p(nodes), p(x), p(logits), p(ls), B, K, N, p(mu), p(lv), L, p(gn), p(glog),
              p(gls), p(gmu), p(glv), p(resp), sym, ws.data_ptr(), p(out), _lib.stream())


def timeit(fn, iters=30):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


bytes_x = 4.0 * B * N * N
res = []
for sym in (0, 1):
    us = timeit(lambda: run(sym))
    res.append(f"sym={sym}: {us:.1f} us ({bytes_x / us / 1e3:.0f} GB/s of x)")
print((libs[0] if libs else "in-tree"), f"B={B} N={N}", "; ".join(res), "(each with its finalize launch)")
for mode in ("poisson_log", "mse"):
    res = []
    for sym in (False, True):
        us = timeit(lambda: M.ops.outer_sum_loss_and_grads(nodes, x, mu, lv, mode, N, symmetric_x=sym))
        res.append(f"sym={int(sym)}: {us:.1f} us")
    print(f"outer_sum {mode}", "; ".join(res), "(with allocations and the finalize launch)")
