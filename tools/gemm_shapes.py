"""Lists the library GEMMs of one mdn_gate training step with their shapes and device times:
python tools/gemm_shapes.py [B] [N]"""
import sys
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import bench                    # noqa: E402
import mdnvae_amd as M          # noqa: E402
from mdnvae_amd import ops      # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
N = int(sys.argv[2]) if len(sys.argv) > 2 else 87
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = M.MDNGateVAE(R=5, n_nodes=N, k=32, n_layers=3, latent_dim=68, n_traits=4, symmetric_input=True).to(dev)
trainer = M.Trainer(model, "mdn")
x = torch.rand(B, N, N, device=dev)
x = (x + x.transpose(1, 2)).reshape(B, N * N).contiguous()
adjacency = [[(i + d) % N for d in range(1, 33)] for i in range(N)]
traits = torch.randn(B, 4, device=dev)
eps = torch.randn(B, 68, device=dev)
for _ in range(3):
    trainer.step((x, adjacency, traits), x, eps=eps)
log = []
inner = ops._library_gemm
inner_fused = ops._library_fused_linear


def timed(a, b):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = inner(a, b)
    e1.record()
    log.append((tuple(a.shape), a.stride(), tuple(b.shape), b.stride(), e0, e1))
    return out


ops._library_gemm = timed


def timed_fused(a, b, bias, relu):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = inner_fused(a, b, bias, relu)
    e1.record()
    log.append((tuple(a.shape), a.stride(), tuple(b.shape), b.stride(), e0, e1))
    return out


ops._library_fused_linear = timed_fused
trainer.step((x, adjacency, traits), x, eps=eps)
torch.cuda.synchronize()
for sa, sta, sb, stb, e0, e1 in log:
    us = e0.elapsed_time(e1) * 1e3
    print(f"{sa} {sta} x {sb} {stb}: {us:8.1f} us  {2.0 * sa[0] * sa[1] * sb[1] / us / 1e6:7.1f} TFLOP/s")
