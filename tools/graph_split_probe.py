"""Eager step vs captured replay at a batch that takes the split first-layer product (single process)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mdnvae_amd as M
from mdnvae_amd import ops

DEV = "cuda:0"
if os.environ.get("ONE_RANK"):
    import torch.distributed as dist
    from mdnvae_amd import trainer as TR
    dist.init_process_group(os.environ["ONE_RANK"], init_method="tcp://127.0.0.1:29617", world_size=1, rank=0,
                            **({"device_id": torch.device(DEV)} if os.environ["ONE_RANK"] == "nccl" else {}))
    TR.MIN_RANKS_FOR_EXCHANGE = 1
n = 68
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
OFF = int(sys.argv[2]) if len(sys.argv) > 2 else 0
x = M.data.synthetic_sc(OFF + B, n, seed=77, device=DEV)[OFF:]
g = torch.Generator().manual_seed(9)
eps = torch.randn(3, OFF + B, 68, generator=g).to(DEV)[:, OFF:]
traits = torch.randn(OFF + B, 4, generator=g).to(DEV)[OFF:]
print("absmax of the three batches", [float((x * (1.0 + 0.05 * s)).abs().max()) for s in range(3)])
adjacency = [[(i + d) % n for d in range(32)] for i in range(n)]


def run(graph, **arith):
    torch.manual_seed(5)
    model = M.MDNGateVAE(R=5, n_nodes=n, k=32, n_layers=3, latent_dim=68, n_traits=4, symmetric_input=True).to(DEV)
    with torch.no_grad():
        model.log_sigma.fill_(1.0)
    tr = M.Trainer(model, "mdn")
    losses = []
    with ops.arithmetic(**arith):
        for s in range(3):
            xb = x * (1.0 + 0.05 * s)
            if graph and s == 0:
                tr.capture((xb, adjacency, traits), xb, eps[s])
            step = tr.replay if graph else tr.step
            losses.append(float(step((xb, adjacency, traits), xb, eps[s]).item()))
    return losses, {k: v.detach().clone() for k, v in model.named_parameters()}


for arith in ({}, dict(wgrad="bf16x3"), dict(wgrad="bf16x6"), dict(factored_dgrad=False)):
    e, g_ = run(False, **arith), run(True, **arith)
    print(arith, "losses eager", e[0], "graph", g_[0])
    worst = max((float((g_[1][k] - e[1][k]).abs().max()), k) for k in e[1])
    print("   worst param diff", worst)
