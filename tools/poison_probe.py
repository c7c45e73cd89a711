"""Run one training step with every torch.empty buffer pre-filled with a poison pattern: a result that changes
shows a kernel reading memory nothing wrote."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mdnvae_amd as M
from mdnvae_amd import ops

DEV = "cuda:0"
n = 68
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
x = M.data.synthetic_sc(B, n, seed=77, device=DEV)
g = torch.Generator().manual_seed(9)
eps = torch.randn(B, 68, generator=g).to(DEV)
traits = torch.randn(B, 4, generator=g).to(DEV)
adjacency = [[(i + d) % n for d in range(32)] for i in range(n)]
orig_empty = torch.empty
POISON = {"v": None}


def poisoned(*a, **kw):
    t = orig_empty(*a, **kw)
    if POISON["v"] is not None and t.is_cuda:
        if t.dtype.is_floating_point:
            t.fill_(POISON["v"])
        else:
            t.view(torch.uint8).fill_(0x7f if POISON["v"] != 0 else 0)
    return t


torch.empty = poisoned


def run(poison):
    POISON["v"] = poison
    torch.manual_seed(5)
    model = M.MDNGateVAE(R=5, n_nodes=n, k=32, n_layers=3, latent_dim=68, n_traits=4, symmetric_input=True).to(DEV)
    with torch.no_grad():
        model.log_sigma.fill_(1.0)
    tr = M.Trainer(model, "mdn")
    loss = float(tr.forward_backward((x, adjacency, traits), x, eps=eps).item())
    POISON["v"] = None
    return loss, {k: p.grad.detach().clone() for k, p in model.named_parameters()}


base = run(0.0)
for poison in (float("nan"), 1e30):
    got = run(poison)
    print("poison", poison, "loss", got[0], "vs", base[0])
    for k in base[1]:
        d = (got[1][k] - base[1][k]).abs().max()
        if not (float(d) == 0.0):
            print("   grad differs:", k, float(d))
