"""One-rank gloo group, split-graph replay at the split-product batch size: where does the second replay go wrong?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import mdnvae_amd as M
from mdnvae_amd import ops, trainer as TR

DEV = "cuda:0"
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:29618", world_size=1, rank=0)
TR.MIN_RANKS_FOR_EXCHANGE = 1
n, B = 68, 1024
x = M.data.synthetic_sc(B, n, seed=77, device=DEV)
g = torch.Generator().manual_seed(9)
eps = torch.randn(3, B, 68, generator=g).to(DEV)
traits = torch.randn(B, 4, generator=g).to(DEV)
adjacency = [[(i + d) % n for d in range(32)] for i in range(n)]


def make():
    torch.manual_seed(5)
    model = M.MDNGateVAE(R=5, n_nodes=n, k=32, n_layers=3, latent_dim=68, n_traits=4, symmetric_input=True).to(DEV)
    with torch.no_grad():
        model.log_sigma.fill_(1.0)
    return model, M.Trainer(model, "mdn")


V = sys.argv[1] if len(sys.argv) > 1 else "V0"
mg, tg = make()
x0, x1 = x * 1.0, x * 1.05
tg.capture((x0, adjacency, traits), x0, eps[0])
def a_only(xx, ee):
    tg._g_inputs[0].copy_(xx)
    tg._g_inputs[2].copy_(traits)
    tg._g_eps.copy_(ee)
    tg._graph.replay()
    torch.cuda.synchronize()
    print("     xmax record", ops.absmax_value(mg._xmax), "slots nonzero", int((mg._xmax != 0).sum()), "asym", int(mg._asym.item()),
          "flat_g finite", bool(torch.isfinite(tg.buf.flat_g).all()), "|g|", float(tg.buf.flat_g[tg._fold_end:].double().abs().sum()))
    return float(tg._g_loss.item())


if V == "T2":
    print("T2 A-only: x0", a_only(x0, eps[0]), "x1", a_only(x1, eps[1]), "x0 again", a_only(x0, eps[0]), "x1 again", a_only(x1, eps[1]),
          "(eager: 9228930.0 / 9244349.0)")
    print("T2 with x1 eps0", a_only(x1, eps[0]), " x0 eps1", a_only(x0, eps[1]))
    m2, t2 = make()
    print("   eager x1 eps0", float(t2.forward_backward((x1, adjacency, traits), x1, eps=eps[0]).item()),
          "x0 eps1", float(t2.forward_backward((x0, adjacency, traits), x0, eps=eps[1]).item()),
          "x1 eps1", float(t2.forward_backward((x1, adjacency, traits), x1, eps=eps[1]).item()))
    sys.exit(0)
l0 = float(tg.replay((x0, adjacency, traits), x0, eps[0]).item())
if V == "V1":
    tg._split_graph = True
    tg.forward_backward((x1, adjacency, traits), x1, eps=eps[1], _defer_fold=True)
    tg._split_graph = False
    tg._folded.clear()
elif V == "V2":
    torch.cuda.synchronize()
elif V == "V3":
    with torch.no_grad():
        mg(x1, adjacency, traits, eps=eps[1], nodes_only=True)
elif V == "V4":
    junk = [torch.empty(1 << 20, device=DEV) for _ in range(8)]
elif V == "V5":
    tg.buf.zero_grad(0)
elif V == "V6":
    junk = torch.empty(64, device=DEV)
    junk.fill_(1.0)
elif V == "V7":
    m2, t2 = make()
    with torch.no_grad():
        t2._split_graph = True
    lo = float(t2.forward_backward((x1, adjacency, traits), x1, eps=eps[1]).item())
    print("loss of the step-1 batch under the INITIAL parameters:", lo)
elif V == "V8":
    tg._g_inputs[0].copy_(x1)
l1 = float(tg.replay((x1, adjacency, traits), x1, eps[1]).item())
print(V, "losses", l0, l1, "(eager: 9228930.0 9249375.0)")
