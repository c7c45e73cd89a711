"""How far the headline step's gradients are from float64: full-size run minus the run without its first S subjects,
against a float64 step on those subjects under the device's first-layer gates; also the same subjects run directly."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mdnvae_amd as M
from mdnvae_amd import ops
from oracle import ref_eager as R

DEV = "cuda:0"
B, n, S = 16384, 87, 4096
gen = torch.Generator(device=DEV).manual_seed(1234)
x = M.data.synthetic_sc(B, n, seed=1234, device=DEV).reshape(B, n * n).contiguous()
traits = torch.randn(B, 4, device=DEV, generator=gen)
eps = torch.randn(B, 68, device=DEV, generator=gen)
adjacency = [[(i + d) % n for d in range(32)] for i in range(n)]


def grads(symmetric, rows, want_gates=0, **kw):
    torch.manual_seed(11111)
    model = M.MDNGateVAE(R=5, n_nodes=n, k=32, n_layers=3, latent_dim=68, n_traits=4, symmetric_input=symmetric)
    model.log_sigma.data.fill_(3.0)
    trainer = M.Trainer(model.to(DEV), "mdn", **kw)
    xs = x[rows].contiguous()
    loss = trainer.forward_backward((xs, adjacency, traits[rows].contiguous()), xs, eps=eps[rows].contiguous())
    out = float(loss.item()), {k: p.grad.detach().double().clone() for k, p in model.named_parameters()}
    if want_gates:
        with torch.no_grad():
            a = model.first_layers(xs)
            h = (model.fc111(a[0], act="relu"), model.fc222(a[1], act="relu"))
            out += (tuple((t[:want_gates] > 0).cpu() for t in a), tuple((t[:want_gates] > 0).cpu() for t in h))
    return out


torch.manual_seed(11111)
ref = M.MDNGateVAE(R=5, n_nodes=n, k=32, n_layers=3, latent_dim=68, n_traits=4)
ref.log_sigma.data.fill_(3.0)
state = {k: v.detach().double().clone() for k, v in ref.state_dict().items()}
rows, rest = slice(0, S), slice(S, B)
x64 = x[rows].double().cpu()


def float64_step(gates):
    p = {k: v.clone().requires_grad_(True) for k, v in state.items()}
    mu, lv = R.fnn_encode(p, x64, gates=gates)
    z = R.reparameterize(mu, lv, eps[rows].double().cpu())
    comp = R.gcn_components(p, z, adjacency)
    logits = R.mdn_gate_logits(traits[rows].double().cpu(), p["gate.weight"], p["gate.bias"])
    total = R.mdn_nll(comp, p["log_sigma"], logits, x64) + R.kl_term(mu, lv)
    total.backward()
    return float(total.item()), {k: v.grad for k, v in p.items()}


for tag, symmetric, kw in (("sym f16x2", True, {}), ("sym bf16x3", True, dict(fast_wgrad=True)), ("general library", False, {})):
    whole = grads(symmetric, slice(0, B), S, **kw)
    other = grads(symmetric, rest, **kw)
    direct = grads(symmetric, rows, S, **kw)
    print(tag, "gates equal full vs direct:", all(torch.equal(a, b) for a, b in zip(whole[2], direct[2])))
    _, want = float64_step(whole[2] + whole[3])
    _, want_free = float64_step(None)
    # second-layer gate flips between the device and float64
    p = {k: v for k, v in state.items()}
    for name in want:
        w, wf = want[name], want_free[name]
        sub = (whole[1][name] - other[1][name]).cpu()
        d = direct[1][name].cpu()
        f = lambda a, b: (float((a - b).norm() / b.norm()), float((a - b).abs().max() / b.abs().max()))
        print(f"  {name:40s} sub-vs-gated {f(sub, w)[0]:.2e}/{f(sub, w)[1]:.2e}  direct-vs-gated {f(d, w)[0]:.2e}/{f(d, w)[1]:.2e}"
              f"  direct-vs-free {f(d, wf)[0]:.2e}/{f(d, wf)[1]:.2e}  sub-vs-direct {f(sub, d)[0]:.2e}/{f(sub, d)[1]:.2e}")
