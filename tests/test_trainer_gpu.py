"""Adam kernel and whole training steps on the GPU against the CPU oracle."""
import pytest
import torch

from conftest import load_fixture, fixture_params
import recipe
from oracle import ref_eager as R
import mdnvae_amd as M
from mdnvae_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_adam_kernel_matches_torch_adam():
    g = torch.Generator().manual_seed(2)
    n = 10007
    p0 = torch.randn(n, generator=g)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref], lr=1e-3)
    p = p0.to(DEV)
    m = torch.zeros(n, device=DEV)
    v = torch.zeros(n, device=DEV)
    for step in range(1, 6):
        grad = torch.randn(n, generator=g) * (10.0 ** (step - 3))
        ref.grad = grad.clone()
        opt.step()
        ops.call("mdnvae_adam_step", ops.ptr(p), ops.ptr(grad.to(DEV)), ops.ptr(m), ops.ptr(v), n, step, 1e-3, 0.9, 0.999,
                 1e-8, 1.0, ops.stream())
        assert float((p.cpu() - ref.detach()).abs().max()) <= 2e-7 * float(ref.detach().abs().max())


def test_two_training_steps_match_the_oracle():
    """function/MM_Gate.py:195-208 for two steps: same data, same epsilon, Adam lr 1e-3."""
    fix = load_fixture("mm_gate")
    params = fixture_params(fix)
    model = M.GCNVAE(R=5, n_nodes=68, k=32, n_layers=3, latent_dim=68)
    model.load_state_dict(params)
    model = model.to(DEV)
    adjacency = [list(map(int, r)) for r in fix["adjacency"]]
    x = torch.from_numpy(fix["x"])
    eps = [torch.from_numpy(fix["eps"]), torch.from_numpy(fix["eps"]).flip(0)]
    trainer = M.Trainer(model, "poisson_log")
    got = [trainer.step((x.to(DEV), adjacency), x.to(DEV), eps=e.to(DEV)).item() for e in eps]

    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    state = {k: (torch.zeros_like(v), torch.zeros_like(v)) for k, v in p.items()}
    want = []
    for t, e in enumerate(eps):
        for v in p.values():
            v.grad = None
        recon, mu, lv, _, _ = R.vae_forward("mm_gate", p, x, e, adjacency=fix["adjacency"])
        loss = R.vae_loss(recon, x, mu, lv, "poisson_log")
        loss.backward()
        want.append(loss.item())
        with torch.no_grad():
            R.adam_step(p, {k: v.grad for k, v in p.items()}, state, t + 1)
    assert abs(got[0] - want[0]) <= 1e-5 * abs(want[0])
    assert abs(got[1] - want[1]) <= 1e-4 * abs(want[1])
    for k, v in model.state_dict().items():
        assert float((v.cpu() - p[k].detach()).abs().max()) <= 5e-4, k     # two lr=1e-3 steps move a weight by <= 2e-3


def test_conv_vae_trainer_reduces_the_loss():
    torch.manual_seed(11111)
    model = M.ConvVAE(decoder="gcn", intercept="diag", final="none", n_nodes=68).to(DEV)
    x = M.data.synthetic_sc(16, 68, seed=1, device=DEV)
    model.set_mask([torch.ones(68, 68)] * 5)
    trainer = M.Trainer(model, "poisson_log")
    gen = torch.Generator(device=DEV).manual_seed(0)
    losses = [trainer.step((x,), x, eps=torch.randn(16, 68, device=DEV, generator=gen)).item() for _ in range(12)]
    assert all(l == l for l in losses)
    assert losses[-1] < losses[0]


def test_train_epoch_matches_the_oracle_loop():
    """train_epoch of function/MM_Gate.py:195-208 with the ragged last batch: 10 subjects in
    batches of 4 (4, 4, 2), unshuffled, epsilon from a seeded generator; the epoch metric is
    sum of losses / number of subjects."""
    fix = load_fixture("mm_gate")
    params = fixture_params(fix)
    model = M.GCNVAE(R=5, n_nodes=68, k=32, n_layers=3, latent_dim=68)
    model.load_state_dict(params)
    model = model.to(DEV)
    adjacency = [list(map(int, r)) for r in fix["adjacency"]]
    x = torch.from_numpy(fix["x"])
    data = torch.cat([x, x.flip(0), x])[:10]
    trainer = M.Trainer(model, "poisson_log")
    got = trainer.train_epoch(data.to(DEV), batch_size=4, extra=(adjacency,), shuffle=False,
                              eps_generator=torch.Generator(device=DEV).manual_seed(5))

    gen = torch.Generator(device=DEV).manual_seed(5)
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    state = {k: (torch.zeros_like(v), torch.zeros_like(v)) for k, v in p.items()}
    total = 0.0
    for t, lo in enumerate(range(0, 10, 4)):
        xb = data[lo:lo + 4]
        e = torch.randn(xb.shape[0], 68, device=DEV, generator=gen).cpu()
        for v in p.values():
            v.grad = None
        recon, mu, lv, _, _ = R.vae_forward("mm_gate", p, xb, e, adjacency=fix["adjacency"])
        loss = R.vae_loss(recon, xb, mu, lv, "poisson_log")
        loss.backward()
        total += loss.item()
        with torch.no_grad():
            R.adam_step(p, {k: v.grad for k, v in p.items()}, state, t + 1)
    assert abs(got - total / 10) <= 1e-4 * abs(total / 10)
    # shuffled epochs visit every subject once: the multiset of batch sizes is the same
    g = torch.Generator(device=DEV).manual_seed(1)
    m2 = trainer.train_epoch(data.to(DEV), batch_size=4, extra=(adjacency,), shuffle=True, generator=g,
                             eps_generator=torch.Generator(device=DEV).manual_seed(6))
    assert m2 == m2 and trainer.steps == 6


def test_hinge_step_runs_and_matches_manual_composition():
    """function_FC_SC/SC-FC.py:134-167 through Trainer.hinge_step: the loss equals the manual
    composition with the same permutation and epsilon (the composition itself is checked against
    the reference fixture in test_decoder_gpu)."""
    torch.manual_seed(3)
    model = M.cVAE(in_dim=100, latent_dim=8, hidden_dim=32).to(DEV)
    twin = M.cVAE(in_dim=100, latent_dim=8, hidden_dim=32).to(DEV)
    twin.load_state_dict(model.state_dict())
    fc, sc = torch.rand(6, 100, device=DEV), torch.rand(6, 100, device=DEV)
    ea, eb = torch.randn(6, 8, device=DEV), torch.randn(6, 8, device=DEV)
    trainer = M.Trainer(model, "mse")
    got = trainer.hinge_step(fc, sc, ea, eb, generator=torch.Generator(device=DEV).manual_seed(9))
    idx = torch.randperm(6, device=DEV, generator=torch.Generator(device=DEV).manual_seed(9))[:3]
    fc_no = fc.clone()
    fc_no[idx] = 0.0
    r, mu, lv = twin(fc, sc, eps=ea)
    a = M.loss_function_mse_terms(r, sc, mu, lv)[0]
    r2, mu2, lv2 = twin(fc_no, sc, eps=eb)
    b = M.loss_function_mse_terms(r2, sc, mu2, lv2)[0]
    want = a + torch.relu(a - b + 0.01)
    assert abs(got.item() - want.item()) <= 1e-6 * abs(want.item())


def test_mdn_trainer_step_reduces_the_loss():
    torch.manual_seed(11111)
    model = M.MDNGateVAE(R=5, n_nodes=68, k=32, n_layers=3, latent_dim=68, n_traits=4).to(DEV)
    x = M.data.synthetic_sc(16, 68, seed=1, device=DEV)
    traits = torch.randn(16, 4, device=DEV)
    adjacency = [[(u + j) % 68 for j in range(32)] for u in range(68)]
    trainer = M.Trainer(model, "mdn")
    gen = torch.Generator(device=DEV).manual_seed(0)
    losses = [trainer.step((x, adjacency, traits), x, eps=torch.randn(16, 68, device=DEV, generator=gen)).item()
              for _ in range(10)]
    assert all(l == l for l in losses) and losses[-1] < losses[0]


@pytest.mark.parametrize("B,N,M", [(4096, 128, 1024), (2048, 68, 128), (1500, 4, 8), (1031, 132, 260), (33, 64, 128)])
def test_small_layer_weight_gradient_kernel(B, N, M):
    """grad_w = grad_pre^T . x of the split-over-the-batch kernel against float64."""
    g = torch.Generator().manual_seed(B + N)
    gp = torch.randn(B, N, generator=g)
    x = torch.randn(B, M, generator=g)
    got = ops._linear_wgrad(gp.to(DEV), x.to(DEV))
    want = gp.double().t() @ x.double()
    assert got.shape == (N, M)
    assert float((got.double().cpu() - want).abs().max()) <= 1e-5 * float(want.abs().max())
    # reproducible: fixed summation order
    again = ops._linear_wgrad(gp.to(DEV), x.to(DEV))
    assert torch.equal(got, again)
    # the same launch pair also giving the bias gradient (column sums of grad_pre): same weight gradient bit for
    # bit, bias gradient against float64; then both added onto what the outputs hold
    gw, gb = ops._linear_wgrad_bias(gp.to(DEV), x.to(DEV), None, None)
    assert torch.equal(gw, got)
    want_b = gp.double().sum(0)
    assert float((gb.double().cpu() - want_b).abs().max()) <= 1e-5 * float(want_b.abs().max()) + 1e-4
    w_par = torch.nn.Parameter(torch.zeros(N, M, device=DEV))
    b_par = torch.nn.Parameter(torch.zeros(N, device=DEV))
    w_par.grad, b_par.grad = torch.full((N, M), 2.0, device=DEV), torch.full((N,), 3.0, device=DEV)
    with ops.accumulate_into_grads():
        none_w, none_b = ops._linear_wgrad_bias(gp.to(DEV), x.to(DEV), w_par, b_par)
    assert none_w is None and none_b is None
    assert float((w_par.grad - 2.0 - got).abs().max()) <= 1e-5 * float(want.abs().max())
    assert float((b_par.grad - 3.0 - gb).abs().max()) <= 1e-5 * float(want_b.abs().max()) + 1e-4


@pytest.mark.parametrize("B,N,M,act", [(16384, 5, 4, None), (1000, 8, 8, "tanh"), (257, 1, 1, "relu"), (3, 2, 7, "sigmoid"),
                                       (1, 5, 4, None)])
def test_narrow_layer_against_float64(B, N, M, act):
    """A layer at most 8 wide (the mixture gate, traits -> K logits): forward, weight, bias and
    input gradients of the one-kernel path against float64 autograd; gradients also when added
    straight into .grad."""
    g = torch.Generator().manual_seed(B + N)
    x = torch.randn(B, M, generator=g)
    w = torch.randn(N, M, generator=g)
    b = torch.randn(N, generator=g)
    up = torch.randn(B, N, generator=g)
    xd, wd, bd = (t.to(DEV).requires_grad_(True) for t in (x, w, b))
    y = ops.linear_act(xd, wd, bd, act)
    y.backward(up.to(DEV))
    x64, w64, b64 = (t.double().requires_grad_(True) for t in (x, w, b))
    pre = x64 @ w64.t() + b64
    y64 = {None: pre, "tanh": torch.tanh(pre), "relu": torch.relu(pre), "sigmoid": torch.sigmoid(pre)}[act]
    y64.backward(up.double())

    def close(got, want):
        return float((got.double().cpu() - want).abs().max()) <= 1e-5 * max(float(want.abs().max()), 1e-30)

    assert close(y.detach(), y64.detach())
    assert close(wd.grad, w64.grad) and close(bd.grad, b64.grad) and close(xd.grad, x64.grad)
    # the sink path adds to what .grad holds
    w2, b2 = (t.to(DEV).requires_grad_(True) for t in (w, b))
    w2.grad, b2.grad = torch.ones_like(w2), torch.ones_like(b2)
    with ops.accumulate_into_grads():
        ops.linear_act(x.to(DEV), w2, b2, act).backward(up.to(DEV))
    assert close(w2.grad - 1.0, w64.grad) and close(b2.grad - 1.0, b64.grad)
    # no bias
    w3 = w.to(DEV).requires_grad_(True)
    ops.linear_act(x.to(DEV), w3, None, act).sum().backward()
    w3_64 = w.double().requires_grad_(True)
    pre = x.double() @ w3_64.t()
    {None: pre, "tanh": torch.tanh(pre), "relu": torch.relu(pre), "sigmoid": torch.sigmoid(pre)}[act].sum().backward()
    assert close(w3.grad, w3_64.grad)


def test_captured_step_replays_the_eager_step():
    """Trainer.capture / replay: three replayed steps leave the same parameters and losses as three
    eager steps from the same start (same kernels, same order), and capture itself changes nothing."""
    fix = load_fixture("mm_gate")
    params = fixture_params(fix)
    x = torch.from_numpy(fix["x"]).to(DEV)
    adjacency = [list(map(int, r)) for r in fix["adjacency"]]
    traits = torch.randn(x.shape[0], 4, generator=torch.Generator().manual_seed(1)).to(DEV)
    g = torch.Generator().manual_seed(3)
    batches = [(x * (1.0 + 0.1 * i), torch.randn(x.shape[0], 68, generator=g).to(DEV)) for i in range(3)]
    out = {}
    for mode in ("eager", "graph"):
        torch.manual_seed(5)                      # the gate layer is not in the fixture: same random init both times
        model = M.MDNGateVAE(R=5, n_nodes=68, k=32, n_layers=3, latent_dim=68, n_traits=4)
        model.load_state_dict(params, strict=False)
        with torch.no_grad():
            model.log_sigma.fill_(1.0)
        trainer = M.Trainer(model.to(DEV), "mdn")
        start = trainer.buf.flat_p.clone()
        losses = []
        if mode == "graph":
            trainer.capture((batches[0][0], adjacency, traits), batches[0][0], batches[0][1])
            assert torch.equal(trainer.buf.flat_p, start) and trainer.steps == 0
        for xb, eps in batches:
            if mode == "graph":
                losses.append(float(trainer.replay((xb, adjacency, traits), xb, eps).item()))
            else:
                losses.append(float(trainer.step((xb, adjacency, traits), xb, eps=eps).item()))
        out[mode] = (losses, trainer.buf.flat_p.clone(), trainer.steps)
    assert out["graph"][2] == out["eager"][2] == 3
    for a, b in zip(out["graph"][0], out["eager"][0]):
        assert abs(a - b) <= 1e-6 * abs(b)
    assert float((out["graph"][1] - out["eager"][1]).abs().max()) <= 1e-6


def test_graph_epoch_equals_eager_epoch():
    """train_epoch(graph=True): full batches replayed from one captured graph, ragged tail stepped
    eagerly - same epoch metric and parameters as the all-eager epoch."""
    fix = load_fixture("mm_gate")
    params = fixture_params(fix)
    x = torch.from_numpy(fix["x"]).to(DEV)
    data = torch.cat([x * (1.0 + 0.05 * i) for i in range(5)])[:19]          # 19 subjects: 4 batches of 4 + 3
    adjacency = [list(map(int, r)) for r in fix["adjacency"]]
    out = {}
    for graph in (False, True):
        model = M.GCNVAE(R=5, n_nodes=68, k=32, n_layers=3, latent_dim=68)
        model.load_state_dict(params, strict=False)
        trainer = M.Trainer(model.to(DEV), "poisson_log")
        metric = trainer.train_epoch(data, batch_size=4, extra=(adjacency,), shuffle=True,
                                     generator=torch.Generator(device=DEV).manual_seed(1),
                                     eps_generator=torch.Generator(device=DEV).manual_seed(2), graph=graph)
        out[graph] = (metric, trainer.buf.flat_p.clone(), trainer.steps)
    assert out[True][2] == out[False][2] == 5
    assert abs(out[True][0] - out[False][0]) <= 1e-6 * abs(out[False][0])
    assert float((out[True][1] - out[False][1]).abs().max()) <= 1e-6


def test_two_graph_epochs_with_ragged_tails_equal_eager_epochs():
    """Replays after eager steps: the ragged tail of epoch one is stepped call by call, so the
    device-side Adam step counter of the captured graph must be brought level before the replays
    of epoch two (its bias corrections depend on the count).  The adjacency is a device tensor
    here: its conversion must not put a device-to-host copy inside the capture."""
    fix = load_fixture("mm_gate")
    params = fixture_params(fix)
    x = torch.from_numpy(fix["x"]).to(DEV)
    data = torch.cat([x * (1.0 + 0.05 * i) for i in range(5)])[:19]          # 19 subjects: 4 batches of 4 + 3
    adjacency = torch.tensor([list(map(int, r)) for r in fix["adjacency"]], dtype=torch.int64, device=DEV)
    out = {}
    for graph in (False, True):
        model = M.GCNVAE(R=5, n_nodes=68, k=32, n_layers=3, latent_dim=68)
        model.load_state_dict(params, strict=False)
        trainer = M.Trainer(model.to(DEV), "poisson_log")
        gen, eg = torch.Generator(device=DEV).manual_seed(1), torch.Generator(device=DEV).manual_seed(2)
        metrics = [trainer.train_epoch(data, batch_size=4, extra=(adjacency,), shuffle=True, generator=gen,
                                       eps_generator=eg, graph=graph) for _ in range(2)]
        # interleave once more: an eager step, then a replay-capable epoch
        out[graph] = (metrics, trainer.buf.flat_p.clone(), trainer.steps)
    assert out[True][2] == out[False][2] == 10
    for a, b in zip(out[True][0], out[False][0]):
        assert abs(a - b) <= 1e-6 * abs(b)
    assert float((out[True][1] - out[False][1]).abs().max()) <= 1e-6


def test_direct_gradients_equal_autograd_accumulation():
    """Trainer(direct_grads=True): the kernels add parameter gradients straight into the flat
    buffer; the result is the buffer autograd's own accumulation leaves, for the FNN/GCN models
    (fused and plain loss) and a conv-encoder model, and accumulates over two backward passes."""
    fix = load_fixture("mm_gate")
    params = fixture_params(fix)
    x = torch.from_numpy(fix["x"]).to(DEV)
    x = 0.5 * (x.reshape(-1, 68, 68) + x.reshape(-1, 68, 68).transpose(1, 2)).reshape(-1, 68 * 68).contiguous()
    eps = torch.from_numpy(fix["eps"]).to(DEV)
    adjacency = [list(map(int, r)) for r in fix["adjacency"]]
    for fused in (True, False):
        got = {}
        for direct in (False, True):
            model = M.GCNVAE(R=5, n_nodes=68, k=32, n_layers=3, latent_dim=68, symmetric_input=True)
            model.load_state_dict(params, strict=False)
            trainer = M.Trainer(model.to(DEV), "poisson_log", fused_loss=fused, direct_grads=direct)
            trainer.forward_backward((x, adjacency), x, eps=eps)
            # by name: the two trainers lay the flat buffers out differently
            got[direct] = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
        scale = max(float(g.abs().max()) for g in got[False].values())
        for name, want in got[False].items():
            assert float((got[True][name] - want).abs().max()) <= 1e-6 * scale, (fused, name)
    # += semantics: a second backward without zeroing doubles the buffer
    model = M.GCNVAE(R=5, n_nodes=68, k=32, n_layers=3, latent_dim=68, symmetric_input=True)
    model.load_state_dict(params, strict=False)
    model = model.to(DEV)
    buf = M.trainer.FlatBuffers(model)
    buf.zero_grad()
    for _ in range(2):
        with ops.accumulate_into_grads():
            recon, mu, lv = model(x, adjacency, eps=eps)
            ops.vae_loss(recon, x, mu, lv, "poisson_log").backward()
    twice = buf.flat_g.clone()
    buf.zero_grad()
    recon, mu, lv = model(x, adjacency, eps=eps)
    ops.vae_loss(recon, x, mu, lv, "poisson_log").backward()
    assert float((twice - 2.0 * buf.flat_g).abs().max()) <= 2e-6 * float(buf.flat_g.abs().max())


def test_headline_step_at_full_size_is_additive_over_subjects_and_path_independent():
    """Size-independent properties at the bench's full size (mdn_gate, B = 16384, N = 87): the
    sum-reduced loss and every gradient of the full batch equal the sums over its two halves (what
    the SUM all-reduce of two ranks would deliver), and the symmetric-input path (packed triangle,
    folded weights, streaming likelihood) agrees with the general path on the same batch.
    Then against float64: the contribution of the first 4096 subjects to the full-size run's own gradients
    (full run minus the run on the other 12288 subjects - gradients add over subjects), and the same subjects
    run on their own, against a float64 step on them that takes the ReLU gates of the encoder towers
    (1024 + 128 units each) from the device's forward pass, so that the comparison shows arithmetic, not which
    side of zero a pre-activation at rounding level fell (one such flip in the 128-unit layer moves the
    first-layer weight gradient of that tower by 1e-4 of its norm, on the library path as on this one)."""
    B, n, S = 16384, 87, 4096
    gen = torch.Generator(device=DEV).manual_seed(1234)
    x = M.data.synthetic_sc(B, n, seed=1234, device=DEV).reshape(B, n * n).contiguous()
    traits = torch.randn(B, 4, device=DEV, generator=gen)
    eps = torch.randn(B, 68, device=DEV, generator=gen)
    adjacency = [[(i + d) % n for d in range(32)] for i in range(n)]

    def grads(symmetric, rows, want_gates=0):
        torch.manual_seed(11111)
        model = M.MDNGateVAE(R=5, n_nodes=n, k=32, n_layers=3, latent_dim=68, n_traits=4, symmetric_input=symmetric)
        model.log_sigma.data.fill_(3.0)          # broad components: responsibilities stay mixed
        trainer = M.Trainer(model.to(DEV), "mdn")
        xs = x[rows].contiguous()
        loss = trainer.forward_backward((xs, adjacency, traits[rows].contiguous()), xs, eps=eps[rows].contiguous())
        model.check_inputs()
        out = float(loss.item()), {k: p.grad.detach().double().clone() for k, p in model.named_parameters()}
        if want_gates:
            with torch.no_grad():
                a = model.first_layers(xs)
                h = (model.fc111(a[0], act="relu"), model.fc222(a[1], act="relu"))
                out += (tuple((t[:want_gates] > 0).cpu() for t in a + h),)
        return out

    full = grads(True, slice(0, B), S)
    lo, hi = grads(True, slice(0, B // 2)), grads(True, slice(B // 2, B))
    plain = grads(False, slice(0, B), S)
    assert abs(full[0] - (lo[0] + hi[0])) <= 1e-5 * abs(full[0])
    assert abs(full[0] - plain[0]) <= 1e-5 * abs(full[0])
    for name, g in full[1].items():
        scale = float(g.abs().max())
        assert float((g - (lo[1][name] + hi[1][name])).abs().max()) <= 1e-4 * scale, name
        # the two first layers round differently (folded weights, other summation order), so the
        # ReLU masks of a few near-zero pre-activations among the 16 M differ; each flip moves one
        # subject's whole rank-one term (entries up to ~95) in the first-layer weight gradients:
        # ~1e-3 of their largest entry.  An indexing or scaling mistake would show as O(1).
        assert float((g - plain[1][name]).abs().max()) <= 5e-3 * scale, name
        assert float((g - plain[1][name]).norm() / g.norm()) <= 1e-4, name
        recipe.record("headline step, B=16384, N=87", "hip", name, float((g - plain[1][name]).abs().max()) / scale, 5e-3,
                      "symmetric path (two-part fp16 first layers) vs general path (library fp32), max entry / largest entry")
        recipe.record("headline step, B=16384, N=87", "hip", name, float((g - plain[1][name]).norm() / g.norm()), 1e-4,
                      "symmetric path (two-part fp16 first layers) vs general path (library fp32), L2 distance / norm")

    # float64 on the first S subjects, each path under its own first-layer gates
    rows = slice(0, S)
    rest = slice(S, B)
    torch.manual_seed(11111)
    ref = M.MDNGateVAE(R=5, n_nodes=n, k=32, n_layers=3, latent_dim=68, n_traits=4)
    ref.log_sigma.data.fill_(3.0)
    state = {k: v.detach().double().clone() for k, v in ref.state_dict().items()}
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

    err = {}
    fix = f"headline step B=16384, N=87: its first {S} subjects"
    for tag, symmetric, whole in (("symmetric path (two-part fp16 first layers)", True, full),
                                  ("general path (library fp32 GEMMs)", False, plain)):
        other = grads(symmetric, rest)
        direct = grads(symmetric, rows, S)
        assert all(torch.equal(a, b) for a, b in zip(whole[2], direct[2]))     # a subject's gates do not depend on the batch
        want_loss, want = float64_step(whole[2])
        assert abs((whole[0] - other[0]) - want_loss) <= 2e-5 * abs(want_loss)
        assert abs(direct[0] - want_loss) <= 1e-5 * abs(want_loss)
        for name, g in whole[1].items():
            w = want[name]
            share = (g - other[1][name]).cpu()             # the first S subjects' share of the full-size run
            own = direct[1][name].cpu()
            for how, got, bar_l2, bar_max in (("share of the full-size run", share, 5e-6, 1e-4), ("run on their own", own, 2e-6, 1e-5)):
                err_max = float((got - w).abs().max() / w.abs().max())
                err_l2 = float((got - w).norm() / w.norm())
                note = f"{tag}, {how}, vs float64 under the device's tower gates"
                recipe.record(fix, "hip", name, err_l2, bar_l2, note + ", L2 distance / norm")
                recipe.record(fix, "hip", name, err_max, bar_max, note + ", max entry / largest entry")
                assert err_l2 <= bar_l2 and err_max <= bar_max, (tag, how, name, err_l2, err_max)
                err.setdefault((name, how), []).append((err_l2, err_max))
    for (name, how), ((l2_sym, max_sym), (l2_gen, max_gen)) in err.items():
        # the hand-written path may not be further from float64 than three times the library fp32 path
        # (floors: the share is a difference of two fp32-rounded sums four times its size; a bias gradient that comes
        # out of the weight-gradient product as one long fp32 dot product per entry carries 2e-6 of that)
        floor_l2, floor_max = (3e-6, 1e-5) if how.startswith("share") else (1e-6, 3e-6)
        assert l2_sym <= max(3.0 * l2_gen, floor_l2), (name, how, l2_sym, l2_gen)
        assert max_sym <= max(3.0 * max_gen, floor_max), (name, how, max_sym, max_gen)


def test_batches_either_side_of_the_split_product_threshold_agree():
    """Below ops.SPLIT_GEMM_MIN_BATCH subjects the first layers run on the library's fp32 GEMM, from there on on
    the hand-written two-part fp16 product: a batch of 1023 and the same subjects in a batch of 1024 give the same
    per-subject outputs to fp32 rounding (1e-5 of the largest) and - up to the 1024th subject's share - the same
    gradients to 3e-4 of their norm: the two products round differently, so a ReLU pre-activation at rounding level
    (here one of the 1023 x 128 of the second tower layer) can fall on either side of zero, and one such flip is
    1e-4 of that tower's first-layer weight gradient.  The same holds between any two fp32 summation orders."""
    n = 68
    Bs = ops.SPLIT_GEMM_MIN_BATCH
    x = M.data.synthetic_sc(Bs, n, seed=21, device=DEV).reshape(Bs, n * n).contiguous()
    g = torch.Generator(device=DEV).manual_seed(3)
    traits = torch.randn(Bs, 4, device=DEV, generator=g)
    eps = torch.randn(Bs, 68, device=DEV, generator=g)
    adjacency = [[(i + d) % n for d in range(32)] for i in range(n)]

    def run(rows):
        torch.manual_seed(11111)
        model = M.MDNGateVAE(R=5, n_nodes=n, k=32, n_layers=3, latent_dim=68, n_traits=4, symmetric_input=True)
        model.log_sigma.data.fill_(3.0)
        trainer = M.Trainer(model.to(DEV), "mdn")
        calls = []
        orig = ops.call

        def spy(name, *a):
            calls.append(name)
            return orig(name, *a)

        ops.call = spy
        try:
            xs = x[rows].contiguous()
            loss = trainer.forward_backward((xs, adjacency, traits[rows].contiguous()), xs, eps=eps[rows].contiguous())
            with torch.no_grad():
                mu, _ = model.encode(xs)
        finally:
            ops.call = orig
        return float(loss.item()), mu, {k: p.grad.detach().clone() for k, p in model.named_parameters()}, calls

    below = run(slice(0, Bs - 1))
    at = run(slice(0, Bs))
    last = run(slice(Bs - 1, Bs))
    assert "mdnvae_split_gemm_f16" in at[3] and "mdnvae_split_gemm_f16" not in below[3]
    assert float((at[1][:Bs - 1] - below[1]).abs().max()) <= 1e-5 * float(below[1].abs().max())
    assert abs(at[0] - (below[0] + last[0])) <= 1e-5 * abs(at[0])
    for name, gfull in at[2].items():
        diff = float((gfull - (below[2][name] + last[2][name])).norm() / gfull.norm())
        recipe.record("first layers at B=1023 (library fp32) + B=1 vs B=1024 (two-part fp16 product), N=68", "hip", name, diff,
                      3e-4, "L2 distance / norm of the gradients")
        assert diff <= 3e-4, (name, diff)


def test_replayed_steps_at_the_split_product_batch_size_equal_eager_steps():
    """Trainer.capture / replay where the first layers take the two-part fp16 product (B = 1024): the replayed
    graph clears and refills the largest-magnitude records of its operands with every batch, so three steps on
    batches of different magnitude give the losses of three eager steps (a record cleared by a memset node
    instead of a kernel came out short in the second replay: the memset overlapped the kernel that fills it)."""
    n, B = 68, 1024
    x = M.data.synthetic_sc(B, n, seed=77, device=DEV)
    g = torch.Generator().manual_seed(9)
    eps = torch.randn(3, B, 68, generator=g).to(DEV)
    traits = torch.randn(B, 4, generator=g).to(DEV)
    adjacency = [[(i + d) % n for d in range(32)] for i in range(n)]

    def run(graph):
        torch.manual_seed(5)
        model = M.MDNGateVAE(R=5, n_nodes=n, k=32, n_layers=3, latent_dim=68, n_traits=4, symmetric_input=True).to(DEV)
        with torch.no_grad():
            model.log_sigma.fill_(1.0)
        tr = M.Trainer(model, "mdn")
        losses, records = [], []
        for s in range(3):
            xb = x * (1.0 + 0.3 * s)
            if graph and s == 0:
                tr.capture((xb, adjacency, traits), xb, eps[s])
            step = tr.replay if graph else tr.step
            losses.append(float(step((xb, adjacency, traits), xb, eps[s]).item()))
            records.append(ops.absmax_value(model._xmax))
            assert records[-1] == float(xb.abs().max())
        return losses, {k: v.detach().clone() for k, v in model.named_parameters()}

    eager, replayed = run(False), run(True)
    assert eager[0] == replayed[0]
    for name, want in eager[1].items():
        assert float((replayed[1][name] - want).abs().max()) <= 1e-5 * float(want.abs().max()), name


def test_folded_adam_steps_equal_unfolded_steps():
    """Trainer(folded_adam=True) - moments of the symmetric-input first-layer weights kept folded,
    both mirror entries stepped from one value - against the trainer that unfolds the gradient and
    runs the flat Adam: same losses and parameters after several eager steps and after graph replays."""
    fix = load_fixture("mm_gate")
    params = fixture_params(fix)
    x = torch.from_numpy(fix["x"]).reshape(-1, 68, 68)
    x = (0.5 * (x + x.transpose(1, 2))).reshape(-1, 68 * 68).contiguous().to(DEV)
    eps = torch.from_numpy(fix["eps"]).to(DEV)
    adjacency = [list(map(int, r)) for r in fix["adjacency"]]
    out = {}
    for folded in (False, True):
        model = M.GCNVAE(R=5, n_nodes=68, k=32, n_layers=3, latent_dim=68, symmetric_input=True)
        model.load_state_dict(params, strict=False)
        trainer = M.Trainer(model.to(DEV), "poisson_log", folded_adam=folded)
        assert bool(trainer._fold) == folded
        losses = [float(trainer.step((x, adjacency), x, eps=eps).item()) for _ in range(3)]
        trainer.capture((x, adjacency), x, eps)
        losses += [float(trainer.replay((x, adjacency), x, eps).item()) for _ in range(2)]
        model.check_inputs()
        out[folded] = (losses, {k: v.detach().clone() for k, v in model.named_parameters()})
        if folded:      # forward_backward on its own still leaves every .grad filled in
            trainer.forward_backward((x, adjacency), x, eps=eps)
            g11 = model.fc11.weight.grad.reshape(-1, 68, 68)
            assert float(g11.abs().max()) > 0 and torch.equal(g11, g11.transpose(1, 2))
    # same formulas, two compilations of them: agreement to rounding, carried over five steps
    for a, b in zip(out[True][0], out[False][0]):
        assert abs(a - b) <= 1e-5 * abs(b)
    for name, want in out[False][1].items():
        got = out[True][1][name]
        assert float((got - want).abs().max()) <= 2e-5 * float(want.abs().max()), name
