"""Host logic of the data-parallel trainer on the CPU: flat-buffer re-homing, subject sharding
and the SUM all-reduce across two gloo ranks (the N>1 path of bench.py uses the same calls
over RCCL)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import mdnvae_amd as M


def test_flat_buffers_alias_parameters_and_grads():
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Linear(7, 3))
    before = [p.detach().clone() for p in net.parameters()]
    buf = M.FlatBuffers(net)
    # every parameter starts at a multiple of 4 floats (16-byte aligned views), padding stays zero
    assert buf.numel == sum((p.numel() + 3) // 4 * 4 for p in net.parameters())
    assert all(o % 4 == 0 for o in buf.offsets)
    for p, b in zip(net.parameters(), before):
        assert torch.equal(p, b)
    net(torch.randn(4, 5)).sum().backward()
    for p, o in zip(buf.params, buf.offsets):
        assert p.grad.data_ptr() == buf.flat_g.data_ptr() + 4 * o
        assert torch.equal(buf.flat_g[o:o + p.numel()].view(p.shape), p.grad)
    buf.flat_p.mul_(2.0)
    for p, b in zip(net.parameters(), before):
        assert torch.equal(p, 2 * b)
    buf.zero_grad()
    assert float(buf.flat_g.abs().sum()) == 0.0


def test_shard_ranges_partition_the_subjects():
    for n in (0, 1, 7, 100, 1_000_003):
        for world in (1, 2, 3, 8):
            spans = [M.data.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(3)
    net = torch.nn.Sequential(torch.nn.Linear(6, 9), torch.nn.Tanh(), torch.nn.Linear(9, 2))
    buf = M.FlatBuffers(net)
    x = torch.arange(8 * 6, dtype=torch.float32).reshape(8, 6) / 10.0
    lo, hi = M.data.shard_range(8, rank, world)
    buf.zero_grad()
    (net(x[lo:hi]) ** 2).sum().backward()                 # sum-reduced loss on the local shard
    M.allreduce_sum(buf.flat_g, bucket_floats=16)         # several buckets on purpose
    out[rank] = buf.flat_g.clone()
    dist.destroy_process_group()


def test_two_rank_sum_allreduce_equals_single_process_gradient():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    torch.manual_seed(3)
    net = torch.nn.Sequential(torch.nn.Linear(6, 9), torch.nn.Tanh(), torch.nn.Linear(9, 2))
    buf = M.FlatBuffers(net)
    x = torch.arange(8 * 6, dtype=torch.float32).reshape(8, 6) / 10.0
    (net(x) ** 2).sum().backward()
    for r in range(world):
        assert torch.allclose(out[r], buf.flat_g, rtol=1e-6, atol=1e-7)
    assert torch.equal(out[0], out[1])


def test_trainer_refuses_cpu_modules():
    net = M.GCNVAE(R=2, n_nodes=10, k=3, n_layers=1, latent_dim=4)
    tr = M.Trainer(net, "poisson_log")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        tr.step((torch.rand(2, 100), [[0, 1, 2]] * 10), torch.rand(2, 100), eps=torch.randn(2, 4))


def _stats_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    st = M.FeatureStats(3, torch.tensor([1.0, 2.0, 3.0]), device="cpu")
    # accumulators as the moment kernel would have left them for this rank's rows
    x = torch.arange(30, dtype=torch.float64).reshape(10, 3) ** 1.5
    lo, hi = M.data.shard_range(10, rank, world)
    xc = x[lo:hi] - st.shift.double()
    st.n, st.sum, st.gram = hi - lo, xc.sum(0), xc.T @ xc
    st.allreduce()
    out[rank] = (st.n, st.sum.clone(), st.gram.clone())
    dist.destroy_process_group()


def test_two_rank_feature_stats_allreduce():
    world = 2
    out = mp.Manager().dict()
    mp.spawn(_stats_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    x = torch.arange(30, dtype=torch.float64).reshape(10, 3) ** 1.5
    xc = x - torch.tensor([1.0, 2.0, 3.0], dtype=torch.float64)
    for r in range(world):
        n, s, g = out[r]
        assert n == 10
        assert torch.allclose(s, xc.sum(0), rtol=1e-14)
        assert torch.allclose(g, xc.T @ xc, rtol=1e-14)


def _overlap_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(3)
    net = torch.nn.Sequential(torch.nn.Linear(6, 40), torch.nn.Tanh(), torch.nn.Linear(40, 2))
    tr = M.Trainer(net, "mse", overlap_floats=200)         # the 6x40 weight goes through the hook path
    assert tr._rest != [(0, tr.buf.numel)]
    x = torch.arange(8 * 6, dtype=torch.float32).reshape(8, 6) / 10.0
    lo, hi = M.data.shard_range(8, rank, world)
    tr.buf.zero_grad()
    (net(x[lo:hi]) ** 2).sum().backward()
    assert len(tr._pending) == 1
    for a, b in tr._rest:
        M.allreduce_sum(tr.buf.flat_g[a:b])
    for w in tr._pending:
        w.wait()
    out[rank] = tr.buf.flat_g.clone()
    dist.destroy_process_group()


def test_two_rank_overlapped_allreduce_equals_single_process_gradient():
    world = 2
    out = mp.Manager().dict()
    mp.spawn(_overlap_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    torch.manual_seed(3)
    net = torch.nn.Sequential(torch.nn.Linear(6, 40), torch.nn.Tanh(), torch.nn.Linear(40, 2))
    buf = M.FlatBuffers(net)
    x = torch.arange(8 * 6, dtype=torch.float32).reshape(8, 6) / 10.0
    (net(x) ** 2).sum().backward()
    for r in range(world):
        assert torch.allclose(out[r], buf.flat_g, rtol=1e-6, atol=1e-7)


# ---------------------------------------------------------------------------------------------
# The trainer's folded-gradient exchange with two ranks.  The HIP kernels cannot run here, so the
# seven entry points this path calls are restated in torch for the test (same argument lists);
# everything else - ops._SymLinearAct's gradient hand-over, Trainer._folded_grad, the early / late /
# bucketed all-reduces, the _rest ranges, _apply_adam's ordering - is the product code.
# ---------------------------------------------------------------------------------------------
def _tri_index(n):
    iu = torch.triu_indices(n, n)                       # row-major walk of the upper triangle
    return iu[0], iu[1]


_CALLS = {}


def _fake_call(fname, *a):
    _CALLS[fname] = _CALLS.get(fname, 0) + 1
    if fname == "mdnvae_tri_pack":
        x, B, n, ld, out = a[0], a[1], a[2], a[3], a[4]
        i, j = _tri_index(n)
        out.zero_()
        out[:, : i.numel()] = x.reshape(B, n, n)[:, i, j]
    elif fname == "mdnvae_tri_fold":
        w, H, n, ld, out = a[0], a[1], a[2], a[3], a[4]
        i, j = _tri_index(n)
        w3 = w.reshape(H, n, n)
        out.zero_()
        out[:, : i.numel()] = torch.where(i == j, w3[:, i, j], w3[:, i, j] + w3[:, j, i])
    elif fname == "mdnvae_tri_unfold":
        g, H, n, st, sh, gw, acc = a[0], a[1], a[2], a[3], a[4], a[5], a[6]
        i, j = _tri_index(n)
        assert (st, sh) == (H, 1)                        # the [T,H] layout of the weight-gradient GEMM
        full = torch.zeros(H, n, n)
        vals = g[: i.numel()].t()                        # [H,T]
        full[:, i, j] = vals
        full[:, j, i] = vals
        gw.copy_(gw + full.reshape(H, n * n) if acc else full.reshape(H, n * n))
    elif fname == "mdnvae_act_bwd":
        y, gy, gpre, _, act = a[0], a[1], a[2], a[3], a[4]
        assert act == 3                                  # relu
        gpre.copy_(gy * (y > 0))
    elif fname == "mdnvae_colsum":
        x, out, acc = a[0], a[4], a[5]
        out.copy_(out + x.sum(0) if acc else x.sum(0))
    elif fname == "mdnvae_adam_step":
        p, g, m, v, n, step, lr, b1, b2, eps, scale = a[:11]
        g = g * scale
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        p.sub_(lr / (1 - b1 ** step) * m / (v.sqrt() / (1 - b2 ** step) ** 0.5 + eps))
    elif fname == "mdnvae_adam_step_folded":
        p, gt, m, v, H, n, step, coef, lr, b1, b2, eps, scale = a[:13]
        assert coef is None
        i, j = _tri_index(n)
        T = i.numel()
        g = gt[:T] * scale
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        upd = (lr / (1 - b1 ** step) * m / (v.sqrt() / (1 - b2 ** step) ** 0.5 + eps)).t()      # [H,T]
        p3 = p.view(H, n, n)
        p3[:, i, j] -= upd
        off = i != j
        p3[:, j[off], i[off]] -= upd[:, off]
    else:
        raise AssertionError(f"unexpected kernel call {fname}")


class _ToySym(torch.nn.Module):
    """One symmetric-input first layer (the product op) and a plain second layer."""

    def __init__(self, n, H):
        super().__init__()
        self.n_nodes, self.symmetric_input = n, True
        self.fc11 = torch.nn.Linear(n * n, H)
        self.fc2 = torch.nn.Linear(H, n * n)
        self.register_buffer("_asym", torch.zeros(1, dtype=torch.int32), persistent=False)

    def folded_grad_params(self):
        return [(self.fc11.weight, self.n_nodes)]

    def forward(self, x):
        from mdnvae_amd import ops
        xt = ops.tri_pack(x, self.n_nodes, self._asym)
        return self.fc2(ops.sym_linear_act(xt, self.fc11.weight, self.fc11.bias, "relu", self.n_nodes))


def _patch_kernels():
    from mdnvae_amd import ops, trainer
    for mod in (ops, trainer):
        mod.call = _fake_call
        mod.ptr = lambda t, *args, **kw: t
        mod.stream = lambda: None


def _symmetric_batch(B, n, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, n, n, generator=g)
    return (x + x.transpose(1, 2)).reshape(B, n * n).contiguous()


def _toy_trainer(n, H, seed=21, **kw):
    class CpuTrainer(M.Trainer):
        def _forward_backward(self, inputs, target, eps=None):        # torch loss in place of the HIP loss kernels
            self._folded.clear()
            self.buf.zero_grad(self._fold_end if self._defer_fold else 0)
            loss = ((self.model(*inputs) - target) ** 2).sum()
            loss.backward()
            return loss.detach()

    torch.manual_seed(seed)
    return CpuTrainer(_ToySym(n, H), "mse", **kw)


FOLD_CASES = {
    "hooked": dict(overlap_floats=100),                       # folded gradient reduced early, from backward
    "late": dict(overlap_floats=0),                           # no hooks: folded gradient summed where Adam consumes it
    "hooked_unfolded_adam": dict(overlap_floats=100, folded_adam=False),
    "small_layer": dict(overlap_floats=1 << 20, folded_adam=False),   # below the early size: unfolded at once, goes with the buckets
}


def _fold_worker(rank, world, port, case, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _patch_kernels()
    n, H, B = 6, 8, 10
    # every rank initialises from its own seed: the trainer's broadcast makes the replicas rank 0's model
    tr = _toy_trainer(n, H, seed=21 + 7 * rank, **FOLD_CASES[case])
    torch.manual_seed(21)
    first = _ToySym(n, H)
    assert all(torch.equal(a, b) for a, b in zip(tr.model.state_dict().values(), first.state_dict().values()))
    if case in ("hooked", "late"):
        assert tr._fold_end > 0 and all(lo >= tr._fold_end for lo, _ in tr._rest)   # the unfolded region is never exchanged
    x = _symmetric_batch(B, n, 5)
    lo, hi = M.data.shard_range(B, rank, world)
    for step in range(3):
        xb = x[lo:hi] * (1.0 + 0.1 * step)
        tr.step((xb,), xb)
    # the case ran the path it is named for
    folded_adam = _CALLS.get("mdnvae_adam_step_folded", 0)
    unfolds = _CALLS.get("mdnvae_tri_unfold", 0)
    assert folded_adam == (3 if case in ("hooked", "late") else 0) and unfolds == (0 if case in ("hooked", "late") else 3)
    assert bool(tr._hook_of) == (case in ("hooked", "hooked_unfolded_adam"))
    out[rank] = {k: v.detach().clone() for k, v in tr.model.state_dict().items()}
    dist.destroy_process_group()


@pytest.mark.parametrize("case", list(FOLD_CASES))
def test_two_rank_folded_gradient_step_equals_single_process_adam(case):
    world = 2
    out = mp.Manager().dict()
    mp.spawn(_fold_worker, args=(world, _free_port(), case, out), nprocs=world, join=True)
    # single process, whole batch, plain torch layers and torch.optim.Adam
    n, H, B = 6, 8, 10
    torch.manual_seed(21)
    ref = _ToySym(n, H)
    net = torch.nn.Sequential(ref.fc11, torch.nn.ReLU(), ref.fc2)
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    x = _symmetric_batch(B, n, 5)
    for step in range(3):
        xb = x * (1.0 + 0.1 * step)
        opt.zero_grad()
        ((net(xb) - xb) ** 2).sum().backward()
        opt.step()
    want = {"fc11.weight": ref.fc11.weight, "fc11.bias": ref.fc11.bias, "fc2.weight": ref.fc2.weight, "fc2.bias": ref.fc2.bias}
    for r in range(world):
        for k, v in want.items():
            assert torch.allclose(out[r][k], v.detach(), rtol=2e-5, atol=2e-6), (case, r, k, float((out[r][k] - v).abs().max()))
    for k in want:
        assert torch.equal(out[0][k], out[1][k])                  # replicas stay bit-identical
