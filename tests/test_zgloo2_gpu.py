"""Two data-parallel ranks with the real kernels.  One GPU is all a test box has and RCCL refuses two
ranks on one device, so the two ranks share the GPU and exchange through gloo; the trainer stages
device buffers through pinned host memory for that backend.  What runs is the multi-GPU step
itself: subject shards, folded first-layer gradients crossing between ranks at half size (early from
backward, or late), bucketed SUM all-reduce of the rest, folded Adam on the summed gradient - and the
result must be the single-process step on the whole batch.  Everything runs at the default arithmetic
(first-layer products from two fp16 parts); the second rank draws its initial gate layer from another seed,
so equal replicas also show that the trainer broadcast rank 0's parameters."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import load_fixture, fixture_params
import mdnvae_amd as M

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
B, STEPS = 48, 2
B_SPLIT = 2048          # 1024 subjects per rank: the first layers take the split-bf16 product


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _batch(kind):
    return B_SPLIT if kind == "mdn_split" else B


def _setup(kind, seed=5):
    split = kind == "mdn_split"
    if split:
        kind = "mdn"
    B = _batch("mdn_split" if split else kind)
    fix = load_fixture("mm_gate")
    params = fixture_params(fix)
    adjacency = [list(map(int, r)) for r in fix["adjacency"]]
    torch.manual_seed(seed)                       # the gate layer is not in the fixture
    if kind == "mdn":
        model = M.MDNGateVAE(R=5, n_nodes=68, k=32, n_layers=3, latent_dim=68, n_traits=4, symmetric_input=True)
    else:
        model = M.GCNVAE(R=5, n_nodes=68, k=32, n_layers=3, latent_dim=68, symmetric_input=True)
    model.load_state_dict(params, strict=False)
    if kind == "mdn":
        with torch.no_grad():
            model.log_sigma.fill_(1.0)
    x = M.data.synthetic_sc(B, 68, seed=77, device=DEV)
    g = torch.Generator().manual_seed(9)
    eps = torch.randn(STEPS, B, 68, generator=g).to(DEV)
    traits = torch.randn(B, 4, generator=g).to(DEV)
    return model.to(DEV), adjacency, x, eps, traits


def _train(kind, lo, hi, seed=5, graph=False, **trainer_kw):
    model, adjacency, x, eps, traits = _setup(kind, seed)
    kind = "mdn" if kind == "mdn_split" else kind
    trainer = M.Trainer(model, "mdn" if kind == "mdn" else "poisson_log", **trainer_kw)
    losses = []
    for s in range(STEPS):
        xb = x[lo:hi] * (1.0 + 0.05 * s)
        extra = (adjacency, traits[lo:hi]) if kind == "mdn" else (adjacency,)
        if graph and s == 0:
            trainer.capture((xb,) + extra, xb, eps[s, lo:hi])
        step = trainer.replay if graph else trainer.step
        losses.append(float(step((xb,) + extra, xb, eps[s, lo:hi]).item()))
    model.check_inputs()
    torch.cuda.synchronize()
    return losses, {k: v.detach().cpu().clone() for k, v in model.named_parameters()}, trainer


def _worker(rank, world, port, kind, overlap, graph, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = M.data.shard_range(_batch(kind), rank, world)
    losses, params, trainer = _train(kind, lo, hi, seed=5 + 100 * rank, graph=graph,
                                     **({} if overlap else {"overlap_floats": 0}))
    assert (trainer._graph_adam is not None) if graph else True        # the captured step is split at the exchange
    assert bool(trainer._hook_of) == overlap
    assert all(a >= trainer._fold_end for a, _ in trainer._rest)      # unfolded first-layer gradients never travel
    out[rank] = (losses, params)
    dist.destroy_process_group()


@pytest.mark.parametrize("kind,overlap,graph", [("mdn", True, False), ("mdn", False, False), ("mm", True, False),
                                                ("mdn_split", True, False), ("mdn", True, True), ("mdn_split", True, True)])
def test_two_ranks_on_one_gpu_take_the_single_process_step(kind, overlap, graph):
    """graph: each rank replays its step as two captured HIP graphs (backward / Adam) with the gradient exchange between them; the reference stays the eager single-process step."""
    world = 2
    out = mp.Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), kind, overlap, graph, out), nprocs=world, join=True)
    want_losses, want, _ = _train(kind, 0, _batch(kind))
    for s in range(STEPS):                                   # sum-reduced loss: the shards' losses add up
        got = out[0][0][s] + out[1][0][s]
        assert abs(got - want_losses[s]) <= 1e-5 * abs(want_losses[s])
    for name, value in want.items():
        assert torch.equal(out[0][1][name], out[1][1][name]), name         # replicas stay identical
        # Adam divides by sqrt(v): an entry whose gradient is near rounding level moves by a visible
        # fraction of lr either way (the two runs add the same numbers in different orders).  So: the
        # typical entry agrees to fp32 rounding, hardly any entry is off by more than 1 % of lr, and
        # none by more than a quarter of the largest possible move (STEPS * lr)
        diff = (out[0][1][name] - value).abs()
        assert float(diff.mean()) <= 2e-7, (name, float(diff.mean()))
        assert float((diff > 1e-5).float().mean()) <= 1e-3, (name, float((diff > 1e-5).float().mean()))
        assert float(diff.max()) <= 0.25 * STEPS * 1e-3, (name, float(diff.max()))
