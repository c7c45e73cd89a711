"""Two data-parallel ranks with the real kernels.  One GPU is all a test box has and RCCL refuses two
ranks on one device, so the two ranks share the GPU and exchange through gloo; the trainer stages
device buffers through pinned host memory for that backend.  What runs is the multi-GPU step
itself: subject shards, folded first-layer gradients crossing between ranks at half size (early from
backward, or late), bucketed SUM all-reduce of the rest, folded Adam on the summed gradient - and the
result must be the single-process step on the whole batch."""
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


def _setup(kind):
    split = kind == "mdn_split"
    if split:
        kind = "mdn"
    B = _batch("mdn_split" if split else kind)
    fix = load_fixture("mm_gate")
    params = fixture_params(fix)
    adjacency = [list(map(int, r)) for r in fix["adjacency"]]
    torch.manual_seed(5)                          # the gate layer is not in the fixture: same init everywhere
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


def _train(kind, lo, hi, **trainer_kw):
    model, adjacency, x, eps, traits = _setup(kind)
    kind = "mdn" if kind == "mdn_split" else kind
    trainer = M.Trainer(model, "mdn" if kind == "mdn" else "poisson_log", **trainer_kw)
    losses = []
    # the question here is whether two ranks take the single-process step, so the first-layer weight
    # gradients run with all six part products: the three-term default leaves 1e-5 of rounding in them,
    # which differs between one product over the whole batch and two over its halves, and Adam's first
    # steps turn any difference in a near-zero gradient entry into a move of +-lr
    terms, M.ops.SPLIT_WGRAD_TERMS = M.ops.SPLIT_WGRAD_TERMS, 6
    try:
        for s in range(STEPS):
            xb = x[lo:hi] * (1.0 + 0.05 * s)
            extra = (adjacency, traits[lo:hi]) if kind == "mdn" else (adjacency,)
            losses.append(float(trainer.step((xb,) + extra, xb, eps=eps[s, lo:hi]).item()))
    finally:
        M.ops.SPLIT_WGRAD_TERMS = terms
    model.check_inputs()
    torch.cuda.synchronize()
    return losses, {k: v.detach().cpu().clone() for k, v in model.named_parameters()}, trainer


def _worker(rank, world, port, kind, overlap, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = M.data.shard_range(_batch(kind), rank, world)
    losses, params, trainer = _train(kind, lo, hi, **({} if overlap else {"overlap_floats": 0}))
    assert bool(trainer._hook_of) == overlap
    assert all(a >= trainer._fold_end for a, _ in trainer._rest)      # unfolded first-layer gradients never travel
    out[rank] = (losses, params)
    dist.destroy_process_group()


@pytest.mark.parametrize("kind,overlap", [("mdn", True), ("mdn", False), ("mm", True), ("mdn_split", True)])
def test_two_ranks_on_one_gpu_take_the_single_process_step(kind, overlap):
    world = 2
    out = mp.Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), kind, overlap, out), nprocs=world, join=True)
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
