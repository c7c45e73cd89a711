"""The data-parallel step on the real RCCL backend with a one-rank group (one GPU is all there is):
every collective is an identity, but the calls - communicator set-up on the device, asynchronous
all-reduce of flat-buffer views launched from inside backward, bucketed all-reduce, waits - are the
ones a multi-GPU run makes."""
import socket

import pytest
import torch
import torch.distributed as dist

from conftest import load_fixture, fixture_params
import mdnvae_amd as M
from mdnvae_amd import trainer as T

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_one_rank_rccl_step_equals_the_local_step():
    fix = load_fixture("mm_gate")
    params = fixture_params(fix)
    x = torch.from_numpy(fix["x"]).to(DEV)
    x = 0.5 * (x.reshape(-1, 68, 68) + x.reshape(-1, 68, 68).transpose(1, 2)).reshape(-1, 68 * 68).contiguous()
    eps = torch.from_numpy(fix["eps"]).to(DEV)
    adjacency = [list(map(int, r)) for r in fix["adjacency"]]

    def run(distributed, overlap=True, graph=False):
        model = M.GCNVAE(R=5, n_nodes=68, k=32, n_layers=3, latent_dim=68, symmetric_input=True)
        model.load_state_dict(params, strict=False)
        trainer = M.Trainer(model.to(DEV), "poisson_log", **({} if overlap else {"overlap_floats": 0}))
        assert bool(trainer._hook_of) == (distributed and overlap)   # the big first-layer matrices reduce early
        losses = []
        if graph:
            # the captured step of a data-parallel trainer: backward graph, exchange, Adam graph
            trainer.capture((x, adjacency), x, eps)
            assert (trainer._graph_adam is not None) == distributed
            losses = [float(trainer.replay((x, adjacency), x, eps).item()) for _ in range(3)]
            torch.cuda.synchronize()
            return losses, {k: v.detach().clone() for k, v in model.named_parameters()}
        for _ in range(3):
            loss = trainer.forward_backward((x, adjacency), x, eps=eps)
            # both first-layer weight gradients travel folded (half size), unfolded after the sum
            assert len(trainer._folded) == 2
            assert all((work is not None) == distributed for work, _, _, _ in trainer._folded)
            assert all(work == "late" for work, _, _, _ in trainer._folded) == (distributed and not overlap)
            assert all(gt.shape == (M.ops.tri_row_stride(68), 1024) for _, gt, _, _ in trainer._folded)
            trainer.optimizer_step()
            assert not trainer._folded and not trainer._pending
            losses.append(float(loss.item()))
        torch.cuda.synchronize()
        if distributed and overlap:                           # big matrices first, the rest is one range
            assert trainer._rest == [(2 * 1024 * 68 * 68, trainer.buf.numel)]
        return losses, {k: v.detach().clone() for k, v in model.named_parameters()}

    local = run(False)
    local_graph = run(False, graph=True)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", world_size=1, rank=0,
                            device_id=torch.device(DEV))
    T.MIN_RANKS_FOR_EXCHANGE = 1
    try:
        dist.barrier()
        exchanged = run(True)
        late = run(True, overlap=False)                       # everything reduced after backward
        replayed = run(True, graph=True)
        stats = M.FeatureStats(8, torch.zeros(8, device=DEV), DEV).update(torch.rand(64, 8, device=DEV)).allreduce()
        assert stats.n == 64
    finally:
        T.MIN_RANKS_FOR_EXCHANGE = 2
        dist.destroy_process_group()
    assert exchanged[0] == local[0]
    assert late[0] == local[0]
    assert replayed[0] == local_graph[0]
    for name, value in local[1].items():
        assert torch.equal(exchanged[1][name], value), name
        assert torch.equal(late[1][name], value), name
        assert torch.equal(replayed[1][name], local_graph[1][name]), name
