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
