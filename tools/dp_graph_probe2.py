"""Two gloo ranks, split-graph replay: before each replay an eager forward/backward on the same inputs; compares the
losses and the gradient buffers the two leave."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import mdnvae_amd as M
import test_zgloo2_gpu as T


def worker(rank, world, port, kind):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = M.data.shard_range(T._batch(kind), rank, world)
    model, adjacency, x, eps, traits = T._setup(kind, 5 + 100 * rank)
    trainer = M.Trainer(model, "mdn")
    xb = x[lo:hi] * 1.0
    inputs = (xb, adjacency, traits[lo:hi])
    e = eps[0, lo:hi]
    trainer.capture(inputs, xb, e)

    def replay_a(tag):
        trainer._graph.replay()
        torch.cuda.synchronize()
        print(f"rank {rank} {tag}: graph-A loss {float(trainer._g_loss.item())}", flush=True)

    replay_a("after capture")
    replay_a("again")
    trainer._split_graph = True
    le = float(trainer.forward_backward(inputs, xb, eps=e, _defer_fold=True).item())
    trainer._split_graph = False
    trainer._folded.clear()
    print(f"rank {rank} eager without exchange: {le}", flush=True)
    replay_a("after eager without exchange")
    le = float(trainer.forward_backward(inputs, xb, eps=e, _defer_fold=True).item())
    for w in trainer._pending:
        w.wait()
    trainer._pending.clear()
    for work, gt, _, _ in trainer._folded:
        if work not in (None, "late"):
            work.wait()
    trainer._folded.clear()
    print(f"rank {rank} eager with exchange: {le}", flush=True)
    replay_a("after eager with exchange")
    with torch.no_grad():
        out = model(*inputs, eps=e, nodes_only=True)
    replay_a("after eager forward only")
    dist.destroy_process_group()


if __name__ == "__main__":
    mp.spawn(worker, args=(2, T._free_port(), sys.argv[1] if len(sys.argv) > 1 else "mdn_split"), nprocs=2, join=True)
