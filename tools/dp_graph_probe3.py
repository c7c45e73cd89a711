"""Two gloo ranks: per step, loss and checksums of parameters / Adam moments, eager vs split-graph replay."""
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


def worker(rank, world, port, kind, graph):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = M.data.shard_range(T._batch(kind), rank, world)
    model, adjacency, x, eps, traits = T._setup(kind, 5 + 100 * rank)
    tr = M.Trainer(model, "mdn")
    cs = lambda t: float(t.double().abs().sum())
    for s in range(2):
        xb = x[lo:hi] * (1.0 + 0.05 * s)
        inputs = (xb, adjacency, traits[lo:hi])
        e = eps[s, lo:hi]
        if graph and s == 0:
            tr.capture(inputs, xb, e)
        loss = float((tr.replay if graph else tr.step)(inputs, xb, e).item())
        torch.cuda.synchronize()
        fm = [cs(v) for v in tr._fold_m.values()]
        print(f"graph {graph} rank {rank} step {s}: loss {loss} p {cs(tr.buf.flat_p[tr._fold_end:]):.6f} w11 {cs(model.fc11.weight):.6f} "
              f"m {cs(tr.exp_avg[tr._fold_end:]):.6e} v {cs(tr.exp_avg_sq[tr._fold_end:]):.6e} fold_m {fm[0]:.6e} g {cs(tr.buf.flat_g[tr._fold_end:]):.6e}", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    for graph in (False, True):
        mp.spawn(worker, args=(2, T._free_port(), "mdn_split", graph), nprocs=2, join=True)
