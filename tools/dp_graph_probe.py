"""Two gloo ranks on one GPU: eager DP step vs split-graph replay, per-rank losses and parameter differences."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import torch.multiprocessing as mp
import test_zgloo2_gpu as T


def worker(rank, world, port, kind, overlap, graph, steps, out):
    T.STEPS = steps
    T._worker(rank, world, port, kind, overlap, graph, out)


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "mdn_split"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    res = {}
    for graph in (False, True):
        out = mp.Manager().dict()
        mp.spawn(worker, args=(2, T._free_port(), kind, True, graph, steps, out), nprocs=2, join=True)
        res[graph] = {r: out[r] for r in range(2)}
        print("graph", graph, "losses", [res[graph][r][0] for r in range(2)])
    for name in res[False][0][1]:
        e0, e1, g0, g1 = (res[gr][r][1][name] for gr in (False, True) for r in (0, 1))
        d = [float((a - b).abs().max()) for a, b in ((e0, e1), (g0, g1), (e0, g0), (e1, g1))]
        if max(d) > 0:
            print(f"  {name:40s} eager r0-r1 {d[0]:.2e}  graph r0-r1 {d[1]:.2e}  r0 eager-graph {d[2]:.2e}  r1 eager-graph {d[3]:.2e}")


if __name__ == "__main__":
    main()
