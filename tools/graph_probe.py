"""Times eager vs graph-replayed steps of a conv-encoder VAE at a small batch."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mdnvae_amd as M
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda", 0)
for graph in (False, True):
    el, loss, _, _, what = bench.run_workload("cnn_gcn", B, 87, 20, 5, 2, dev, 0, 1, False, graph=graph)
    print("graph" if graph else "eager", f"{el / 20 * 1e3:.3f} ms/step  {B * 20 / el:.0f} samples/s  loss {loss / B:.3f}")
