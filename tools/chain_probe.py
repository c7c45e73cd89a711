"""Time the chain decoder's node-vector forward (and backward) alone at the headline size."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import mdnvae_amd as M
DEV = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
n = 87
dec = M.MultiChannelGCNDecoder(R=5, n_nodes=n, k=32, n_layers=3, latent_dim=68).to(DEV)
adjacency = [[(u + 3 * j) % n for j in range(32)] for u in range(n)]
z = torch.randn(B, 68, device=DEV, requires_grad=True)
cot = torch.randn(B, 5, 96, device=DEV)


def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


with torch.no_grad():
    print(f"forward, nothing saved     {timed(lambda: dec.node_vectors(z, adjacency)):7.1f} us (includes the panel packing launch)")
print(f"forward with saved inputs  {timed(lambda: dec.node_vectors(z, adjacency)):7.1f} us")


def fb():
    z.grad = None
    (dec.node_vectors(z, adjacency) * cot).sum().backward()


print(f"forward + backward         {timed(fb):7.1f} us (with the torch product and sum)")
