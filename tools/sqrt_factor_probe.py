"""Trace error and iteration counts of sqrt_factor under the rigorous and the power-iteration scale."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mdnvae_amd as M
from mdnvae_amd import fid

def psd(d, floor, rng):
    q, _ = np.linalg.qr(rng.standard_normal((d, d)))
    a = (q * np.exp(rng.uniform(np.log(floor), 0.0, d))) @ q.T
    return (a + a.T) / 2

orig = fid._scaled
for d, floor in ((1000, 1e-3), (1008, 1e-3), (640, 0.05), (1000, 1e-2)):
    a = psd(d, floor, np.random.default_rng(d))
    want = np.sqrt(np.clip(np.linalg.eigvalsh(a), 0, None)).sum()
    for tight, stop in ((False, 2e-6), (True, 2e-6), (True, 1e-4), (True, 1e-3), (True, 1e-2)):
        fid._FAST_STOP = stop
        fid._scaled = lambda m, who, t=True, _tight=tight: orig(m, who, _tight)
        f, iters = M.sqrt_factor(torch.from_numpy(a).cuda())
        got = f.cpu().numpy()
        print(d, floor, "tight" if tight else "loose", stop, iters, abs(np.trace(got) - want) / want,
              np.linalg.norm(got @ got.T - a) / np.linalg.norm(a), flush=True)
