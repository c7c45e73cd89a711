"""Largest rstd per normalised layer of the conv encoder on a fixture (near-dead planes amplify rounding)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import mdnvae_amd as M
from mdnvae_amd import ops
from conftest import load_fixture, fixture_params
from test_conv_gpu import VARIANT_CTOR
variant = sys.argv[1]
fix = load_fixture(variant)
DEV = "cuda:0"
x = torch.from_numpy(fix["x"]).to(DEV); eps = torch.from_numpy(fix["eps"]).to(DEV)
orig = ops._cb_norm_fwd
def spy(y, stats, *a, **kw):
    out = orig(y, stats, *a, **kw)
    st = out[3].view(-1, 2)
    r = st[:, 1]
    print(f"  layer planes {r.numel()}: rstd max {float(r.max()):.3g} median {float(r.median()):.3g}  planes with rstd > 50: {int((r > 50).sum())}  |y|max {float(y.float().abs().max()):.3g}")
    return out
ops._cb_norm_fwd = spy
model = M.ConvVAE(precision="fp32", **VARIANT_CTOR[variant]).eval()
model.load_state_dict(fixture_params(fix), strict=False)
model = model.to(DEV)
with torch.no_grad():
    model(x, eps=eps)
