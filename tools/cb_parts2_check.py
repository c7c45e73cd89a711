"""Gradients of the conv VAE per precision mode against the fixture's float64 entries (kept entries), layer by layer."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch
import mdnvae_amd as M
import recipe
from conftest import load_fixture, fixture_params
from oracle import ref_eager as R
from test_conv_gpu import VARIANT_CTOR
variant = sys.argv[1] if len(sys.argv) > 1 else "cnn_nn_sc"
fix = load_fixture(variant)
DEV = "cuda:0"
x = torch.from_numpy(fix["x"]).to(DEV); eps = torch.from_numpy(fix["eps"]).to(DEV)
res = {}
for precision in ("fp32_mfma", "fp32_planar", "fp32"):
    model = M.ConvVAE(precision=precision, **VARIANT_CTOR[variant]).eval()
    model.load_state_dict(fixture_params(fix), strict=False)
    model = model.to(DEV)
    if "mask" in fix.files:
        model.set_mask([torch.from_numpy(fix["mask"])] * 5 + [torch.eye(68 * 68)])
    recon, mu, logvar, x_latent = model(x, eps=eps)
    loss = M.vae_loss(recon, x, mu, logvar, R.loss_mode(variant))
    loss.backward()
    res[precision] = {n: p.grad.detach().double().cpu().reshape(-1).numpy() for n, p in model.named_parameters()}
    print(precision, "loss", loss.item(), float(fix["loss"]))
for name in res["fp32"]:
    if not name.startswith("ec.") or name.endswith("bias"):
        continue
    full = f"hfull/{name}" in fix.files
    key = f"hfull/{name}" if full else f"hsamp/{name}"
    if key not in fix.files:
        continue
    ref = fix[key].astype(np.float64)
    ref32 = fix[("gfull/" if full else "gsamp/") + name].astype(np.float64)
    idx = None if full else recipe.sample_index(name, res["fp32"][name].size)
    out = [f"{name:16s} ref-fp32 noise {np.linalg.norm(ref32 - ref) / np.linalg.norm(ref):.2e}"]
    for precision in res:
        got = res[precision][name] if full else res[precision][name][idx]
        out.append(f"{precision} {np.linalg.norm(got - ref) / np.linalg.norm(ref):.2e}")
    print("  ".join(out))
