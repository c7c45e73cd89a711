#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the reference's own
class definitions.  Runs ONLY in the build container (it reads /root/reference);
nothing under tests/, bench.py or smoke() imports this file.

Method: parse a reference script, keep its top-level class / function
definitions (the scripts themselves train at import time and load files that
are not in the mount), execute those definitions in a namespace that provides
the names they close over, fill every parameter with the seeded recipe of
recipe.py, run forward + loss + backward in fp32 (and fp64 for the loss and the
gradient samples), and store inputs and expected outputs as .npz.

    python tests/golden/make_goldens.py            # all fixtures
    python tests/golden/make_goldens.py mm_gate    # one fixture
"""
import ast
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn, optim
from torch.autograd import Variable
from scipy.io import loadmat
from scipy.linalg import sqrtm
import networkx as nx

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import recipe  # noqa: E402

REF = "/root/reference"
B = 4
EPS_SEED = 4242


def load_defs(relpath, extra=None):
    src = open(os.path.join(REF, relpath)).read()
    tree = ast.parse(src)
    keep = [n for n in tree.body if isinstance(n, (ast.ClassDef, ast.FunctionDef))]
    mod = ast.Module(body=keep, type_ignores=[])
    ns = dict(torch=torch, F=F, nn=nn, optim=optim, np=np, Variable=Variable, sqrtm=sqrtm, nx=nx,
              n_size=32, hidden_dim=1024)
    if extra:
        ns.update(extra)
    exec(compile(mod, relpath, "exec"), ns)
    return ns


def sc_inputs():
    """Structural-connectome rows and A_mat per function/VAE_CNN_NN.py:17-41."""
    tensor = loadmat(os.path.join(REF, "data/HCP_samples.mat"))["loaded_tensor_sub"]
    A_mat = np.mean(np.squeeze(tensor[18:86, 18:86, 1, :]), axis=2)
    A_mat = A_mat + A_mat.transpose()
    rows, raw = [], []
    for i in range(tensor.shape[3]):
        ith = np.float32(tensor[:, :, 0, i] + np.transpose(tensor[:, :, 0, i]))
        np.fill_diagonal(ith, np.mean(ith, 0))
        raw.append(ith.copy())
        rows.append((ith[18:86, 18:86].flatten()) / 200)
    return np.array(rows, dtype=np.float32), A_mat, np.array(raw, dtype=np.float32), tensor


def fc_inputs():
    """Functional-connectome rows: the ten preprocessed 68x68 matrices the reference
    holds as result/sample_i_real.txt (written by function_FC/VAE_CNN_NN.py:402-412)."""
    mats = [np.loadtxt(os.path.join(REF, f"result/sample_{i}_real.txt")) for i in range(10)]
    arr = np.array(mats, dtype=np.float32)
    return arr.reshape(10, -1), arr.mean(axis=0).astype(np.float64)


def refill(model):
    """Overwrite every parameter by the seeded recipe; returns shapes and bounds."""
    shapes, bounds = {}, {}
    with torch.no_grad():
        for name, p in model.named_parameters():
            b = float(p.abs().max())
            if b == 0.0:
                b = 0.05
            shapes[name] = tuple(p.shape)
            bounds[name] = b
            p.copy_(recipe.fill_param(name, p.shape, b))
    return shapes, bounds


def run_with_eps(model, args, seed):
    torch.manual_seed(seed)
    return model(*args)


def draw_eps(seed, shape, dtype=torch.float32):
    torch.manual_seed(seed)
    return torch.randn(shape, dtype=dtype)


def meta_arrays(shapes, bounds):
    names = sorted(shapes)
    return {
        "param_names": np.array(names),
        "param_shapes": np.array([",".join(map(str, shapes[n])) for n in names]),
        "param_bounds": np.array([bounds[n] for n in names], dtype=np.float64),
    }


def grads_of(model, prefix="", tag="g"):
    out = {}
    for name, p in model.named_parameters():
        out.update(recipe.pack_grad(prefix + name, p.grad, tag))
    return out


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"  wrote {path}  ({os.path.getsize(path) / 1024:.0f} KiB)")


def conv_vae_fixture(tag, relpath, x, masks=None, eval_mode=False):
    """Fixtures for the conv-encoder VAE scripts (with or without masks)."""
    ns = load_defs(relpath)
    torch.manual_seed(11111)
    model = ns["VAE"]()
    if eval_mode:
        model.eval()
    shapes, bounds = refill(model)
    if masks is not None:
        model.set_mask([torch.from_numpy(np.float32(m)) for m in masks])
    xt = torch.from_numpy(x)
    L = (model.fc_mean if hasattr(model, "fc_mean") else model.fc21).out_features
    eps = draw_eps(EPS_SEED, (x.shape[0], L))
    recon, mu, logvar, x_latent = run_with_eps(model, (xt,), EPS_SEED)
    loss = ns["loss_function"](recon, xt, mu, logvar)
    model.zero_grad()
    loss.backward()
    z = mu + eps * torch.exp(0.5 * logvar)
    arrays = dict(x=x, eps=eps.numpy(), mu=mu.detach().numpy(), logvar=logvar.detach().numpy(),
                  z=z.detach().numpy(), recon=recon.detach().numpy(),
                  x_latent=x_latent.detach().numpy(), loss=np.float64(loss.item()))
    arrays.update(meta_arrays(shapes, bounds))
    arrays.update(grads_of(model))
    # the weights actually in effect after set_mask (masked entries zeroed) are what the
    # recipe gives times the mask, so nothing else needs storing.
    # fp64 pass for the loss (noise-floor reference)
    model64 = ns["VAE"]().double()
    if eval_mode:
        model64.eval()
    with torch.no_grad():
        for n, p in model64.named_parameters():
            p.copy_(recipe.fill_param(n, p.shape, bounds[n], torch.float64))
    if masks is not None:
        model64.set_mask([torch.from_numpy(np.float64(m)) for m in masks])
    r64, m64, lv64, _ = run_with_eps(model64, (xt.double(),), EPS_SEED)
    # randn_like on a float64 tensor draws different numbers: overwrite z path by recomputing
    z64 = m64 + eps.double() * torch.exp(0.5 * lv64)
    r64, _ = model64.decode(z64)
    loss64 = ns["loss_function"](r64.view(x.shape[0], -1), xt.double(), m64, lv64)
    arrays["loss64"] = np.float64(loss64.item())
    model64.zero_grad()
    loss64.backward()
    arrays.update(grads_of(model64, tag="h"))
    if masks is not None:
        arrays["mask"] = np.float32(masks[0])
    save(tag, **arrays)


def rank_mask(A_mat, n_size=32):
    # function/VAE-CNN-GCN.py:209 (identical expression in function_FC/VAE_CNN_GCN.py:216)
    return (np.argsort(np.argsort(A_mat, axis=-1), axis=-1) < n_size + 1)


def make_cnn_nn_sc():
    x, _, _, _ = sc_inputs()
    conv_vae_fixture("cnn_nn_sc", "function/VAE_CNN_NN.py", x[:B])


def make_cnn_gcn_sc():
    x, A_mat, _, _ = sc_inputs()
    m = rank_mask(A_mat)
    conv_vae_fixture("cnn_gcn_sc", "function/VAE-CNN-GCN.py", x[:B],
                     masks=[m] * 5 + [np.identity(68 * 68)])


def make_cnn_gcn_fc():
    x, A_mat = fc_inputs()
    m = rank_mask(A_mat)
    conv_vae_fixture("cnn_gcn_fc", "function_FC/VAE_CNN_GCN.py", x[:B],
                     masks=[m] * 5 + [np.identity(68 * 68)])


def make_cnn_nn_fc():
    x, _ = fc_inputs()
    conv_vae_fixture("cnn_nn_fc", "function_FC/VAE_CNN_NN.py", x[:B], eval_mode=True)


def make_fnn_gcn_fc():
    """function_FC/MM_Gate_test.py: FNN towers + tanh masked decoder + identity-masked intercept."""
    x, A_mat = fc_inputs()
    m = rank_mask(A_mat)
    conv_vae_fixture("fnn_gcn_fc", "function_FC/MM_Gate_test.py", x[:B],
                     masks=[m] * 5 + [np.identity(68 * 68)])


def make_mm_gate():
    x, A_mat, _, _ = sc_inputs()
    ns = load_defs("function/MM_Gate.py")
    adjacency = ns["build_knn_adjacency"](A_mat, k=32)
    adjacency = [list(map(int, a)) for a in adjacency]
    torch.manual_seed(11111)
    model = ns["VAE"](R=5, n_nodes=68, k=32, n_layers=3, latent_dim=68)
    shapes, bounds = refill(model)
    xt = torch.from_numpy(x[:B])
    eps = draw_eps(EPS_SEED, (B, 68))
    recon, mu, logvar = run_with_eps(model, (xt, adjacency), EPS_SEED)
    loss = ns["loss_function"](recon, xt, mu, logvar)
    model.zero_grad()
    loss.backward()
    z = mu + eps * torch.exp(0.5 * logvar)
    arrays = dict(x=x[:B], eps=eps.numpy(), A_mat=A_mat, adjacency=np.array(adjacency, dtype=np.int64),
                  mu=mu.detach().numpy(), logvar=logvar.detach().numpy(), z=z.detach().numpy(),
                  recon=recon.detach().numpy(), loss=np.float64(loss.item()))
    arrays.update(meta_arrays(shapes, bounds))
    arrays.update(grads_of(model))
    save("mm_gate", **arrays)


def make_cvae():
    fc, _ = fc_inputs()
    sc, _, _, _ = sc_inputs()
    fc = fc[:B]
    sc = (sc[:B] / sc[:B].max()).astype(np.float32)
    ns = load_defs("function_FC_SC/SC-FC.py", extra=dict(os=os))
    torch.manual_seed(11111)
    model = ns["cVAE"](in_dim=68 * 68, latent_dim=64)
    shapes, bounds = refill(model)
    fct, sct = torch.from_numpy(fc), torch.from_numpy(sc)
    mask_idx = np.array([1, 3])
    fc_no = fct.clone()
    fc_no[torch.from_numpy(mask_idx)] = 0.0
    eps_a = draw_eps(EPS_SEED, (B, 64))
    eps_b = draw_eps(EPS_SEED + 1, (B, 64))
    # SC-FC.py:148-157 with the two epsilon draws pinned
    recon, mu, logvar = run_with_eps(model, (fct, sct), EPS_SEED)
    loss_with, rl, kl = ns["loss_function"](recon, sct, mu, logvar)
    recon_no, mu_no, logvar_no = run_with_eps(model, (fc_no, sct), EPS_SEED + 1)
    loss_no, _, _ = ns["loss_function"](recon_no, sct, mu_no, logvar_no)
    reg = torch.relu(loss_with - loss_no + 0.01)
    loss = loss_with + reg
    model.zero_grad()
    loss.backward()
    arrays = dict(fc=fc, sc=sc, mask_idx=mask_idx, eps_a=eps_a.numpy(), eps_b=eps_b.numpy(),
                  recon=recon.detach().numpy(), mu=mu.detach().numpy(), logvar=logvar.detach().numpy(),
                  recon_no=recon_no.detach().numpy(), loss_with=np.float64(loss_with.item()),
                  loss_no=np.float64(loss_no.item()), recon_loss=np.float64(rl.item()),
                  kld=np.float64(kl.item()), loss=np.float64(loss.item()))
    arrays.update(meta_arrays(shapes, bounds))
    arrays.update(grads_of(model))
    save("cvae", **arrays)


def make_ec87():
    """Encoder stack on un-cropped 87x87 inputs (the reference `ec` accepts them unchanged)."""
    _, _, raw, _ = sc_inputs()
    x = (raw[:2] / 200).astype(np.float32)              # [2,87,87]
    ns = load_defs("function/VAE_CNN_NN.py")
    torch.manual_seed(11111)
    model = ns["VAE"]()
    shapes, bounds = refill(model)
    xt = torch.from_numpy(x).view(2, 1, 87, 87)
    feat = model.max_pool(model.ec(xt)).reshape(2, -1)
    mu = model.fc_mean(feat)
    logvar = model.fc_log_var(feat)
    g = torch.Generator().manual_seed(99)
    cot_mu = torch.randn(mu.shape, generator=g)
    cot_lv = torch.randn(mu.shape, generator=g)
    model.zero_grad()
    ((mu * cot_mu).sum() + (logvar * cot_lv).sum()).backward()
    keep = {k: v for k, v in shapes.items() if k.startswith("ec.") or k.startswith("fc_mean") or k.startswith("fc_log_var")}
    arrays = dict(x=x, feat=feat.detach().numpy(), mu=mu.detach().numpy(), logvar=logvar.detach().numpy(),
                  cot_mu=cot_mu.numpy(), cot_lv=cot_lv.numpy())
    arrays.update(meta_arrays(keep, {k: bounds[k] for k in keep}))
    for name, p in model.named_parameters():
        if name in keep:
            arrays.update(recipe.pack_grad(name, p.grad))
    model64 = ns["VAE"]().double()
    with torch.no_grad():
        for n, p in model64.named_parameters():
            p.copy_(recipe.fill_param(n, p.shape, bounds[n], torch.float64))
    feat64 = model64.max_pool(model64.ec(xt.double())).reshape(2, -1)
    ((model64.fc_mean(feat64) * cot_mu.double()).sum() + (model64.fc_log_var(feat64) * cot_lv.double()).sum()).backward()
    for name, p in model64.named_parameters():
        if name in keep:
            arrays.update(recipe.pack_grad(name, p.grad, "h"))
    save("ec87", **arrays)


def make_losses():
    """loss_function of each flavour on small random inputs."""
    g = torch.Generator().manual_seed(5)
    Bq, D, L = 6, 4624, 68
    x_cnt = torch.poisson(torch.rand(Bq, D, generator=g) * 3, generator=g) / 2.0
    rate = torch.rand(Bq, D, generator=g) * 4 + 1e-3
    lograte = torch.randn(Bq, D, generator=g)
    x01 = torch.rand(Bq, D, generator=g)
    r01 = torch.rand(Bq, D, generator=g)
    mu = torch.randn(Bq, L, generator=g)
    logvar = torch.randn(Bq, L, generator=g) * 0.5
    out = dict(x_cnt=x_cnt.numpy(), rate=rate.numpy(), lograte=lograte.numpy(), x01=x01.numpy(),
               r01=r01.numpy(), mu=mu.numpy(), logvar=logvar.numpy())
    cases = [("poisson_rate", "function/VAE_CNN_NN.py", rate, x_cnt),
             ("poisson_log", "function/MM_Gate.py", lograte, x_cnt),
             ("mse", "function_FC/VAE_CNN_GCN.py", r01, x01)]
    for tag, rel, rec, tgt in cases:
        ns = load_defs(rel)
        for dt, suf in ((torch.float32, ""), (torch.float64, "64")):
            r = rec.to(dt).clone().requires_grad_(True)
            m = mu.to(dt).clone().requires_grad_(True)
            lv = logvar.to(dt).clone().requires_grad_(True)
            loss = ns["loss_function"](r, tgt.to(dt), m, lv)
            loss.backward()
            out[f"{tag}/loss{suf}"] = np.float64(loss.item())
            if suf == "":
                out[f"{tag}/grad_recon"] = r.grad.numpy()
                out[f"{tag}/grad_mu"] = m.grad.numpy()
                out[f"{tag}/grad_logvar"] = lv.grad.numpy()
    save("losses", **out)


def make_fid():
    """validation/FID.py end to end (its own seeds) + calculate_fid on random feature sets."""
    t0 = time.time()
    ns = load_defs("validation/FID.py")
    gen, feat, fid = ns["generate_sbm_graphs"], ns["compute_graph_features"], ns["calculate_fid"]
    pi = [0.3, 0.4, 0.3]
    P_real = [[0.8, 0.05, 0.02], [0.05, 0.7, 0.03], [0.02, 0.03, 0.6]]
    P_small = [[0.8, 0.10, 0.07], [0.10, 0.7, 0.08], [0.07, 0.08, 0.6]]
    P_large = [[0.8, 0.3, 0.3], [0.3, 0.7, 0.3], [0.3, 0.3, 0.6]]
    f_real = np.array([feat(m) for m in gen(50, 100, 3, pi, P_real, seed=42)])
    f_small = np.array([feat(m) for m in gen(50, 100, 3, pi, P_small, seed=43)])
    f_large = np.array([feat(m) for m in gen(50, 100, 3, pi, P_large, seed=44)])
    out = dict(f_real=f_real, f_small=f_small, f_large=f_large,
               fid_small=np.float64(fid(f_real, f_small)), fid_large=np.float64(fid(f_real, f_large)))
    print("  FID.py:", out["fid_small"], out["fid_large"], f"({time.time() - t0:.0f}s)")
    ns2 = load_defs("function/VAE_CNN_NN.py")
    rng = np.random.default_rng(7)
    for d in (10, 30):
        mix = rng.normal(size=(d, d)) / np.sqrt(d)
        a = rng.normal(size=(500, d)) @ mix + rng.normal(size=d)
        b = rng.normal(size=(400, d)) @ (mix + 0.1 * rng.normal(size=(d, d))) + rng.normal(size=d) * 1.1
        val = ns2["calculate_fid"](a.mean(0), np.cov(a, rowvar=False), b.mean(0), np.cov(b, rowvar=False))
        out[f"rand{d}/a"] = a
        out[f"rand{d}/b"] = b
        out[f"rand{d}/fid"] = np.float64(val)
    save("fid", **out)


def make_resnet_fc():
    """function_fc_new/ResNet.py VAE with the constructor arguments of its main() (:250-257)."""
    x, _ = fc_inputs()
    ns = load_defs("function_fc_new/ResNet.py", extra=dict(utils=None, loadmat=None))
    torch.manual_seed(11111)
    model = ns["VAE"](latent_dim=256, decode_dimen_number_mlp1=8, hidden_dim_mlp1=512, decode_dimen_number_mlp2=8,
                      hidden_dim_mlp2=256, decode_dimen_number_gcn=10)
    shapes, bounds = refill(model)
    xt = torch.from_numpy(x[:B])
    eps = draw_eps(EPS_SEED, (B, 256))
    recon, mu, logvar = run_with_eps(model, (xt,), EPS_SEED)
    loss = ns["loss_function"](recon, xt, mu, logvar)
    model.zero_grad()
    loss.backward()
    z = mu + eps * torch.exp(0.5 * logvar)
    arrays = dict(x=x[:B], eps=eps.numpy(), mu=mu.detach().numpy(), logvar=logvar.detach().numpy(),
                  z=z.detach().numpy(), recon=recon.detach().numpy(), loss=np.float64(loss.item()))
    arrays.update(meta_arrays(shapes, bounds))
    arrays.update(grads_of(model))
    save("resnet_fc", **arrays)


def make_graphs():
    """Graph features of validation/FID.py (edge 'distance' attached) and of function/VAE_CNN_NN.py
    (no 'distance' attribute: hop counts) on the SBM graphs of validation/FID.py (its own seeds, stored
    as packed bits), on four real connectomes and on a disconnected matrix."""
    t0 = time.time()
    ns = load_defs("validation/FID.py")
    ns2 = load_defs("function/VAE_CNN_NN.py")
    gen = ns["generate_sbm_graphs"]
    pi = [0.3, 0.4, 0.3]
    P = dict(real=[[0.8, 0.05, 0.02], [0.05, 0.7, 0.03], [0.02, 0.03, 0.6]],
             small=[[0.8, 0.10, 0.07], [0.10, 0.7, 0.08], [0.07, 0.08, 0.6]],
             large=[[0.8, 0.3, 0.3], [0.3, 0.7, 0.3], [0.3, 0.3, 0.6]])
    out = {}
    for (name, p), seed in zip(P.items(), (42, 43, 44)):
        graphs = np.array(gen(50, 100, 3, pi, p, seed=seed))
        out[f"sbm_{name}_bits"] = np.packbits(graphs.astype(np.uint8).reshape(50, -1), axis=1)
        out[f"sbm_{name}_features"] = np.array([ns["compute_graph_features"](m) for m in graphs])
    x, _, _, _ = sc_inputs()
    mats = x[:4].reshape(4, 68, 68).astype(np.float32)
    two = np.zeros((10, 10), dtype=np.float32)          # two components + an isolated node
    two[:4, :4] = np.float32([[0, 1, .5, 0], [1, 0, .25, .75], [.5, .25, 0, 0], [0, .75, 0, 0]])
    two[4:9, 4:9] = np.float32(np.triu(np.arange(25).reshape(5, 5) % 4, 1) / 4.0)
    two[4:9, 4:9] += two[4:9, 4:9].T
    out["connectomes"] = mats
    out["disconnected"] = two
    for tag, fn in (("fidpy", ns["compute_graph_features"]), ("script", ns2["compute_graph_features"])):
        out[f"connectome_features_{tag}"] = np.array([fn(m) for m in mats])
        out[f"disconnected_features_{tag}"] = np.array(fn(two))
    print(f"  graph features ({time.time() - t0:.0f}s)")
    save("graphs", **out)


def _run_reference_loop(relpath, names, inside=None):
    """Executes, unmodified, the data-preparation `for` loop of a reference script (the first
    top-level For statement, or the first one inside function `inside`) in the namespace `names`
    (which provides `tensor`, `net_data = []` and whatever constants the loop reads)."""
    tree = ast.parse(open(os.path.join(REF, relpath)).read())
    body = tree.body
    defs = [n for n in body if isinstance(n, ast.FunctionDef)]
    if inside is not None:
        body = [n for n in defs if n.name == inside][0].body
    loop = [n for n in body if isinstance(n, ast.For)][0]
    mod = ast.Module(body=[d for d in defs if d.name != inside] + [loop], type_ignores=[])
    ns = dict(np=np, torch=torch, nn=nn, F=F, optim=optim, Variable=Variable)
    ns.update(names)
    exec(compile(mod, relpath, "exec"), ns)
    return np.array(ns["net_data"], dtype=np.float32)


def make_prep():
    """Data preparation: adjacency list and rank mask for the real 100-subject sample, and the
    reference's own preprocessing loops run on 4 subjects (raw inputs + the rows they produce)."""
    _, A_mat, _, tensor = sc_inputs()
    ns = load_defs("function/MM_Gate.py")
    adjacency = np.array([list(map(int, a)) for a in ns["build_knn_adjacency"](A_mat, k=32)], dtype=np.int64)
    t4 = np.ascontiguousarray(tensor[:, :, :, :4]).astype(np.float64)
    rows_sc = _run_reference_loop("function/VAE_CNN_NN.py", dict(tensor=t4, offset=200, net_data=[]))
    rows_log = _run_reference_loop("function/Full_NN.py", dict(tensor=t4, net_data=[]))
    rng = np.random.RandomState(7)
    fc = []
    for _ in range(4):                      # correlation-like matrices, as the (absent) FC file holds
        u = rng.randn(87, 8)
        c = u @ u.T + 0.5 * np.eye(87)
        d = 1.0 / np.sqrt(np.diag(c))
        fc.append(c * d[:, None] * d[None, :])
    fc = np.stack(fc, axis=-1)
    rows_fc = _run_reference_loop("function_FC/VAE_CNN_GCN.py", dict(tensor=fc, net_data=[]))
    rows_fz = _run_reference_loop("function_fc_new/ResNet.py", dict(tensor=fc, net_data=[]), inside="load_data")
    save("prep", A_mat=A_mat, adjacency=adjacency, rank_mask=rank_mask(A_mat),
         raw_sc0=np.float32(np.moveaxis(t4[:, :, 0, :], -1, 0)), raw_sc1=np.float32(np.moveaxis(t4[:, :, 1, :], -1, 0)),
         raw_fc=np.float32(np.moveaxis(fc, -1, 0)), rows_sc=rows_sc, rows_log=rows_log, rows_fc=rows_fc,
         rows_fisher=rows_fz)


def make_traits():
    """The reference's own conditioning data for the trait-gated mixture model: data/trait.pt (a plain
    [100,4] fp32 tensor, read with a loader that executes nothing from the file) next to the 100
    structural connectomes of data/HCP_samples.mat prepared as function/VAE_CNN_NN.py:32-39 does,
    and the k-NN adjacency of function/MM_Gate.py:55-65 on their group mean.  Data only: the mixture
    head is build-defined, so its expected values come from the build's own float64 statement at
    test time."""
    traits = torch.load(os.path.join(REF, "data/trait.pt"), weights_only=True)
    assert isinstance(traits, torch.Tensor) and tuple(traits.shape) == (100, 4)
    x, A_mat, _, _ = sc_inputs()
    ns = load_defs("function/MM_Gate.py")
    adjacency = [list(map(int, a)) for a in ns["build_knn_adjacency"](A_mat, k=32)]
    save("traits", traits=traits.numpy().astype(np.float32), x=x.astype(np.float32),
         adjacency=np.array(adjacency, dtype=np.int64))


ALL = dict(traits=make_traits, cnn_nn_sc=make_cnn_nn_sc, cnn_gcn_sc=make_cnn_gcn_sc, cnn_gcn_fc=make_cnn_gcn_fc,
           cnn_nn_fc=make_cnn_nn_fc, mm_gate=make_mm_gate, fnn_gcn_fc=make_fnn_gcn_fc, cvae=make_cvae, ec87=make_ec87,
           losses=make_losses, fid=make_fid, prep=make_prep, graphs=make_graphs, resnet_fc=make_resnet_fc)

if __name__ == "__main__":
    torch.set_num_threads(8)
    which = sys.argv[1:] or list(ALL)
    for w in which:
        print(w)
        ALL[w]()
