"""CPU oracle for the connectome-VAE hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, in plain eager PyTorch / NumPy on the CPU, what the
reference scripts compute on the path the HIP library accelerates.  It is the
checker used by tests/, by __graft_entry__.smoke() and by bench.py's
cpu_baseline leg; nothing in the product package may import it.

Every function is written against a flat ``params`` dict keyed by the
reference's own state-dict names (``ec.1.weight`` ... ``fcintercept.bias``), so
the same tensors can be fed to the reference classes, to this oracle and to the
HIP path.  Gradients come from autograd on these functions.

Pinning: tests/test_oracle_golden.py checks every function here against the
fixtures in tests/golden/*.npz, which were produced by executing the
reference's class definitions (tests/golden/make_goldens.py).  The one piece
with no reference counterpart is the trait-gated mixture head (``mdn_*``): the
reference contains no such code, so that part is build-defined and its parity
against the reference is unpinned.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------
# data preparation  (function/VAE_CNN_NN.py:21-41, function_FC/VAE_CNN_GCN.py:31-43)
# --------------------------------------------------------------------------

def symmetrise_fill_crop(raw: np.ndarray, crop=(18, 86)) -> np.ndarray:
    """raw [n,n] -> raw+raw.T, diagonal := column means, optional crop.
    function/VAE_CNN_NN.py:34-36."""
    m = np.float32(raw + raw.T)
    np.fill_diagonal(m, m.mean(axis=0))
    if crop is not None:
        m = m[crop[0]:crop[1], crop[0]:crop[1]]
    return m


def prepare_sc(raw: np.ndarray, offset: float = 200.0, crop=(18, 86)) -> np.ndarray:
    """function/VAE_CNN_NN.py:33-39: one structural connectome -> flat fp32 row."""
    return (symmetrise_fill_crop(raw, crop).flatten() / offset).astype(np.float32)


def prepare_fc(raw: np.ndarray, crop=(18, 86)) -> np.ndarray:
    """function_FC/VAE_CNN_GCN.py:35-42: functional connectome -> [0,1] flat row."""
    return ((symmetrise_fill_crop(raw, crop) + 1.0) / 2.0).flatten().astype(np.float32)


def prepare_log(raw: np.ndarray) -> np.ndarray:
    """function/Full_NN.py:106-112: full 87x87 matrix, log(x+1)."""
    return np.log(symmetrise_fill_crop(raw, None).flatten() + 1).astype(np.float32)


def prepare_fisher(raw: np.ndarray, crop=(18, 86), eps=1e-7) -> np.ndarray:
    """function_fc_new/ResNet.py:17-19,33-41: Fisher z of the clipped correlations."""
    r = np.clip(symmetrise_fill_crop(raw, crop), -1 + eps, 1 - eps)
    return (0.5 * np.log((1 + r) / (1 - r))).flatten().astype(np.float32)


def rank_mask(A_mat: np.ndarray, n_size: int = 32) -> np.ndarray:
    """function/VAE-CNN-GCN.py:209: ascending rank of each row entry < n_size+1."""
    order = np.argsort(A_mat, axis=-1)
    rank = np.argsort(order, axis=-1)
    return rank < (n_size + 1)


def knn_adjacency(A_mat: np.ndarray, k: int = 32) -> np.ndarray:
    """function/MM_Gate.py:55-65: per node, itself first, then its k-1 strongest
    other neighbours in descending order of A_mat."""
    n = A_mat.shape[0]
    out = np.empty((n, k), dtype=np.int64)
    for u in range(n):
        desc = np.argsort(A_mat[u, :])[::-1][:k]
        desc = desc[desc != u][: k - 1]
        out[u, 0] = u
        out[u, 1:] = desc
    return out


# --------------------------------------------------------------------------
# conv / ResNet encoder  (function/VAE_CNN_NN.py:83-98, 110-141, 155-162)
# --------------------------------------------------------------------------

def _in(x):
    # nn.InstanceNorm2d defaults: affine=False, eps=1e-5, biased variance, no running stats
    return F.instance_norm(x, eps=1e-5)


def conv_features(p: dict, x: torch.Tensor, n_nodes: int = 68, n_res: int = 5, gates=None, normed=None) -> torch.Tensor:
    """x [B, n*n] -> pooled feature [B, 768].  Stem, two stride-2 convs, n_res residual
    blocks, 8x8 max pool; parameter indices follow the reference nn.Sequential.
    gates: 3 + n_res boolean masks, one per ReLU in network order: those ReLUs pass exactly the masked entries (the ones
    another run passed), so that a comparison with that run measures arithmetic and not which side of zero a
    rounding-level normalised value fell.  normed: a list that receives every ReLU's input (detached)."""
    it = iter(gates) if gates is not None else None

    def relu(t):
        if normed is not None:
            normed.append(t.detach())
        return F.relu(t) if it is None else t * next(it).to(t.dtype)

    h = x.view(-1, 1, n_nodes, n_nodes)
    h = relu(_in(F.conv2d(F.pad(h, (3, 3, 3, 3), mode="reflect"), p["ec.1.weight"], p["ec.1.bias"])))
    h = relu(_in(F.conv2d(h, p["ec.4.weight"], p["ec.4.bias"], stride=2, padding=1)))
    h = relu(_in(F.conv2d(h, p["ec.7.weight"], p["ec.7.bias"], stride=2, padding=1)))
    for i in range(10, 10 + n_res):
        a = F.conv2d(F.pad(h, (1, 1, 1, 1), mode="reflect"), p[f"ec.{i}.block.1.weight"], p[f"ec.{i}.block.1.bias"])
        a = relu(_in(a))
        b = F.conv2d(F.pad(a, (1, 1, 1, 1), mode="reflect"), p[f"ec.{i}.block.5.weight"], p[f"ec.{i}.block.5.bias"])
        h = h + _in(b)
    h = F.max_pool2d(h, kernel_size=8, stride=8)
    return h.reshape(h.shape[0], -1)


def conv_encode(p, x, n_nodes=68, gates=None, normed=None):
    f = conv_features(p, x, n_nodes, gates=gates, normed=normed)
    return F.linear(f, p["fc_mean.weight"], p["fc_mean.bias"]), F.linear(f, p["fc_log_var.weight"], p["fc_log_var.bias"])


def fnn_encode(p, x, gates=None):
    """function/MM_Gate.py:157-162: two ReLU towers D->1024->128->L.  gates = (g11, g12[, g111, g222]) boolean masks
    [B,1024] (and [B,128]): those ReLUs pass exactly the masked entries (the ones another run passed) instead of the
    positive ones, so that a comparison with that run measures arithmetic and not which side of zero a rounding-level
    pre-activation fell."""
    def gated(pre, i):
        return F.relu(pre) if gates is None or len(gates) <= i else pre * gates[i].to(pre.dtype)
    a1 = gated(F.linear(x, p["fc11.weight"], p["fc11.bias"]), 0)
    a2 = gated(F.linear(x, p["fc12.weight"], p["fc12.bias"]), 1)
    a = gated(F.linear(a1, p["fc111.weight"], p["fc111.bias"]), 2)
    b = gated(F.linear(a2, p["fc222.weight"], p["fc222.bias"]), 3)
    return F.linear(a, p["fc21.weight"], p["fc21.bias"]), F.linear(b, p["fc22.weight"], p["fc22.bias"])


def reparameterize(mu, logvar, eps):
    """function/VAE_CNN_NN.py:164-167 with the normal draw passed in."""
    return mu + eps * torch.exp(0.5 * logvar)


# --------------------------------------------------------------------------
# decoders
# --------------------------------------------------------------------------
_FIRST = ("fc3", "fc32", "fc33", "fc34", "fc35")
_SECOND = ("fc4", "fc5", "fc6", "fc7", "fc8")


def rank1_decode(p, z, *, mask=None, intercept="dense", final="exp", act=torch.sigmoid,
                 drop_masks=None, n_nodes=68):
    """Five-branch rank-1 decoder.
      SC, dense intercept + exp      function/VAE_CNN_NN.py:169-189   (intercept='dense', final='exp')
      masked second layers, diagonal function/VAE-CNN-GCN.py:166-185, function_FC/VAE_CNN_GCN.py:166-191
        intercept                      (mask=[n,n] 0/1, intercept='diag', final='none')
      FC, dropout, no intercept      function_FC/VAE_CNN_NN.py:165-197  (intercept='none', final='none')
    Returns (recon [B,n*n], h1+h2+h3+h4) -- the fifth branch is left out of the
    second output, as in the reference."""
    hs = []
    for r, (a, b) in enumerate(zip(_FIRST, _SECOND)):
        h = act(F.linear(z, p[a + ".weight"], p[a + ".bias"]))
        w2 = p[b + ".weight"]
        if mask is not None:
            w2 = w2 * mask
        h = act(F.linear(h, w2, p[b + ".bias"]))
        if drop_masks is not None:
            h = h * drop_masks[r]
        hs.append(h)
    s = sum(h.unsqueeze(2) * h.unsqueeze(1) for h in hs)
    s = act(s).reshape(z.shape[0], n_nodes * n_nodes)
    if intercept == "dense":
        s = F.linear(s, p["fcintercept.weight"], p["fcintercept.bias"])
    elif intercept == "diag":
        # GraphCNN with an identity mask: weight*I keeps the diagonal only
        s = s * torch.diagonal(p["fcintercept.weight"]) + p["fcintercept.bias"]
    elif intercept == "diagvec":
        s = s * p["fcintercept.weight"] + p["fcintercept.bias"]
    if final == "exp":
        s = torch.exp(s)
    return s, hs[0] + hs[1] + hs[2] + hs[3]


def gcn_layer(x, weight, bias, adjacency):
    """function/MM_Gate.py:84-93: out[:,u] = sum_j x[:,adj[u][j]] * W[u,j] + b[u], sigmoid."""
    gathered = x[:, adjacency]                      # [B, n, k]
    return torch.sigmoid((gathered * weight.unsqueeze(0)).sum(dim=2) + bias)


def gcn_decode(p, z, adjacency, R=5, n_layers=3):
    """function/MM_Gate.py:126-138: R channels, each Linear then n_layers GCN layers,
    outer products summed and symmetrised; returns log-rates [B, n*n]."""
    adjacency = torch.as_tensor(np.asarray(adjacency), dtype=torch.long)
    total = 0
    for r in range(R):
        g = F.linear(z, p[f"decoder.init_fc.{r}.weight"], p[f"decoder.init_fc.{r}.bias"])
        for l in range(n_layers):
            pre = f"decoder.channel_gcns.{r}.layers.{l}."
            g = gcn_layer(g, p[pre + "weight"], p[pre + "bias"], adjacency)
        total = total + g.unsqueeze(2) * g.unsqueeze(1)
    total = 0.5 * (total + total.transpose(1, 2))
    return total.reshape(z.shape[0], -1)


def cvae_forward(p, fc, sc, eps):
    """function_FC_SC/SC-FC.py:95-119."""
    h = F.relu(F.linear(torch.cat([fc, sc], dim=1), p["fc1.weight"], p["fc1.bias"]))
    h = F.relu(F.linear(h, p["fc2.weight"], p["fc2.bias"]))
    mu = F.linear(h, p["fc_mu.weight"], p["fc_mu.bias"])
    logvar = F.linear(h, p["fc_logvar.weight"], p["fc_logvar.bias"])
    z = reparameterize(mu, logvar, eps)
    h = F.relu(F.linear(torch.cat([z, fc], dim=1), p["fc3.weight"], p["fc3.bias"]))
    h = F.relu(F.linear(h, p["fc4.weight"], p["fc4.bias"]))
    return torch.sigmoid(F.linear(h, p["fc_out.weight"], p["fc_out.bias"])), mu, logvar


def cvae_hinge_loss(p, fc, sc, mask_idx, eps_a, eps_b, margin=0.01):
    """function_FC_SC/SC-FC.py:142-157: loss with FC + relu(loss_with - loss_without + margin)."""
    recon, mu, logvar = cvae_forward(p, fc, sc, eps_a)
    with_fc = vae_loss(recon, sc, mu, logvar, "mse")
    fc_no = fc.clone()
    fc_no[torch.as_tensor(mask_idx, dtype=torch.long)] = 0.0
    recon_no, mu_no, logvar_no = cvae_forward(p, fc_no, sc, eps_b)
    without = vae_loss(recon_no, sc, mu_no, logvar_no, "mse")
    return with_fc + torch.relu(with_fc - without + margin), with_fc, without, recon, recon_no


# --------------------------------------------------------------------------
# losses
# --------------------------------------------------------------------------

def kl_term(mu, logvar):
    return -0.5 * torch.sum(1 + logvar - mu * mu - torch.exp(logvar))


def recon_term(recon, x, mode):
    """poisson_rate  function/VAE_CNN_NN.py:198   sum(r - x*log(r+1e-8))
       poisson_log   function/MM_Gate.py:180      sum(exp(r) - x*r)
       mse           function_FC/VAE_CNN_GCN.py:209  sum((r-x)^2)"""
    if mode == "poisson_rate":
        return torch.sum(recon - x * torch.log(recon + 1e-8))
    if mode == "poisson_log":
        return torch.sum(torch.exp(recon) - x * recon)
    if mode == "mse":
        return torch.sum((recon - x) ** 2)
    raise ValueError(mode)


def vae_loss(recon, x, mu, logvar, mode):
    return recon_term(recon, x, mode) + kl_term(mu, logvar)


# --------------------------------------------------------------------------
# trait-gated Gaussian-mixture head -- BUILD-DEFINED (no reference code exists)
# --------------------------------------------------------------------------

def mdn_gate_logits(traits, gate_w, gate_b):
    """Gate logits [B,K] = traits [B,T] @ gate_w[K,T]^T + gate_b[K]."""
    return F.linear(traits, gate_w, gate_b)


def mdn_nll(comp_mean, log_sigma, gate_logits, x):
    """Negative log-likelihood of a trait-gated mixture of K isotropic-per-component
    Gaussians over whole connectomes:
        comp_mean [B,K,D], log_sigma [K], gate_logits [B,K], x [B,D]
        ll_bk = sum_d ( -0.5*((x_bd-m_bkd)/s_k)^2 - log s_k - 0.5*log(2*pi) )
        nll   = - sum_b logsumexp_k( log_softmax(gate_logits)_bk + ll_bk )."""
    D = x.shape[1]
    diff = x.unsqueeze(1) - comp_mean
    inv_var = torch.exp(-2.0 * log_sigma).view(1, -1)
    ll = -0.5 * (diff * diff).sum(dim=2) * inv_var - D * log_sigma.view(1, -1) - 0.5 * D * math.log(2 * math.pi)
    return -torch.logsumexp(F.log_softmax(gate_logits, dim=1) + ll, dim=1).sum()


def gcn_components(p, z, adjacency, R=5, n_layers=3):
    """Per-channel symmetric rank-1 means g_r g_r^T of the GCN decoder, [B,R,n*n]."""
    adjacency = torch.as_tensor(np.asarray(adjacency), dtype=torch.long)
    comps = []
    for r in range(R):
        g = F.linear(z, p[f"decoder.init_fc.{r}.weight"], p[f"decoder.init_fc.{r}.bias"])
        for l in range(n_layers):
            pre = f"decoder.channel_gcns.{r}.layers.{l}."
            g = gcn_layer(g, p[pre + "weight"], p[pre + "bias"], adjacency)
        comps.append((g.unsqueeze(2) * g.unsqueeze(1)).reshape(z.shape[0], -1))
    return torch.stack(comps, dim=1)


# --------------------------------------------------------------------------
# whole-model forward + loss for each script
# --------------------------------------------------------------------------
VARIANTS = {
    # name: (encoder, decoder kwargs, loss mode)
    "cnn_nn_sc": ("conv", dict(intercept="dense", final="exp"), "poisson_rate"),      # function/VAE_CNN_NN.py
    "cnn_gcn_sc": ("conv", dict(intercept="diag", final="none"), "poisson_log"),      # function/VAE-CNN-GCN.py
    "cnn_gcn_fc": ("conv", dict(intercept="diag", final="none"), "mse"),              # function_FC/VAE_CNN_GCN.py
    "cnn_nn_fc": ("conv", dict(intercept="none", final="none"), "mse"),               # function_FC/VAE_CNN_NN.py (eval)
    "fnn_gcn_fc": ("fnn", dict(intercept="diag", final="none", act=torch.tanh), "mse"),   # function_FC/MM_Gate_test.py
}


def vae_forward(variant, p, x, eps, mask=None, adjacency=None, n_nodes=68, conv_gates=None, conv_normed=None):
    if variant == "mm_gate":                                                          # function/MM_Gate.py:172-177
        mu, logvar = fnn_encode(p, x)
        z = reparameterize(mu, logvar, eps)
        return gcn_decode(p, z, adjacency), mu, logvar, z, None
    enc, dkw, _ = VARIANTS[variant]
    mu, logvar = fnn_encode(p, x) if enc == "fnn" else conv_encode(p, x, n_nodes, conv_gates, conv_normed)
    z = reparameterize(mu, logvar, eps)
    recon, x_latent = rank1_decode(p, z, mask=mask, n_nodes=n_nodes, **dkw)
    return recon, mu, logvar, z, x_latent


def loss_mode(variant):
    return "poisson_log" if variant == "mm_gate" else VARIANTS[variant][2]


def train_step_loss(variant, p, x, eps, **kw):
    recon, mu, logvar, _, _ = vae_forward(variant, p, x, eps, **kw)
    return vae_loss(recon, x, mu, logvar, loss_mode(variant))


def adam_step(params, grads, state, step, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam defaults as used at function/VAE_CNN_NN.py:204 (no weight decay,
    no amsgrad): in-place update of params and of state = {name: (m, v)}."""
    bc1 = 1.0 - b1 ** step
    bc2 = 1.0 - b2 ** step
    for k in params:
        m, v = state[k]
        m.mul_(b1).add_(grads[k], alpha=1 - b1)
        v.mul_(b2).addcmul_(grads[k], grads[k], value=1 - b2)
        denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
        params[k].addcdiv_(m, denom, value=-lr / bc1)


# --------------------------------------------------------------------------
# FID  (function/VAE_CNN_NN.py:310-316, 331-337; validation/FID.py:61-81)
# --------------------------------------------------------------------------

def fid_from_stats(mu1, sigma1, mu2, sigma2):
    from scipy.linalg import sqrtm
    diff = mu1 - mu2
    covmean = sqrtm(sigma1.dot(sigma2))
    if np.iscomplexobj(covmean):
        covmean = covmean.real
    return float(diff.dot(diff) + np.trace(sigma1 + sigma2 - 2 * covmean))


def fid_from_features(a: np.ndarray, b: np.ndarray) -> float:
    return fid_from_stats(a.mean(axis=0), np.cov(a, rowvar=False), b.mean(axis=0), np.cov(b, rowvar=False))


def fid_symmetric(mu1, sigma1, mu2, sigma2):
    """Same value through the symmetric form tr sqrt(S1^1/2 S2 S1^1/2) (eigenvalues of
    S1*S2 are those of the symmetric PSD matrix), used to cross-check the iteration the
    GPU path runs."""
    w, v = np.linalg.eigh(sigma1)
    root = (v * np.sqrt(np.clip(w, 0, None))) @ v.T
    ev = np.linalg.eigvalsh(root @ sigma2 @ root)
    diff = mu1 - mu2
    return float(diff.dot(diff) + np.trace(sigma1) + np.trace(sigma2) - 2 * np.sqrt(np.clip(ev, 0, None)).sum())


# ---- evaluation loop (function/VAE_CNN_NN.py:239-289) --------------------------------------------
def reconstruct(variant, params, x, eps, batch_size=64, **kw):
    """function/VAE_CNN_NN.py:242-260: forward passes in batches of 64 without gradients; the
    reconstruction is the first output.  `eps` [n,L] stands in for the draws of reparameterize."""
    out = []
    with torch.no_grad():
        for lo in range(0, x.shape[0], batch_size):
            res = vae_forward(variant, params, x[lo:lo + batch_size], eps[lo:lo + batch_size], **kw)
            out.append(res[0])
    return torch.cat(out)


def normalise_for_metrics(real, generated):
    """function/VAE_CNN_NN.py:268-271."""
    real, generated = np.asarray(real), np.asarray(generated)
    max_value = real.max()
    return real / max_value, np.maximum(generated / max_value, 0)


def compute_mse(real, generated):
    """function/VAE_CNN_NN.py:273-276 on [n,N,N] stacks."""
    return ((real - generated) ** 2).mean(axis=(1, 2)).mean()


def compute_psnr(real, generated, max_value=1.0):
    """function/VAE_CNN_NN.py:281-286."""
    mse_values = ((real - generated) ** 2).mean(axis=(1, 2))
    with np.errstate(divide="ignore"):
        psnr = np.where(mse_values == 0, np.inf, 20 * np.log10(max_value) - 10 * np.log10(mse_values))
    return psnr[np.isfinite(psnr)].mean()


def compute_spectral_features(matrix, k=None):
    """function/VAE_CNN_NN.py:302-308."""
    eigenvalues = np.sort(np.real(np.linalg.eigvals(matrix)))[::-1]
    return eigenvalues[:k] if k is not None else eigenvalues


def ssim(im1, im2, data_range=1.0, win_size=7, K1=0.01, K2=0.03):
    """Mean SSIM as skimage.metrics.structural_similarity computes it with its defaults (uniform
    window, use_sample_covariance=True, crop of (win_size-1)//2 before the mean) -- the call at
    function/VAE_CNN_NN.py:294.  skimage is not installed here: this follows its published
    algorithm (parity unpinned)."""
    from scipy.ndimage import uniform_filter
    im1, im2 = np.asarray(im1, dtype=np.float64), np.asarray(im2, dtype=np.float64)
    NP = win_size ** im1.ndim
    cov_norm = NP / (NP - 1)
    ux, uy = uniform_filter(im1, size=win_size), uniform_filter(im2, size=win_size)
    uxx, uyy, uxy = (uniform_filter(im1 * im1, size=win_size), uniform_filter(im2 * im2, size=win_size),
                     uniform_filter(im1 * im2, size=win_size))
    vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
    C1, C2 = (K1 * data_range) ** 2, (K2 * data_range) ** 2
    S = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux ** 2 + uy ** 2 + C1) * (vx + vy + C2))
    pad = (win_size - 1) // 2
    return S[pad:-pad, pad:-pad].mean()


def graph_features(matrix, weighted_paths=True):
    """compute_graph_features: validation/FID.py:6-58 (weighted_paths=True: every edge carries
    distance = 1/(w + 1e-5)) and function/VAE_CNN_NN.py:341-379 (weighted_paths=False: the script
    never attaches the 'distance' attribute, so path lengths are hop counts).  Restates the networkx
    3.4 calls it makes: from_numpy_array (an edge wherever either triangle is non-zero; the weight is
    the lower-triangle entry when that is non-zero), weighted degree (a self-loop counts twice),
    Onnela's weighted clustering on weights / max edge weight, average shortest path length
    (per connected component with more than one node when the graph is disconnected).
    Returns [degree mean, degree std, clustering mean, clustering std, average path length]."""
    m = np.asarray(matrix, dtype=np.float64)
    n = m.shape[0]
    mx = m.max()
    an = m / mx if mx > 0 else np.zeros_like(m)
    low = np.tril(an)
    w = np.where(low != 0, low, np.triu(an).T)            # lower entry, else the upper one
    w = w + np.tril(w, -1).T                              # symmetric edge weights, diagonal = self-loops
    edge = w != 0
    off = edge & ~np.eye(n, dtype=bool)
    deg = (w * off).sum(axis=1) + 2 * np.diag(w)
    wmax = w[edge].max() if edge.any() else 1.0
    wh = np.cbrt(w / wmax)
    clust = np.zeros(n)
    for i in range(n):
        nb = np.flatnonzero(off[i])
        d = len(nb)
        if d < 2:
            continue
        sub = wh[np.ix_(nb, nb)] * off[np.ix_(nb, nb)]
        t = (wh[i, nb][:, None] * wh[i, nb][None, :] * sub).sum()      # ordered pairs = 2 x triangles
        clust[i] = 0.0 if t == 0 else t / (d * (d - 1))
    upper = np.triu(an) + np.triu(an, 1).T                              # distance_matrix[i, j] with i <= j
    dist = np.where(off, (1.0 / (upper + 1e-5)) if weighted_paths else 1.0, np.inf)
    np.fill_diagonal(dist, 0.0)
    for k in range(n):
        dist = np.minimum(dist, dist[:, k:k + 1] + dist[k:k + 1, :])
    comp = np.array([np.flatnonzero(np.isfinite(dist[i]))[0] for i in range(n)])
    lengths = []
    for r in np.unique(comp):
        idx = np.flatnonzero(comp == r)
        if len(idx) > 1:
            lengths.append(dist[np.ix_(idx, idx)].sum() / (len(idx) * (len(idx) - 1)))
    if len(np.unique(comp)) == 1:
        path = lengths[0] if lengths else 0.0
    else:
        path = float(np.mean(lengths)) if lengths else 0.0
    return np.array([deg.mean(), deg.std(), clust.mean(), clust.std(), path])


# ---- multi-branch ResNet-MLP VAE (function_fc_new/ResNet.py:83-238) -----------------------------
def _lin(p, name, x):
    return F.linear(x, p[name + ".weight"], p[name + ".bias"])


def _branch_mlp(p, pre, x, n_blocks):
    h = F.relu(_lin(p, pre + ".init_fc", x))                                   # :124-129
    for i in range(n_blocks):
        b = f"{pre}.resblocks.{i}"
        out = _lin(p, b + ".fc2", F.relu(_lin(p, b + ".fc1", h)))               # :96-106
        ident = _lin(p, b + ".shortcut", h) if (b + ".shortcut.weight") in p else h
        h = F.relu(out + ident)
    return _lin(p, pre + ".final_fc", h)


def _branch_gcn(p, pre, x, n_layers, mask=None):
    h = F.relu(_lin(p, pre + ".init_fc", x))                                   # :141-149
    for i in range(n_layers):
        w = p[f"{pre}.gcn_layers.{i}.weight"]
        if mask is not None:
            w = w * mask
        h = F.relu(F.linear(h, w, p[f"{pre}.gcn_layers.{i}.bias"]) + h)
    return h


def resnet_vae_forward(p, x, eps, n_mlp1=8, n_mlp2=8, n_gcn=10, mask=None, n_nodes=68):
    """function_fc_new/ResNet.py:208-238; returns (recon, mu, logvar, z)."""
    mu, logvar = fnn_encode(p, x)
    z = reparameterize(mu, logvar, eps)
    h = _lin(p, "fc_init_decode", z)
    d = "multi_branch_decoder"
    merged = (_branch_mlp(p, d + ".branch_mlp1", h, n_mlp1) + _branch_gcn(p, d + ".branch_gcn", h, n_gcn, mask)
              + _branch_mlp(p, d + ".branch_mlp2", h, n_mlp2))                  # :174-180
    outer = (merged.unsqueeze(2) * merged.unsqueeze(1)).reshape(-1, n_nodes * n_nodes)
    return _lin(p, d + ".final_fc", outer), mu, logvar, z
