"""Module surface of the reference scripts, computing through the HIP library.

Class, attribute and state-dict names follow the reference so that a script can swap its
own `VAE` / `loss_function` for these:
    conv-encoder VAEs   function/VAE_CNN_NN.py:100-195, function/VAE-CNN-GCN.py:97-199,
                        function_FC/VAE_CNN_GCN.py:97-205, function_FC/VAE_CNN_NN.py:95-203
    GCN-decoder VAE     function/MM_Gate.py:75-177
    conditional VAE     function_FC_SC/SC-FC.py:79-119
Submodules are created in the reference's order, so `torch.manual_seed(s); VAE()` draws the
same initial weights as the reference class does under the same seed.
"""
import math

import numpy as np
import torch
from torch import nn

from . import ops


# ---------------------------------------------------------------------------------------
# parameter holders
# ---------------------------------------------------------------------------------------
class Linear(nn.Linear):
    """nn.Linear whose forward is a library GEMM plus the HIP bias/activation kernel."""

    def forward(self, x, act=None):
        return ops.linear_act(x, self.weight, self.bias, act)


class GraphCNN(Linear):
    """Masked linear layer (function/VAE-CNN-GCN.py:62-78): set_mask zeroes the masked
    weights once, forward multiplies by the mask again."""

    def __init__(self, in_features, out_features, bias=True):
        super().__init__(in_features, out_features, bias)
        self.mask_flag = False
        self.register_buffer("mask", None, persistent=False)

    def set_mask(self, mask):
        mask = torch.as_tensor(mask, dtype=torch.float32).to(self.weight.device)
        self.mask = mask.contiguous()
        with torch.no_grad():
            self.weight.mul_(self.mask)
        self.mask_flag = True

    def get_mask(self):
        return self.mask

    def forward(self, x, act=None):
        w = ops.mul(self.weight, self.mask) if self.mask_flag else self.weight
        return ops.linear_act(x, w, self.bias, act)


class DiagonalIntercept(nn.Module):
    """`GraphCNN(D, D)` under an identity mask (function/VAE-CNN-GCN.py:150,219-220) is a
    per-edge scale and shift; only the diagonal is stored.  A dense [D,D] weight in a
    state dict is accepted (its diagonal is taken) and `dense_weight()` gives one back."""

    def __init__(self, features):
        super().__init__()
        self.features = features
        # draw what nn.Linear(D, D) would draw, keep the diagonal
        bound = 1.0 / math.sqrt(features)
        rows = []
        for r0 in range(0, features, 512):   # kaiming_uniform(a=sqrt(5)) == U(-1/sqrt(fan_in), +)
            blk = torch.empty(min(512, features - r0), features).uniform_(-bound, bound)
            rows.append(torch.diagonal(blk, offset=r0).clone())
        self.weight = nn.Parameter(torch.cat(rows))
        self.bias = nn.Parameter(torch.empty(features).uniform_(-bound, bound))
        self.mask_flag = True

    def set_mask(self, mask):
        """Accepts the identity mask of the reference scripts (anything else is not a diagonal layer)."""
        m = torch.as_tensor(mask)
        if m.dim() == 2:
            if m.shape != (self.features, self.features) or not bool((m.diagonal() == 1).all()) \
                    or int(m.sum().item()) != self.features:
                raise ValueError("DiagonalIntercept only supports the identity mask")

    def dense_weight(self):
        return torch.diag(self.weight.detach())

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        key = prefix + "weight"
        if key in state_dict and state_dict[key].dim() == 2:
            state_dict[key] = torch.diagonal(state_dict[key]).clone()
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def forward(self, x):
        return ops.diag_affine(x, self.weight, self.bias)


class _Slots(nn.Module):
    """A container whose children are registered under the given names (keeps the reference's
    nn.Sequential indices in the state dict without instantiating the parameter-free layers)."""

    def __init__(self, named):
        super().__init__()
        for name, mod in named:
            self.add_module(str(name), mod)

    def __getitem__(self, idx):
        return self._modules[str(idx)]


class _ResBlockParams(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.block = _Slots([(1, nn.Conv2d(ch, ch, 3)), (5, nn.Conv2d(ch, ch, 3))])


def _dev_int(a, device):
    return torch.as_tensor(np.asarray(a, dtype=np.int32), device=device).contiguous()


# ---------------------------------------------------------------------------------------
# conv-encoder VAE family
# ---------------------------------------------------------------------------------------
CONV_PRECISIONS = {"fp32_mfma": 0, "bf16": 1, "bf16_planar": 1, "bf16x2": 2, "fp32": 2, "fp32_planar": 3}
CONV_BLOCKED = {"bf16", "fp32"}  # precisions whose activations live in the channel-blocked 16-bit layout


class ConvVAE(nn.Module):
    """Conv/ResNet encoder (or, with encoder='fnn', the two-tower FNN encoder of
    function_FC/MM_Gate_test.py) + five-branch rank-1 decoder.

    precision: how the encoder's 3x3 convolutions multiply (CONV_PRECISIONS): 'fp32_mfma' the fp32 matrix
              instruction; 'fp32' (default) fp32 accuracy from two fp16 parts per operand (three fp16 matrix-core
              products, 3 * 2^-22 per product), activations kept between the convolutions in the channel-blocked
              layout the matrix cores read directly; 'fp32_planar' fp32 accuracy from three bf16 parts with fp32 planes
              between the layers (weight gradients from two parts: 6e-6 of their norm); 'bf16x2' two bf16 parts
              (2^-16 per product); 'bf16' operands rounded to bf16, channel-blocked ('bf16_planar': the same rounding
              with fp32 planes between the layers).  fp32 accumulation always.
    decoder:  'nn'   plain second layers                          (VAE_CNN_NN scripts)
              'gcn'  GraphCNN second layers + diagonal intercept  (VAE-CNN-GCN scripts)
    intercept: 'dense' | 'diag' | 'none';  final: 'exp' | 'none';  dropout: p or 0
    """

    def __init__(self, decoder="nn", intercept="dense", final="exp", dropout=0.0, n_nodes=68, latent_dim=68,
                 act="sigmoid", encoder="conv", precision="fp32"):
        super().__init__()
        if precision not in CONV_PRECISIONS:
            raise ValueError(f"precision {precision!r}: one of {sorted(CONV_PRECISIONS)}")
        self.precision = precision
        self.n_nodes, self.latent_dim = n_nodes, latent_dim
        self.decoder_kind, self.intercept_kind, self.final, self.act = decoder, intercept, final, act
        self.encoder_kind = encoder
        N = n_nodes
        if encoder == "fnn":
            # two ReLU towers D->1024->128->L (function_FC/MM_Gate_test.py:87-92,104-109)
            self.fc11 = Linear(N * N, 1024)
            self.fc12 = Linear(N * N, 1024)
            self.fc111 = Linear(1024, 128)
            self.fc222 = Linear(1024, 128)
            self.fc21 = Linear(128, latent_dim)
            self.fc22 = Linear(128, latent_dim)
        elif encoder == "conv":
            init_depth = 48
            ch = init_depth * 4
            ec = [(1, nn.Conv2d(1, init_depth, 7)), (4, nn.Conv2d(init_depth, init_depth * 2, 3, stride=2, padding=1)),
                  (7, nn.Conv2d(init_depth * 2, ch, 3, stride=2, padding=1))]
            ec += [(10 + i, _ResBlockParams(ch)) for i in range(5)]
            self.ec = _Slots(ec)
            self.n_res = 5
            s = (n_nodes + 1) // 2
            s = (s + 1) // 2
            pooled = s // 8
            if pooled < 1:
                raise ValueError(f"n_nodes={n_nodes} is too small for the 8x8 max pool")
            flat = ch * pooled * pooled
            self.fc_mean = Linear(flat, latent_dim)
            self.fc_log_var = Linear(flat, latent_dim)
        else:
            raise ValueError(f"unknown encoder {encoder!r}")
        for name in ("fc3", "fc32", "fc33", "fc34", "fc35"):
            setattr(self, name, Linear(latent_dim, N))
        second = GraphCNN if decoder == "gcn" else Linear
        for name in ("fc4", "fc5", "fc6", "fc7", "fc8"):
            setattr(self, name, second(N, N))
        self.dropout_p = float(dropout)
        if intercept == "dense":
            self.fcintercept = Linear(N * N, N * N)
        elif intercept == "diag":
            self.fcintercept = DiagonalIntercept(N * N)
        self._spec = None

    # -- encoder -------------------------------------------------------------------
    def conv_params(self):
        p = [self.ec[1], self.ec[4], self.ec[7]]
        for i in range(self.n_res):
            blk = self.ec[10 + i].block
            p += [blk[1], blk[5]]
        return p

    def encode(self, x):
        x = x.reshape(-1, self.n_nodes * self.n_nodes)
        if self.encoder_kind == "fnn":
            h11 = self.fc111(self.fc11(x, act="relu"), act="relu")
            h12 = self.fc222(self.fc12(x, act="relu"), act="relu")
            return self.fc21(h11), self.fc22(h12)
        feat = ops.conv_encoder(x, self.n_nodes, [(m.weight, m.bias) for m in self.conv_params()],
                                parts=CONV_PRECISIONS[self.precision], blocked=self.precision in CONV_BLOCKED)
        return self.fc_mean(feat), self.fc_log_var(feat)

    def reparameterize(self, mu, logvar, eps=None):
        return ops.reparameterize(mu, logvar, eps)

    # -- decoder -------------------------------------------------------------------
    def set_mask(self, masks):
        """masks[0..4]: [N,N] 0/1 masks of fc4..fc8, masks[5]: identity for the intercept
        (function/VAE-CNN-GCN.py:193-199)."""
        for name, m in zip(("fc4", "fc5", "fc6", "fc7", "fc8"), masks[:5]):
            getattr(self, name).set_mask(m)
        if len(masks) > 5 and hasattr(self, "fcintercept") and hasattr(self.fcintercept, "set_mask"):
            self.fcintercept.set_mask(masks[5])
        self._spec = None

    def _chain_spec(self):
        if self._spec is None:
            firsts = ("fc3", "fc32", "fc33", "fc34", "fc35")
            seconds = ("fc4", "fc5", "fc6", "fc7", "fc8")
            masks = [getattr(getattr(self, s), "mask", None) if getattr(getattr(self, s), "mask_flag", False) else None
                     for s in seconds]
            if all(m is None for m in masks):
                mask = None
            elif all(m is not None and torch.equal(m, masks[0]) for m in masks):
                mask = masks[0]
            else:
                raise NotImplementedError("the fused decoder needs the five second-layer masks to be equal, "
                                          "as they are in the reference scripts")
            layers = [dict(kind="linear", act=self.act), dict(kind="linear", act=self.act, mask=mask)]
            self._spec = ops.ChainSpec(self.n_nodes, self.latent_dim, 5, layers, epi_act=self.act, xlat_channels=4)
            self._names = list(zip(firsts, seconds))
        return self._spec

    def decode(self, z):
        spec = self._chain_spec()
        params = []
        for a, b in self._names:
            ma, mb = getattr(self, a), getattr(self, b)
            params += [ma.weight, ma.bias, mb.weight, mb.bias]
        drop = None
        if self.dropout_p > 0 and self.training:
            keep = 1.0 - self.dropout_p
            drop = torch.bernoulli(torch.full((5, z.shape[0], self.n_nodes), keep, device=z.device)).div_(keep)
        iw = ib = None
        if self.intercept_kind == "diag":
            iw, ib = self.fcintercept.weight, self.fcintercept.bias
        out, xlat = ops.chain_decoder(z, spec, params, iw, ib, drop)
        if self.intercept_kind == "dense":
            out = self.fcintercept(out)
        if self.final == "exp":
            out = ops.exp(out)
        return out, xlat

    def forward(self, x, eps=None):
        mu, logvar = self.encode(x)
        z = self.reparameterize(mu, logvar, eps)
        recon, x_latent = self.decode(z)
        return recon, mu, logvar, x_latent


# ---------------------------------------------------------------------------------------
# GCN-decoder VAE (function/MM_Gate.py)
# ---------------------------------------------------------------------------------------
class GCNLayer(nn.Module):
    def __init__(self, n_nodes=68, k=32):
        super().__init__()
        self.n_nodes, self.k = n_nodes, k
        self.weight = nn.Parameter(torch.zeros(n_nodes, k))
        nn.init.xavier_uniform_(self.weight)
        self.bias = nn.Parameter(torch.zeros(n_nodes))


class ChannelGCN(nn.Module):
    def __init__(self, n_nodes=68, k=32, n_layers=3):
        super().__init__()
        self.n_layers = n_layers
        self.layers = nn.ModuleList([GCNLayer(n_nodes, k) for _ in range(n_layers)])


class MultiChannelGCNDecoder(nn.Module):
    """function/MM_Gate.py:113-138 as one fused chain-decoder launch."""

    def __init__(self, R=5, n_nodes=68, k=32, n_layers=3, latent_dim=68):
        super().__init__()
        self.R, self.n_nodes, self.k, self.n_layers, self.latent_dim = R, n_nodes, k, n_layers, latent_dim
        self.init_fc = nn.ModuleList([Linear(latent_dim, n_nodes) for _ in range(R)])
        self.channel_gcns = nn.ModuleList([ChannelGCN(n_nodes, k, n_layers) for _ in range(R)])
        self._adj_key = None
        self._adj_ident = None
        self._adj_src = None
        self._adj = None
        self._spec = None

    def _adjacency(self, adjacency, device):
        """int32 device copy of the adjacency and the chain spec built on it.  The conversion (a host
        list walk, or a device-to-host copy for a tensor) runs only when the object changes: the same
        list object, or the same tensor storage at the same version, is a cache hit - so a step
        issues no host synchronisation for it and can be captured into a HIP graph."""
        if isinstance(adjacency, torch.Tensor):
            ident = ("t", adjacency.data_ptr(), adjacency._version, tuple(adjacency.shape), str(device))
        else:
            ident = ("o", id(adjacency), len(adjacency), str(device))
        if ident == self._adj_ident and self._adj is not None:
            return self._adj
        arr = np.asarray([list(map(int, a)) for a in adjacency], dtype=np.int32) if not isinstance(adjacency, torch.Tensor) \
            else adjacency.detach().cpu().numpy().astype(np.int32)
        key = (arr.tobytes(), str(device))
        if key != self._adj_key:
            if arr.shape != (self.n_nodes, self.k):
                raise ValueError(f"adjacency must be [{self.n_nodes},{self.k}], got {arr.shape}")
            if arr.min() < 0 or arr.max() >= self.n_nodes:
                raise ValueError("adjacency holds node ids outside [0, n_nodes)")
            self._adj = torch.from_numpy(arr).to(device).contiguous()
            self._adj_key = key
            layers = [dict(kind="linear", act=None)] + [dict(kind="gcn", act="sigmoid", adj=self._adj)
                                                        for _ in range(self.n_layers)]
            self._spec = ops.ChainSpec(self.n_nodes, self.latent_dim, self.R, layers, epi_act=None, xlat_channels=0)
        self._adj_ident = ident
        self._adj_src = adjacency        # keeps the object alive so its id() stays its own
        return self._adj

    def components(self, z, adjacency):
        """Per-channel matrices g_r g_r^T [B,R,N*N]: the component means of the mixture head."""
        self._adjacency(adjacency, z.device)
        if getattr(self, "_comp_spec", None) is None or self._comp_spec_src is not self._spec:
            sp = self._spec
            self._comp_spec = ops.ChainSpec(sp.N, sp.L, sp.R, sp.layers, epi_act=None, xlat_channels=0, components=True)
            self._comp_spec_src = sp
        out, _ = ops.chain_decoder(z, self._comp_spec, self.chain_params())
        return out

    def node_vectors(self, z, adjacency):
        """The channels' node vectors g_r [B,R,NP] alone (columns >= n_nodes are padding): what the
        mixture likelihood needs when it forms g_r g_r^T on the fly."""
        self._adjacency(adjacency, z.device)
        if getattr(self, "_node_spec", None) is None or self._node_spec_src is not self._spec:
            sp = self._spec
            self._node_spec = ops.ChainSpec(sp.N, sp.L, sp.R, sp.layers, epi_act=None, xlat_channels=0, nodes_only=True)
            self._node_spec_src = sp
        out, _ = ops.chain_decoder(z, self._node_spec, self.chain_params())
        return out

    def chain_params(self):
        params = []
        for r in range(self.R):
            params += [self.init_fc[r].weight, self.init_fc[r].bias]
            for ly in self.channel_gcns[r].layers:
                params += [ly.weight, ly.bias]
        return params

    def forward(self, z, adjacency):
        self._adjacency(adjacency, z.device)
        out, _ = ops.chain_decoder(z, self._spec, self.chain_params())
        return out


class GCNVAE(nn.Module):
    """`VAE` of function/MM_Gate.py:143-177."""

    # forward(..., nodes_only=True) gives node vectors whose plain sum of outer products is the
    # reconstruction: the trainer may take loss and gradients from them directly
    rank_one_sum_decoder = True

    def __init__(self, R=5, n_nodes=68, k=32, n_layers=3, latent_dim=68, symmetric_input=False, first_layer_gemm="split"):
        super().__init__()
        self.n_nodes = n_nodes
        self.symmetric_input = symmetric_input
        if first_layer_gemm not in ("split", "library"):
            raise ValueError("first_layer_gemm must be 'split' or 'library'")
        # "split": with symmetric_input and batches of at least ops.SPLIT_GEMM_MIN_BATCH both first layers
        # run as one hand-written MFMA product on three-way bf16 splits (fp32 accuracy); "library":
        # always the library's fp32 GEMMs
        self.first_layer_gemm = first_layer_gemm
        self.register_buffer("_asym", torch.zeros(1, dtype=torch.int32), persistent=False)
        # largest magnitude of the packed batch (bit pattern of a float), left by the packing pass for the
        # operand split of the first-layer product
        self.register_buffer("_xmax", torch.zeros(ops.ABSMAX_SLOTS, dtype=torch.int32), persistent=False)
        D = n_nodes * n_nodes
        self.fc11 = Linear(D, 1024)
        self.fc12 = Linear(D, 1024)
        self.fc111 = Linear(1024, 128)
        self.fc222 = Linear(1024, 128)
        self.fc21 = Linear(128, latent_dim)
        self.fc22 = Linear(128, latent_dim)
        self.decoder = MultiChannelGCNDecoder(R=R, n_nodes=n_nodes, k=k, n_layers=n_layers, latent_dim=latent_dim)

    def first_layers(self, x):
        """relu(fc11(x)), relu(fc12(x)) (function/MM_Gate.py:157-158) the way this model computes them.
        `symmetric_input`: the batch is promised to hold symmetric matrices (what the data
        preparation produces), so both layers run on the packed upper triangle with folded
        weights, half the GEMM work for the same sums.  The promise is checked on the device with
        every batch; `check_inputs()` reports a violation.
        Batches of at least ops.SPLIT_GEMM_MIN_BATCH take the hand-written split product, smaller ones the
        library's fp32 GEMM: both are fp32-level arithmetic (results agree to a few 1e-7 of sum |x||w|),
        but not bit for bit - a batch of 1023 and one of 1024 round differently."""
        if self.symmetric_input:
            split = self.first_layer_gemm == "split" and x.shape[0] >= ops.SPLIT_GEMM_MIN_BATCH
            xmax = self._xmax if split and ops.ARITH_FWD == "f16x2" else None
            xt = ops.tri_pack(x, self.n_nodes, self._asym, xmax)
            if split:
                return ops.sym_linear_pair(xt, self.fc11.weight, self.fc11.bias, self.fc12.weight, self.fc12.bias,
                                           self.n_nodes, xmax)
            return (ops.sym_linear_act(xt, self.fc11.weight, self.fc11.bias, "relu", self.n_nodes),
                    ops.sym_linear_act(xt, self.fc12.weight, self.fc12.bias, "relu", self.n_nodes))
        if (self.first_layer_gemm == "split" and x.dim() == 2 and x.shape[0] >= ops.SPLIT_GEMM_MIN_BATCH
                and not x.requires_grad):
            return ops.linear_pair(x, self.fc11.weight, self.fc11.bias, self.fc12.weight, self.fc12.bias)
        return self.fc11(x, act="relu"), self.fc12(x, act="relu")

    def encode(self, x):
        a11, a12 = self.first_layers(x)
        h11 = self.fc111(a11, act="relu")
        h12 = self.fc222(a12, act="relu")
        return self.fc21(h11), self.fc22(h12)

    def folded_grad_params(self):
        """[(parameter, n_nodes)]: the weights whose gradient the symmetric-input first layers produce
        folded onto the upper triangle (a trainer may exchange and apply it in that form)."""
        if not self.symmetric_input:
            return []
        return [(self.fc11.weight, self.n_nodes), (self.fc12.weight, self.n_nodes)]

    def check_inputs(self):
        """Raises if a batch seen since the last call was not symmetric although the model was
        built with symmetric_input=True (its outputs for that batch are then wrong).  Reads one
        integer from the device, so call it where the host synchronises anyway."""
        if self.symmetric_input and int(self._asym.item()) != 0:
            self._asym.zero_()
            raise ValueError("symmetric_input=True, but a batch held a matrix with x[i,j] != x[j,i] (or a NaN)")

    def reparameterize(self, mu, logvar, eps=None):
        return ops.reparameterize(mu, logvar, eps)

    def decode(self, z, adjacency):
        return self.decoder(z, adjacency)

    def forward(self, x, adjacency, eps=None, nodes_only=False):
        """(recon, mu, logvar) as in function/MM_Gate.py:170-177; with nodes_only the first entry is
        the decoder's node vectors [B,R,NP] (recon = sum_r g_r g_r^T), which the trainer's fused
        step turns into loss and gradients without writing the reconstruction out."""
        x = x.reshape(-1, self.n_nodes * self.n_nodes)
        mu, logvar = self.encode(x)
        z = self.reparameterize(mu, logvar, eps)
        if nodes_only:
            return self.decoder.node_vectors(z, adjacency), mu, logvar
        return self.decode(z, adjacency), mu, logvar


class MDNGateVAE(GCNVAE):
    """GCN-decoder VAE with a trait-gated Gaussian-mixture head.  BUILD-DEFINED: the reference
    has no mixture code; this is the model its name and data/trait.pt point at.  The R channel
    matrices of the GCN decoder are the component means, a linear gate on the subject's traits
    gives the mixture weights, and each component has one learned log-scale."""

    rank_one_sum_decoder = False        # its channels are mixture components, not summands

    def __init__(self, R=5, n_nodes=68, k=32, n_layers=3, latent_dim=68, n_traits=4, symmetric_input=False,
                 first_layer_gemm="split"):
        super().__init__(R=R, n_nodes=n_nodes, k=k, n_layers=n_layers, latent_dim=latent_dim,
                         symmetric_input=symmetric_input, first_layer_gemm=first_layer_gemm)
        self.gate = Linear(n_traits, R)
        self.log_sigma = nn.Parameter(torch.zeros(R))

    def forward(self, x, adjacency, traits, eps=None, nodes_only=False):
        """(component means [B,R,N*N], gate logits, mu, logvar); with nodes_only the first entry is
        the node vectors [B,R,NP] whose outer products the component means are, for
        `loss_from_nodes` / the trainer's fused step, which never write the matrices out."""
        x = x.reshape(-1, self.n_nodes * self.n_nodes)
        mu, logvar = self.encode(x)
        z = self.reparameterize(mu, logvar, eps)
        comp = self.decoder.node_vectors(z, adjacency) if nodes_only else self.decoder.components(z, adjacency)
        return comp, self.gate(traits), mu, logvar

    def loss(self, comp, gate_logits, x, mu, logvar):
        return ops.mdn_loss(comp, x, gate_logits, self.log_sigma, mu, logvar)

    def loss_from_nodes(self, nodes, gate_logits, x, mu, logvar):
        """`loss` of the component means g_r g_r^T without materialising them."""
        x = x.reshape(-1, self.n_nodes * self.n_nodes)
        return ops.mdn_outer_loss_terms(nodes, x, gate_logits, self.log_sigma, mu, logvar, self.n_nodes)[0]

    def expected(self, comp, gate_logits):
        """Mixture mean sum_k pi_k m_k [B, N*N] (a library batched GEMM over the K components)."""
        pi = ops.softmax_rows(gate_logits)
        return torch.bmm(pi.unsqueeze(1), comp).squeeze(1)


# ---------------------------------------------------------------------------------------
# multi-branch ResNet-MLP decoder family (function_fc_new/ResNet.py)
# ---------------------------------------------------------------------------------------
class ResidualBlock(nn.Module):
    """function_fc_new/ResNet.py:83-106: relu(fc2(relu(fc1 x)) + shortcut(x))."""

    def __init__(self, in_dim, out_dim):
        super().__init__()
        self.fc1 = Linear(in_dim, out_dim)
        self.fc2 = Linear(out_dim, out_dim)
        self.shortcut = Linear(in_dim, out_dim) if in_dim != out_dim else None

    def forward(self, x):
        out = self.fc2(self.fc1(x, act="relu"))
        identity = x if self.shortcut is None else self.shortcut(x)
        return ops.add_act(out, identity, "relu")


class BranchMLP(nn.Module):
    """function_fc_new/ResNet.py:109-129."""

    def __init__(self, in_dim=68 * 68, out_dim=68, decode_dimen_number=10, hidden_dim=256):
        super().__init__()
        self.init_fc = Linear(in_dim, hidden_dim)
        self.resblocks = nn.Sequential(*[ResidualBlock(hidden_dim, hidden_dim) for _ in range(decode_dimen_number)])
        self.final_fc = Linear(hidden_dim, out_dim)

    def forward(self, x):
        return self.final_fc(self.resblocks(self.init_fc(x, act="relu")))


class BranchGCN(nn.Module):
    """function_fc_new/ResNet.py:133-149: masked layers with identity joins."""

    def __init__(self, in_dim=68 * 68, n_nodes=68, num_gcn=10):
        super().__init__()
        self.init_fc = Linear(in_dim, n_nodes)
        self.gcn_layers = nn.ModuleList([GraphCNN(n_nodes, n_nodes) for _ in range(num_gcn)])

    def forward(self, x):
        h = self.init_fc(x, act="relu")
        for gcn in self.gcn_layers:
            h = ops.add_act(gcn(h), h, "relu")
        return h


class MultiBranchDecoder(nn.Module):
    """function_fc_new/ResNet.py:152-186: three branches summed, outer product, dense 4624^2 layer."""

    def __init__(self, decode_dimen_number_mlp1, hidden_dim_mlp1, decode_dimen_number_mlp2, hidden_dim_mlp2,
                 decode_dimen_number_gcn, n_nodes=68):
        super().__init__()
        D = n_nodes * n_nodes
        self.n_nodes = n_nodes
        self.branch_mlp1 = BranchMLP(D, n_nodes, decode_dimen_number_mlp1, hidden_dim_mlp1)
        self.branch_gcn = BranchGCN(D, n_nodes, decode_dimen_number_gcn)
        self.branch_mlp2 = BranchMLP(D, n_nodes, decode_dimen_number_mlp2, hidden_dim_mlp2)
        self.final_fc = Linear(D, D)

    def forward(self, x):
        merged = ops.add_act(ops.add_act(self.branch_mlp1(x), self.branch_gcn(x)), self.branch_mlp2(x))
        return self.final_fc(ops.outer_sum(merged.unsqueeze(1)))


class ResNetVAE(nn.Module):
    """`VAE` of function_fc_new/ResNet.py:189-238 (FNN towers, latent 256, multi-branch decoder)."""

    def __init__(self, latent_dim=256, decode_dimen_number_mlp1=8, hidden_dim_mlp1=512, decode_dimen_number_mlp2=8,
                 hidden_dim_mlp2=256, decode_dimen_number_gcn=10, n_nodes=68):
        super().__init__()
        D = n_nodes * n_nodes
        self.n_nodes = n_nodes
        self.fc11 = Linear(D, 1024)
        self.fc12 = Linear(D, 1024)
        self.fc111 = Linear(1024, 128)
        self.fc222 = Linear(1024, 128)
        self.fc21 = Linear(128, latent_dim)
        self.fc22 = Linear(128, latent_dim)
        self.fc_init_decode = Linear(latent_dim, D)
        self.multi_branch_decoder = MultiBranchDecoder(decode_dimen_number_mlp1, hidden_dim_mlp1,
                                                       decode_dimen_number_mlp2, hidden_dim_mlp2,
                                                       decode_dimen_number_gcn, n_nodes)

    def encode(self, x):
        h11 = self.fc111(self.fc11(x, act="relu"), act="relu")
        h12 = self.fc222(self.fc12(x, act="relu"), act="relu")
        return self.fc21(h11), self.fc22(h12)

    def reparameterize(self, mu, logvar, eps=None):
        return ops.reparameterize(mu, logvar, eps)

    def decode(self, z):
        return self.multi_branch_decoder(self.fc_init_decode(z))

    def forward(self, x, eps=None):
        mu, logvar = self.encode(x.reshape(-1, self.n_nodes * self.n_nodes))
        return self.decode(self.reparameterize(mu, logvar, eps)), mu, logvar


# ---------------------------------------------------------------------------------------
# conditional VAE (function_FC_SC/SC-FC.py)
# ---------------------------------------------------------------------------------------
class cVAE(nn.Module):
    def __init__(self, in_dim=68 * 68, latent_dim=64, hidden_dim=1024):
        super().__init__()
        self.in_dim, self.latent_dim = in_dim, latent_dim
        self.fc1 = Linear(in_dim * 2, hidden_dim)
        self.fc2 = Linear(hidden_dim, hidden_dim)
        self.fc_mu = Linear(hidden_dim, latent_dim)
        self.fc_logvar = Linear(hidden_dim, latent_dim)
        self.fc3 = Linear(latent_dim + in_dim, hidden_dim)
        self.fc4 = Linear(hidden_dim, hidden_dim)
        self.fc_out = Linear(hidden_dim, in_dim)

    def encode(self, fc, sc):
        h = self.fc2(self.fc1(torch.cat([fc, sc], dim=1), act="relu"), act="relu")
        return self.fc_mu(h), self.fc_logvar(h)

    def reparameterize(self, mu, logvar, eps=None):
        return ops.reparameterize(mu, logvar, eps)

    def decode(self, z, fc):
        h = self.fc4(self.fc3(torch.cat([z, fc], dim=1), act="relu"), act="relu")
        return self.fc_out(h, act="sigmoid")

    def forward(self, fc, sc, eps=None):
        mu, logvar = self.encode(fc, sc)
        z = self.reparameterize(mu, logvar, eps)
        return self.decode(z, fc), mu, logvar


# ---------------------------------------------------------------------------------------
# loss functions with the reference signatures
# ---------------------------------------------------------------------------------------
def loss_function_poisson_rate(recon_x, x, mu, logvar):
    """function/VAE_CNN_NN.py:197-200"""
    return ops.vae_loss(recon_x, x, mu, logvar, "poisson_rate")


def loss_function_poisson_log(recon_x, x, mu, logvar):
    """function/MM_Gate.py:179-182, function/VAE-CNN-GCN.py:201-204"""
    return ops.vae_loss(recon_x, x, mu, logvar, "poisson_log")


def loss_function_mse(recon_x, x, mu, logvar):
    """function_FC/VAE_CNN_GCN.py:207-211"""
    return ops.vae_loss(recon_x, x, mu, logvar, "mse")


def loss_function_mse_terms(recon_sc, sc, mu, logvar):
    """function_FC_SC/SC-FC.py:124-127: (total, reconstruction, KL)."""
    return ops.vae_loss_terms(recon_sc, sc, mu, logvar, "mse")
