"""MI355X-native training and sampling path for the brain-connectome VAE family.

Host side: Python on PyTorch-ROCm (device memory, streams, torch.distributed); compute:
hand-written gfx950 kernels in lib/libmdnvae_hip.so, reached through a C ABI
(include/mdnvae_hip.h).  The module surface mirrors the reference scripts'
(VAE.encode / reparameterize / decode / forward / set_mask, loss_function)."""
from . import _lib  # noqa: F401
from .ops import reparameterize, vae_loss, vae_loss_terms, vae_loss_and_grads, mdn_loss, mdn_loss_terms  # noqa: F401

from . import ops, data, fid, evaluate  # noqa: F401
from .fid import calculate_fid, FeatureStats, feature_stats, fid_from_stats, sqrtm_psd, sqrt_factor  # noqa: F401
from .trainer import Trainer, FlatBuffers, allreduce_sum  # noqa: F401
from .modules import (ConvVAE, GCNVAE, MDNGateVAE, cVAE, GraphCNN, Linear, GCNLayer, ChannelGCN, MultiChannelGCNDecoder,  # noqa: F401
                      DiagonalIntercept, loss_function_poisson_rate, loss_function_poisson_log, loss_function_mse,
                      loss_function_mse_terms, ResNetVAE, ResidualBlock, BranchMLP, BranchGCN, MultiBranchDecoder)

__all__ = ["reparameterize", "vae_loss", "vae_loss_terms", "vae_loss_and_grads", "ops", "ConvVAE", "GCNVAE", "MDNGateVAE", "cVAE",
           "GraphCNN", "Linear", "loss_function_poisson_rate", "loss_function_poisson_log", "loss_function_mse",
           "loss_function_mse_terms"]
