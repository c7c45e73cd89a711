"""CPU checks of the C-ABI boundary: the library loads without a GPU, exports every
symbol the header declares, and the host wrappers refuse anything but device tensors."""
import ctypes
import os

import pytest
import torch

import mdnvae_amd
from mdnvae_amd import _lib


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = [n for n, _, _ in _lib.declared_functions()]
    assert len(names) >= 8
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/mdnvae_hip.h but not exported"
    assert lib.mdnvae_version() >= 100
    assert lib.mdnvae_loss_workspace_bytes() > 0


def test_every_exported_mdnvae_symbol_is_declared():
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if " T " in ln and "mdnvae_" in ln and "_Z" not in ln}
    declared = {n for n, _, _ in _lib.declared_functions()}
    assert exported == declared, f"undeclared exports: {exported - declared}; missing: {declared - exported}"


def test_argument_rejection_needs_no_gpu():
    lib = _lib.load()
    # negative size is refused before anything touches the device
    rc = lib.mdnvae_reparam_fwd(None, None, None, None, -1, None)
    assert rc < 0 and b"negative" in lib.mdnvae_last_error()
    rc = lib.mdnvae_vae_loss_fwd(7, None, None, 0, 0, None, None, 0, None, None, None, None, None)
    assert rc < 0 and b"mode" in lib.mdnvae_last_error()


def test_shape_limits_of_the_dense_and_optimizer_entry_points_need_no_gpu():
    lib = _lib.load()
    # a weight wider than 8 on either side is not a narrow layer
    assert lib.mdnvae_narrow_linear_fwd(None, None, None, 4, 9, 4, 0, None, None) < 0
    assert b"narrow" in lib.mdnvae_last_error()
    assert lib.mdnvae_narrow_linear_wgrad(None, None, 0, 4, 4, None, None, None, 0, None) < 0      # empty batch
    assert lib.mdnvae_narrow_linear_workspace_floats() >= 64 * 72
    # decoder panel gradients: sizes first, then pointers
    assert lib.mdnvae_decoder_wgrad(None, None, 0, 20, None, None, None) < 0
    assert lib.mdnvae_decoder_wgrad(None, None, 64, 20, None, None, None) < 0 and b"null" in lib.mdnvae_last_error()
    assert lib.mdnvae_decoder_wgrad_workspace_floats(16384, 20) == 32 * 20 * 96 * 96
    assert lib.mdnvae_decoder_wgrad_workspace_floats(8, 20) == 20 * 96 * 96
    # folded Adam: a host step count is needed when no device coefficients are given
    assert lib.mdnvae_adam_step_folded(None, None, None, None, 1024, 87, 0, None, 1e-3, 0.9, 0.999, 1e-8, 1.0, None) < 0
    assert b"step" in lib.mdnvae_last_error()
    assert lib.mdnvae_adam_step_folded(None, None, None, None, 1024, 87, 1, None, 1e-3, 0.9, 0.999, 1e-8, 1.0, None) < 0
    assert b"null" in lib.mdnvae_last_error()
    # symmetric-target likelihood passes: same argument checks as the general form
    assert lib.mdnvae_mdn_outer_fwdbwd(None, None, None, None, 4, 9, 87, None, None, 0, None, None, None, None, None, None,
                                       1, None, None, None) < 0
    assert b"components" in lib.mdnvae_last_error()
    assert lib.mdnvae_outer_loss_fwdbwd(1, None, None, 4, 5, 96, None, None, 0, None, None, None, 1, None, None, None) < 0
    assert b"N = 96" in lib.mdnvae_last_error()


def test_two_part_fp16_and_fused_gradient_entry_points_check_their_arguments_without_a_gpu():
    lib = _lib.load()
    # the largest-magnitude record is required wherever fp16 parts are written or read
    assert lib.mdnvae_absmax(None, 4, 4, 4, 0.0, 1, None, None) < 0 and b"null" in lib.mdnvae_last_error()
    assert lib.mdnvae_absmax(None, 4, 8, 4, 0.0, 1, None, None) < 0 and b"shape" in lib.mdnvae_last_error()
    assert lib.mdnvae_split_rows_f16(None, 4, 16, 16, 0, None, None, None) < 0 and b"slot" in lib.mdnvae_last_error()
    assert lib.mdnvae_split_cols_f16(None, None, 16, 4, 4, 0, 0, None, None, None) < 0 and b"slot" in lib.mdnvae_last_error()
    assert lib.mdnvae_split_gemm_f16(None, None, None, None, 256, 256, 16, 256, 256, None, None, 0, 0, None, None, 256, 256,
                                     256, -1, None) < 0 and b"slot" in lib.mdnvae_last_error()
    assert lib.mdnvae_split_cols_dgrad_f16(None, None, None, None, 64, 64, 64, 0, None, None, None, None) < 0
    assert b"slot" in lib.mdnvae_last_error()
    assert lib.mdnvae_dgrad_bound(None, None, 64, 64, None, 1, None, None) < 0 and b"null" in lib.mdnvae_last_error()
    assert lib.mdnvae_tri_pack_absmax(None, 4, 87, 3840, None, None, 1, None, None) < 0 and b"slot" in lib.mdnvae_last_error()
    assert lib.mdnvae_tri_fold_absmax(None, 4, 87, 3840, None, 1, None, None) < 0 and b"slot" in lib.mdnvae_last_error()
    assert lib.mdnvae_split_rows_f16_cols(None, 64, 40, 48, 1, 4, None, None, None, None) < 0 and b"parts" in lib.mdnvae_last_error()
    assert lib.mdnvae_split_rows_f16_cols(None, 64, 50, 48, 1, 2, None, None, None, None) < 0 and b"shape" in lib.mdnvae_last_error()
    # weight + bias gradient of a small layer: the bias gradient is what the entry point is for
    assert lib.mdnvae_linear_wgrad_bias(None, None, 64, 8, 8, None, None, 0, None, 0, None) < 0
    assert b"grad_b" in lib.mdnvae_last_error()
    assert lib.mdnvae_linear_wgrad_workspace_floats(16384, 128, 1024) % 4 == 0
    # the per-sample terms' sum: rows and their column sums come together
    ws = ctypes.create_string_buffer(64)
    out = (ctypes.c_float * 3)()
    assert lib.mdnvae_mdn_finalize(ws, 4, out, None, 5, out, None) < 0 and b"together" in lib.mdnvae_last_error()
    assert lib.mdnvae_mdn_finalize(None, 4, out, None, 5, None, None) < 0


def test_host_wrappers_refuse_cpu_tensors():
    a = torch.zeros(2, 4)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        mdnvae_amd.vae_loss(a, a, a, a, "mse")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        mdnvae_amd.reparameterize(a, a, a)
