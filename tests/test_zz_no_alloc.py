"""mdnvae_mdn_outer_fwdbwd_static (the entry point that allocates nothing: its per-sample partial
sums live in a static device array inside the library) against mdnvae_mdn_outer_fwdbwd with a
caller-supplied workspace, through the C ABI: two calls in a row, the second on a smaller batch, so
that whatever the first call left in the shared array cannot leak into the second result."""
import pytest
import torch

from mdnvae_amd import _lib
from mdnvae_amd._lib import call, ptr, stream
from mdnvae_amd.ops import DEC_NP

pytestmark = pytest.mark.gpu

STATIC_MAX_B = 65536


def _inputs(B, K, N, L, seed, symmetric):
    g = torch.Generator(device="cuda").manual_seed(seed)
    dev = "cuda"
    nodes = torch.zeros(B, K, DEC_NP, device=dev)
    nodes[:, :, :N] = torch.rand(B, K, N, device=dev, generator=g)
    x = torch.rand(B, N, N, device=dev, generator=g)
    if symmetric:
        x = 0.5 * (x + x.transpose(1, 2))
    x = x.reshape(B, N * N).contiguous()
    logits = torch.randn(B, K, device=dev, generator=g)
    log_sigma = torch.randn(K, device=dev, generator=g) * 0.1
    mu = torch.randn(B, L, device=dev, generator=g)
    logvar = torch.randn(B, L, device=dev, generator=g) * 0.3
    return nodes, x, logits, log_sigma, mu, logvar


def _run(fname, inputs, N, symmetric, workspace):
    nodes, x, logits, log_sigma, mu, logvar = inputs
    B, K, _ = nodes.shape
    dev = nodes.device
    out = {
        "loss": torch.full((3,), float("nan"), device=dev),
        "gnodes": torch.full_like(nodes, float("nan")),
        "glogits": torch.full_like(logits, float("nan")),
        "gls_rows": torch.full_like(logits, float("nan")),
        "gmu": torch.full_like(mu, float("nan")),
        "glv": torch.full_like(mu, float("nan")),
        "resp": torch.full_like(logits, float("nan")),
    }
    args = [ptr(nodes), ptr(x), ptr(logits), ptr(log_sigma), B, K, N, ptr(mu), ptr(logvar), mu.shape[1],
            ptr(out["gnodes"]), ptr(out["glogits"]), ptr(out["gls_rows"]), ptr(out["gmu"]), ptr(out["glv"]),
            ptr(out["resp"]), int(symmetric)]
    if workspace:
        ws = torch.empty(_lib.load().mdnvae_mdn_workspace_bytes(B), dtype=torch.uint8, device=dev)
        args.append(ws.data_ptr())
    call(fname, *args, ptr(out["loss"]), stream())
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("symmetric", [True, False])
def test_static_scratch_entry_point_equals_workspace_entry_point(symmetric):
    N, K, L = 87, 5, 68
    for B, seed in ((1500, 3), (37, 4)):          # the second call is smaller: stale pairs stay behind it
        inputs = _inputs(B, K, N, L, seed, symmetric)
        want = _run("mdnvae_mdn_outer_fwdbwd", inputs, N, symmetric, workspace=True)
        got = _run("mdnvae_mdn_outer_fwdbwd_static", inputs, N, symmetric, workspace=False)
        for key in want:
            assert torch.isfinite(got[key]).all(), key
            # same kernels, same launch shapes, fixed-order sums: bit-identical
            assert torch.equal(got[key], want[key]), (B, key)


def test_static_scratch_entry_point_rejects_oversize_batch():
    inputs = _inputs(4, 2, 8, 4, 1, True)
    nodes, x, logits, log_sigma, mu, logvar = inputs
    lib = _lib.load()
    out = torch.zeros(3, device="cuda")
    dummy = ptr(out)
    rc = lib.mdnvae_mdn_outer_fwdbwd_static(ptr(nodes), ptr(x), ptr(logits), ptr(log_sigma), STATIC_MAX_B + 1, 2, 8, ptr(mu),
                                            ptr(logvar), 4, dummy, dummy, dummy, dummy, dummy, dummy, 1, ptr(out), stream())
    assert rc == -2
    assert b"static scratch" in lib.mdnvae_last_error()
    torch.cuda.synchronize()
