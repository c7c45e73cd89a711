"""Every data-gradient convolution and instance-norm backward of the two-part encoder on a fixture, checked against
float64 on the same operands (decoded from their fp16 parts)."""
import sys, os, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import torch.nn.functional as F
import mdnvae_amd as M
from mdnvae_amd import ops
from conftest import load_fixture, fixture_params
from oracle import ref_eager as R
from test_conv_gpu import VARIANT_CTOR
variant = sys.argv[1]
fix = load_fixture(variant)
DEV = "cuda:0"
x = torch.from_numpy(fix["x"]).to(DEV); eps = torch.from_numpy(fix["eps"]).to(DEV)


def pow2_scale(v):
    if v == 0:
        return 1.0
    e = math.floor(math.log2(v))
    return 2.0 ** (14 - e)


def decode(t, ps, groups, B, PP, scale):
    """two-part channel-blocked tensor -> float64 [B, 8 groups, PP]"""
    n = groups * B * PP * 8
    hi = t[:n * 2].view(torch.float16).double()
    lo = t[ps:ps + n * 2].view(torch.float16).double()
    return ((hi + lo) / scale).view(groups, B, PP, 8).permute(1, 0, 3, 2).reshape(B, groups * 8, PP)


orig_conv, orig_bwd = ops._cb_conv, ops._cb_norm_bwd
W = {}


def conv(act, ps, wp, bias, B, cout, cin, kind, rows, pitch, PP, out16=False, parts=1, out_scale=1.0, amax_a=None, amax_w=None):
    out = orig_conv(act, ps, wp, bias, B, cout, cin, kind, rows, pitch, PP, out16, parts, out_scale, amax_a, amax_w)
    if kind == 2 and parts == 2:
        sc = pow2_scale(ops.absmax_value(amax_a))
        dc = decode(act, ps, cout // 8, B, PP, sc)[:, :, :(rows + 2) * pitch].reshape(B, cout, rows + 2, pitch)
        w = W[amax_w.data_ptr()].double().cpu()
        # out[b, ci, r, c] on the (H+2) x pitch grid = sum over co, taps of dc[b, co, r + 2 - ky .. ] : full correlation with flipped weights
        ref = F.conv2d(dc.cpu(), w.flip(2, 3).transpose(0, 1))          # valid conv of the zero-bordered gradient
        got = out[:, :, :rows * pitch].reshape(B, cin, rows, pitch)[:, :, :, :pitch - 2].double().cpu()
        err = float((got - ref[:, :, :rows, :pitch - 2]).norm() / ref.norm())
        print(f"  dgrad conv rows {rows}: vs float64 on its operands {err:.2e}   |dc|max {ops.absmax_value(amax_a):.3g}  largest decoded {float(dc.abs().max()):.3g}")
    return out


ops._cb_conv = conv


def norm_bwd(g, gmode, pitch_g, y, stats, pitch_y, relu, C, H, Wd, omode, PP, base=None, want_sum=False, parts=1, amax_dc=None):
    out = orig_bwd(g, gmode, pitch_g, y, stats, pitch_y, relu, C, H, Wd, omode, PP, base, want_sum, parts, amax_dc)
    if parts == 2 and gmode == 1 and omode == 0:
        B = y.shape[0]
        gp = g.double().cpu()[:, :, :(H + 2) * pitch_g].reshape(B, C, H + 2, pitch_g)[:, :, :, :Wd + 2].clone()
        # adjoint of reflection padding by one
        gp[:, :, 2] += gp[:, :, 0]; gp[:, :, H - 1] += gp[:, :, H + 1]
        gp = gp[:, :, 1:H + 1]
        gp[:, :, :, 2] += gp[:, :, :, 0]; gp[:, :, :, Wd - 1] += gp[:, :, :, Wd + 1]
        gi = gp[:, :, :, 1:Wd + 1]
        if base is not None:
            gi = gi + base.double().cpu()
        yy = y.double().cpu().reshape(B, C, -1)[:, :, :H * pitch_y].reshape(B, C, H, pitch_y)[:, :, :, :Wd]
        st = stats.double().cpu().view(B, C, 2)
        yh = (yy - st[:, :, 0, None, None]) * st[:, :, 1, None, None]
        da = gi * (yh > 0) if relu else gi
        m1 = da.mean((2, 3), keepdim=True); m2 = (da * yh).mean((2, 3), keepdim=True)
        ref = st[:, :, 1, None, None] * (da - m1 - yh * m2)
        sc = pow2_scale(ops.absmax_value(amax_dc))
        dc = decode(out[0], out[1], C // 8, B, PP, sc)[:, :, :(H + 4) * pitch_g].reshape(B, C, H + 4, pitch_g)
        got = dc[:, :, 2:2 + H, 2:2 + Wd].cpu()
        err = float((got - ref).norm() / ref.norm())
        border = float(dc.abs().sum() - got.abs().sum())
        e2 = ""
        if relu:
            i = int(yh.abs().argmin())
            e2 = f"  smallest |yhat| {float(yh.abs().flatten()[i]):.3g} with g there / |g| {float(gi.flatten()[i].abs() / gi.norm()):.3g}"
        if out[3] is not None:
            e2 = f"  stream sum {float((out[3].double().cpu() - gi).norm() / gi.norm()):.2e}"
        print(f"  norm bwd relu {int(relu)} base {int(base is not None)}: dc vs float64 on its operands {err:.2e}  border {border:.3g}"
              f"  |dc|max rec {ops.absmax_value(amax_dc):.4g} true {float(ref.abs().max()):.4g}{e2}")
    return out


ops._cb_norm_bwd = norm_bwd
orig_pack = ops._cb_pack


def pack(w, kind, dev, parts=1, amax=None):
    if amax is not None:
        W[amax.data_ptr()] = w.detach()
    return orig_pack(w, kind, dev, parts, amax)


ops._cb_pack = pack
model = M.ConvVAE(precision="fp32", **VARIANT_CTOR[variant]).eval()
model.load_state_dict(fixture_params(fix), strict=False)
model = model.to(DEV)
recon, mu, logvar, x_latent = model(x, eps=eps)
loss = M.vae_loss(recon, x, mu, logvar, R.loss_mode(variant))
loss.backward()
