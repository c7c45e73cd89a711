"""Layer-by-layer comparison of the channel-blocked bf16 encoder with the planar bf16 encoder and fp32 (forward)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import torch
from conftest import load_fixture, fixture_params
import mdnvae_amd as M
from mdnvae_amd import ops

DEV = "cuda:0"
fix = load_fixture("cnn_gcn_fc")
model = M.ConvVAE(decoder="gcn", intercept="diag", final="none").eval()
model.load_state_dict(fixture_params(fix), strict=False)
model = model.to(DEV)
x = torch.from_numpy(fix["x"]).to(DEV).reshape(-1, 68 * 68)
B, n = x.shape[0], 68
wb = [(m.weight.detach(), m.bias.detach()) for m in model.conv_params()]
ws = [w for w, _ in wb]; bs = [b for _, b in wb]
layers = ops.encoder_layers(n, 5)
G = ops._CBGeom(n)


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max())


def planar(parts):
    out = {}
    mp = [ops._mul_parts(parts, ly) for ly in layers]
    packed = [ly.pack(w, DEV, False, m) for ly, w, m in zip(layers, ws, mp)]
    raw1 = layers[0].fwd(x, None, 0, packed[0][0], bs[0], B, mp[0]); st1 = ops._plane_stats(raw1)
    raw2 = layers[1].fwd(raw1, st1, 1, packed[1][0], bs[1], B, mp[1]); st2 = ops._plane_stats(raw2)
    raw3 = layers[2].fwd(raw2, st2, 1, packed[2][0], bs[2], B, mp[2])
    xk, st3 = ops._inorm_fwd(raw3, None, 1)
    out.update(raw1=raw1, raw2=raw2, raw3=raw3, x0=xk)
    for k in range(5):
        ra = layers[3 + 2 * k].fwd(xk, None, 0, packed[3 + 2 * k][0], bs[3 + 2 * k], B, mp[3 + 2 * k]); sa = ops._plane_stats(ra)
        rb = layers[4 + 2 * k].fwd(ra, sa, 1, packed[4 + 2 * k][0], bs[4 + 2 * k], B, mp[4 + 2 * k])
        xk, sb = ops._inorm_fwd(rb, xk, 0)
        out[f"ra{k}"] = ra; out[f"rb{k}"] = rb; out[f"x{k+1}"] = xk
    return out


def blocked():
    out = {}
    stem = layers[0]
    wstem, _ = stem.pack(ws[0], DEV, False)
    kinds = [1, 1] + [0] * 10
    wf = [ops._cb_pack(w, k, DEV) for w, k in zip(ws[1:], kinds)]
    raw1 = stem.fwd(x, None, 0, wstem, bs[0], B); st1 = ops._plane_stats(raw1)
    a1, _, _, _ = ops._cb_norm_fwd(raw1, st1, n, n, n, True, 1, G.pp_a1)
    y2 = ops._cb_conv(a1, 0, wf[0], bs[1], B, 96, 48, 1, G.h2, G.p2, G.pp_a1); st2 = ops._cb_stats(y2, G.h2, G.h2, G.p2)
    a2, _, _, _ = ops._cb_norm_fwd(y2, st2, G.h2, G.h2, G.p2, True, 1, G.pp_a2)
    y3 = ops._cb_conv(a2, 0, wf[1], bs[2], B, 192, 96, 1, G.h3, G.p3, G.pp_a2); st3 = ops._cb_stats(y3, G.h3, G.h3, G.p3)
    H = G.h3
    xp, _, xs, _ = ops._cb_norm_fwd(y3, st3, H, H, G.p3, True, 0, G.pp_r, want_stream=True)
    crop = lambda y, h, p: y[:, :, :h * p].reshape(B, -1, h, p)[:, :, :, :h]
    out.update(raw1=raw1, raw2=crop(y2, G.h2, G.p2), raw3=crop(y3, G.h3, G.p3), x0=xs)
    for k in range(5):
        ia, ib = 2 + 2 * k, 3 + 2 * k
        ya = ops._cb_conv(xp, 0, wf[ia], bs[ia + 1], B, 192, 192, 0, H, G.pr, G.pp_r); sa = ops._cb_stats(ya, H, H, G.pr)
        aa, _, _, _ = ops._cb_norm_fwd(ya, sa, H, H, G.pr, True, 0, G.pp_r)
        yb = ops._cb_conv(aa, 0, wf[ib], bs[ib + 1], B, 192, 192, 0, H, G.pr, G.pp_r); sb = ops._cb_stats(yb, H, H, G.pr)
        xp, _, xs, _ = ops._cb_norm_fwd(yb, sb, H, H, G.pr, False, 0, G.pp_r, skip=xs, want_stream=True)
        out[f"ra{k}"] = crop(ya, H, G.pr); out[f"rb{k}"] = crop(yb, H, G.pr); out[f"x{k+1}"] = xs
    return out


with torch.no_grad():
    f32 = planar(0)
    pl = planar(1)
    cb = blocked()
for k in f32:
    print(f"{k:6s} planar-bf16 vs fp32 {rel(pl[k], f32[k]):.2e}   blocked vs fp32 {rel(cb[k], f32[k]):.2e}   blocked vs planar {rel(cb[k], pl[k]):.2e}")
