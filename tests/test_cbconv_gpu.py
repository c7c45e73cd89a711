"""The encoder's kernels on channel-blocked 16-bit operands (csrc/cbconv.hip), each against eager PyTorch in
float64 on the same bf16-rounded operands (only the fp32 accumulation differs: bar 1e-5 of the largest entry, as
for the fp32 kernels), the normalisation passes against autograd (fp32 arithmetic: 2e-5), then the whole encoder
against the planar bf16 encoder it replaces."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import recipe
from conftest import load_fixture, fixture_params
import mdnvae_amd as M
from mdnvae_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rel(a, b):
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def _bf16(t):
    return t.to(torch.bfloat16).double()


def _blocked(planes, PP):
    """[B, C, rows, pitch] float -> channel-blocked bytes [C/8][B][PP][8] bf16 (zeros behind the plane)."""
    B, C, R, P = planes.shape
    flat = planes.reshape(B, C // 8, 8, R * P).permute(1, 0, 3, 2)
    out = torch.zeros(C // 8, B, PP, 8)
    out[:, :, :R * P] = flat
    return out


def _to_dev(blocked):
    t = blocked.to(torch.bfloat16).contiguous().view(torch.uint8).reshape(-1)
    return torch.cat([t, torch.zeros(4096, dtype=torch.uint8)]).to(DEV), blocked[0].numel() * 2 * blocked.shape[0]


def cb_reflect(a, PP):
    return _blocked(F.pad(a, (1, 1, 1, 1), mode="reflect"), PP)


def cb_zero2(g, PP):
    """zero border of 2 with row pitch W + 2: a row's two right zeros are the next row's two left zeros"""
    B, C, H, W = g.shape
    flat = torch.zeros(B, C, PP)
    for r in range(H):
        s = (r + 2) * (W + 2) + 2
        flat[:, :, s:s + W] = g[:, :, r]
    return flat.reshape(B, C // 8, 8, PP).permute(1, 0, 3, 2).contiguous()


def cb_zero_behind(g, PP):
    return _blocked(F.pad(g, (0, 1, 0, 1)), PP)


def cb_parity(a, PP):
    """four parity planes with a zero block row / column in front; group-plane order [chunk][plane][k half]"""
    B, C, H, W = a.shape
    oh, ow = (H + 1) // 2, (W + 1) // 2
    out = torch.zeros((C // 16) * 8, B, PP, 8)
    for py in range(2):
        for px in range(2):
            pl = torch.zeros(B, C, oh + 1, ow + 1)
            sub = a[:, :, py::2, px::2]
            pl[:, :, 1:1 + sub.shape[2], 1:1 + sub.shape[3]] = sub
            blk = _blocked(pl, PP)                       # [C/8][B][PP][8]
            for g8 in range(C // 8):
                out[(g8 // 2) * 8 + (py * 2 + px) * 2 + (g8 % 2)] = blk[g8]
    return out


def unblock(buf, groups, B, PP):
    """device bytes -> [groups][B][PP][8] float64"""
    n = groups * B * PP * 8
    return buf[:2 * n].view(torch.bfloat16).reshape(groups, B, PP, 8).double().cpu()


def planes_of(blocked, rows, pitch):
    """[C/8][B][PP][8] -> [B, C, rows, pitch]"""
    G8, B, PP, _ = blocked.shape
    return blocked[:, :, :rows * pitch].permute(1, 0, 3, 2).reshape(B, G8 * 8, rows, pitch)


@pytest.mark.parametrize("h", [22, 17])
@pytest.mark.parametrize("B", [1, 3])
def test_stride1_forward_dgrad_wgrad(h, B):
    C = 192
    g = torch.Generator().manual_seed(h * 10 + B)
    a = torch.randn(B, C, h, h, generator=g).relu()
    w = torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5
    b = torch.randn(C, generator=g)
    cot = torch.randn(B, C, h, h, generator=g)
    G = ops._CBGeom(87 if h == 22 else 68)
    assert G.h3 == h
    ap = F.pad(_bf16(a), (1, 1, 1, 1), mode="reflect").requires_grad_(True)
    wd = _bf16(w).requires_grad_(True)
    want = F.conv2d(ap, wd, b.double())
    (want * _bf16(cot)).sum().backward()

    act, _ = _to_dev(cb_reflect(a, G.pp_r))
    wf = ops._cb_pack(w.to(DEV), 0, DEV)
    y = ops._cb_conv(act, 0, wf, b.to(DEV), B, C, C, 0, h, G.pr, G.pp_r)
    got = y[:, :, :h * G.pr].reshape(B, C, h, G.pr)[:, :, :, :h]
    tag = f"h{h} B{B}"
    recipe.record("conv blocked bf16", "forward vs float64 on the rounded operands", tag, _rel(got, want), 1e-5)
    assert _rel(got, want) <= 1e-5
    y16 = ops._cb_conv(act, 0, wf, b.to(DEV), B, C, C, 0, h, G.pr, G.pp_r, out16=True)       # the same planes as fp16
    assert y16.dtype == torch.float16 and torch.equal(y16[:, :, :h * G.pr], y[:, :, :h * G.pr].to(torch.float16))

    dc, _ = _to_dev(cb_zero2(cot, G.pp_dr))
    wt = ops._cb_pack(w.to(DEV), 2, DEV)
    dp = ops._cb_conv(dc, 0, wt, None, B, C, C, 2, h + 2, G.pr, G.pp_dr)
    gotp = dp[:, :, :(h + 2) * G.pr].reshape(B, C, h + 2, G.pr)
    recipe.record("conv blocked bf16", "data gradient (padded grid) vs float64", tag, _rel(gotp, ap.grad), 1e-5)
    assert _rel(gotp, ap.grad) <= 1e-5

    gw = ops._cb_wgrad(dc, act, B, C, C, 1, G.pp_dr, G.pp_r, G.k_r, G.pr)
    recipe.record("conv blocked bf16", "weight gradient vs float64", tag, _rel(gw, wd.grad), 1e-5)
    assert _rel(gw, wd.grad) <= 1e-5


@pytest.mark.parametrize("cin,cout,h,n", [(48, 96, 87, 87), (48, 96, 68, 68), (96, 192, 44, 87), (96, 192, 34, 68)])
@pytest.mark.parametrize("B", [1, 3])
def test_stride2_forward_dgrad_wgrad(cin, cout, h, n, B):
    g = torch.Generator().manual_seed(h * 10 + B + cin)
    a = torch.randn(B, cin, h, h, generator=g).relu()
    w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    b = torch.randn(cout, generator=g)
    G = ops._CBGeom(n)
    first = cin == 48
    oh, pitch = (G.h2, G.p2) if first else (G.h3, G.p3)
    PPa, PPd = (G.pp_a1, G.pp_d2) if first else (G.pp_a2, G.pp_d3)
    kr = G.k_2 if first else G.k_3
    cot = torch.randn(B, cout, oh, oh, generator=g)
    ad = _bf16(a).requires_grad_(True)
    wd = _bf16(w).requires_grad_(True)
    want = F.conv2d(ad, wd, b.double(), stride=2, padding=1)
    assert want.shape[-1] == oh
    (want * _bf16(cot)).sum().backward()

    act, _ = _to_dev(cb_parity(a, PPa))
    wf = ops._cb_pack(w.to(DEV), 1, DEV)
    y = ops._cb_conv(act, 0, wf, b.to(DEV), B, cout, cin, 1, oh, pitch, PPa)
    got = y[:, :, :oh * pitch].reshape(B, cout, oh, pitch)[:, :, :, :oh]
    tag = f"{cin}->{cout} h{h} B{B}"
    recipe.record("conv blocked bf16", "stride-2 forward vs float64 on the rounded operands", tag, _rel(got, want), 1e-5)
    assert _rel(got, want) <= 1e-5

    dc, _ = _to_dev(cb_zero_behind(cot, PPd))
    wt = ops._cb_pack(w.to(DEV), 3, DEV)
    dp = ops._cb_conv(dc, 0, wt, None, B, cout, cin, 3, oh, pitch, PPd)          # [B, 4 cin, Qr]
    par = dp[:, :, :oh * pitch].reshape(B, 4, cin, oh, pitch).cpu()
    gin = torch.zeros(B, cin, 2 * oh, 2 * oh)
    for py in range(2):
        for px in range(2):
            gin[:, :, py::2, px::2] = par[:, py * 2 + px, :, :, :oh]
    gin = gin[:, :, :h, :h]
    recipe.record("conv blocked bf16", "stride-2 data gradient vs float64", tag, _rel(gin, ad.grad), 1e-5)
    assert _rel(gin, ad.grad) <= 1e-5

    gw = ops._cb_wgrad(dc, act, B, cout, cin, 4, PPd, PPa, kr, pitch)
    recipe.record("conv blocked bf16", "stride-2 weight gradient vs float64", tag, _rel(gw, wd.grad), 1e-5)
    assert _rel(gw, wd.grad) <= 1e-5


# ---- two fp16 parts per operand (the fp32-accurate mode): three products per multiply-add carry 22 bits of each
# operand, so the float64 reference runs on the unrounded fp32 operands; 2e-6 of the largest entry covers the
# parts' 3 * 2^-22 and the fp32 accumulation
TWO_PART_BAR = 2e-6


def _two_parts(blocked, scale):
    """[groups][B][PP][8] float -> device bytes [2][groups][B][PP][8] fp16 (high part, then what it left) + part stride"""
    v = blocked.float() * scale
    hi = v.to(torch.float16)
    lo = (v - hi.float()).to(torch.float16)
    t = torch.stack([hi, lo]).contiguous().view(torch.uint8).reshape(-1)
    return torch.cat([t, torch.zeros(4096, dtype=torch.uint8)]).to(DEV), hi.numel() * 2


def _record(t):
    """(record of the largest magnitude of a tensor, the power-of-two scale the kernels derive from it)"""
    rec = torch.zeros(ops.ABSMAX_SLOTS, dtype=torch.int32, device=DEV)
    ops._absmax(t.to(DEV).reshape(t.shape[0], -1).contiguous(), rec, reset=False)
    top = float(t.abs().max())
    assert ops.absmax_value(rec) == top
    return rec, 2.0 ** (14 - int(np.floor(np.log2(top))))


@pytest.mark.parametrize("h,B", [(22, 3), (17, 1)])
def test_stride1_two_fp16_parts(h, B):
    C = 192
    g = torch.Generator().manual_seed(h * 10 + B + 1)
    a = torch.randn(B, C, h, h, generator=g).relu() * 3.0
    w = torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5
    b = torch.randn(C, generator=g)
    cot = torch.randn(B, C, h, h, generator=g) * 37.0
    G = ops._CBGeom(87 if h == 22 else 68)
    ap = F.pad(a.double(), (1, 1, 1, 1), mode="reflect").requires_grad_(True)
    wd = w.double().requires_grad_(True)
    want = F.conv2d(ap, wd, b.double())
    (want * cot.double()).sum().backward()
    sa = ops._pow2_below(6 * h)
    act, pa = _two_parts(cb_reflect(a, G.pp_r), sa)
    wrec, _ = _record(w)
    wf = ops._cb_pack(w.to(DEV), 0, DEV, 2, wrec)
    y = ops._cb_conv(act, pa, wf, b.to(DEV), B, C, C, 0, h, G.pr, G.pp_r, parts=2, out_scale=1.0 / sa, amax_w=wrec)
    got = y[:, :, :h * G.pr].reshape(B, C, h, G.pr)[:, :, :, :h]
    tag = f"h{h} B{B}"
    recipe.record("conv blocked, two fp16 parts", "forward vs float64", tag, _rel(got, want), TWO_PART_BAR)
    assert _rel(got, want) <= TWO_PART_BAR

    grec, sg = _record(cot)
    dc, pd = _two_parts(cb_zero2(cot, G.pp_dr), sg)
    wt = ops._cb_pack(w.to(DEV), 2, DEV, 2, wrec)
    dp = ops._cb_conv(dc, pd, wt, None, B, C, C, 2, h + 2, G.pr, G.pp_dr, parts=2, amax_a=grec, amax_w=wrec)
    gotp = dp[:, :, :(h + 2) * G.pr].reshape(B, C, h + 2, G.pr)
    recipe.record("conv blocked, two fp16 parts", "data gradient (padded grid) vs float64", tag, _rel(gotp, ap.grad), TWO_PART_BAR)
    assert _rel(gotp, ap.grad) <= TWO_PART_BAR

    gw = ops._cb_wgrad(dc, act, B, C, C, 1, G.pp_dr, G.pp_r, G.k_r, G.pr, 2, 1.0 / sa, pd, pa, grec)
    recipe.record("conv blocked, two fp16 parts", "weight gradient vs float64", tag, _rel(gw, wd.grad), TWO_PART_BAR)
    assert _rel(gw, wd.grad) <= TWO_PART_BAR


@pytest.mark.parametrize("cin,cout,h,n,B", [(48, 96, 87, 87, 1), (48, 96, 68, 68, 3), (96, 192, 44, 87, 3), (96, 192, 34, 68, 1)])
def test_stride2_two_fp16_parts(cin, cout, h, n, B):
    g = torch.Generator().manual_seed(h * 10 + B + cin + 1)
    a = torch.randn(B, cin, h, h, generator=g).relu() * 2.0
    w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    b = torch.randn(cout, generator=g)
    G = ops._CBGeom(n)
    first = cin == 48
    oh, pitch = (G.h2, G.p2) if first else (G.h3, G.p3)
    PPa, PPd = (G.pp_a1, G.pp_d2) if first else (G.pp_a2, G.pp_d3)
    kr = G.k_2 if first else G.k_3
    cot = torch.randn(B, cout, oh, oh, generator=g) * 0.011
    ad = a.double().requires_grad_(True)
    wd = w.double().requires_grad_(True)
    want = F.conv2d(ad, wd, b.double(), stride=2, padding=1)
    (want * cot.double()).sum().backward()
    sa = ops._pow2_below(h)
    act, pa = _two_parts(cb_parity(a, PPa), sa)
    wrec, _ = _record(w)
    wf = ops._cb_pack(w.to(DEV), 1, DEV, 2, wrec)
    y = ops._cb_conv(act, pa, wf, b.to(DEV), B, cout, cin, 1, oh, pitch, PPa, parts=2, out_scale=1.0 / sa, amax_w=wrec)
    got = y[:, :, :oh * pitch].reshape(B, cout, oh, pitch)[:, :, :, :oh]
    tag = f"{cin}->{cout} h{h} B{B}"
    recipe.record("conv blocked, two fp16 parts", "stride-2 forward vs float64", tag, _rel(got, want), TWO_PART_BAR)
    assert _rel(got, want) <= TWO_PART_BAR

    grec, sg = _record(cot)
    dc, pd = _two_parts(cb_zero_behind(cot, PPd), sg)
    wt = ops._cb_pack(w.to(DEV), 3, DEV, 2, wrec)
    dp = ops._cb_conv(dc, pd, wt, None, B, cout, cin, 3, oh, pitch, PPd, parts=2, amax_a=grec, amax_w=wrec)
    par = dp[:, :, :oh * pitch].reshape(B, 4, cin, oh, pitch).cpu()
    gin = torch.zeros(B, cin, 2 * oh, 2 * oh)
    for py in range(2):
        for px in range(2):
            gin[:, :, py::2, px::2] = par[:, py * 2 + px, :, :, :oh]
    gin = gin[:, :, :h, :h]
    recipe.record("conv blocked, two fp16 parts", "stride-2 data gradient vs float64", tag, _rel(gin, ad.grad), TWO_PART_BAR)
    assert _rel(gin, ad.grad) <= TWO_PART_BAR

    gw = ops._cb_wgrad(dc, act, B, cout, cin, 4, PPd, PPa, kr, pitch, 2, 1.0 / sa, pd, pa, grec)
    recipe.record("conv blocked, two fp16 parts", "stride-2 weight gradient vs float64", tag, _rel(gw, wd.grad), TWO_PART_BAR)
    assert _rel(gw, wd.grad) <= TWO_PART_BAR


def _pitched(t, pitch, Q):
    """[B,C,H,W] -> [B,C,Q] with row pitch (scratch columns filled with a large finite value)"""
    B, C, H, W = t.shape
    out = torch.full((B, C, Q), 7.5e3)
    out[:, :, :H * pitch].reshape(B, C, H, pitch)[:, :, :, :W] = t
    return out


@pytest.mark.parametrize("C,H,relu,mode,y16", [(16, 22, True, 0, False), (8, 17, False, 0, False), (16, 44, True, 1, False),
                                               (16, 87, True, 1, False), (16, 34, True, 1, False), (16, 22, True, 0, True),
                                               (16, 44, True, 1, True), (8, 17, False, 0, True)])
def test_norm_forward_pass(C, H, relu, mode, y16):
    g = torch.Generator().manual_seed(C * H + mode)
    B = 3
    raw = torch.randn(B, C, H, H, generator=g) * 2 + 0.5
    if y16:
        raw = raw.half().float()              # the planes arrive as fp16: the pass sees exactly these values
    skip = torch.randn(B, C, H, H, generator=g)
    pitch = H + 2
    Q = ops._ru(H * pitch, 8)
    if y16:
        y = _pitched(raw, pitch, Q).half().to(DEV)
        st = None                             # fp16 planes: statistics come from the pass itself
    elif Q <= 2048:
        y = _pitched(raw, pitch, Q).to(DEV)
        st = ops._cb_stats(y, H, H, pitch)
    else:                                   # the stem's output: planes without a pitch, statistics by the planar kernel
        y, pitch = raw.to(DEV), H
        st = ops._plane_stats(y)
    mean = raw.double().mean(dim=(2, 3)).reshape(-1)
    rstd = 1.0 / torch.sqrt(raw.double().var(dim=(2, 3), unbiased=False) + 1e-5).reshape(-1)
    if st is not None:
        assert _rel(st[:, 0], mean) <= 1e-6 and _rel(st[:, 1], rstd) <= 1e-6
    want = F.instance_norm(raw.double(), eps=1e-5)
    if relu:
        want = want.relu()
    if mode == 0:
        PP = ops._ru((H + 2) * (H + 2) + 5, 4)
        act, _, so, st2 = ops._cb_norm_fwd(y, st, H, H, pitch, relu, 0, PP, skip=skip.to(DEV), want_stream=True)
        assert _rel(st2[:, 0], mean) <= 1e-6 and _rel(st2[:, 1], rstd) <= 1e-6
        want = want + skip.double()
        assert _rel(so, want) <= 2e-6
        got = unblock(act, C // 8, B, PP)
        ref = cb_reflect(want.float(), PP).to(torch.bfloat16).double()
        assert torch.equal(got[:, :, (H + 2) * (H + 2):], torch.zeros_like(got[:, :, (H + 2) * (H + 2):]))
    else:
        oh = (H + 1) // 2
        PP = ops._ru((oh + 1) * (oh + 1) + 3, 4)
        act, _, _, _ = ops._cb_norm_fwd(y, st, H, H, pitch, relu, 1, PP)
        got = unblock(act, (C // 8) * 4, B, PP)
        ref = cb_parity(want.float(), PP).to(torch.bfloat16).double()
    # bf16 rounding of values that differ in the last fp32 bit may land one bf16 step apart
    assert float((got - ref).abs().max()) <= 2.0 ** -7 * float(ref.abs().max())
    assert float((got != ref).double().mean()) < 1e-3


@pytest.mark.parametrize("y16", [False, True])
@pytest.mark.parametrize("C,H,relu,gmode,omode", [(16, 22, False, 0, 0), (16, 22, True, 1, 0), (8, 17, True, 1, 1),
                                                  (16, 44, True, 2, 1), (16, 34, True, 2, 1), (4, 87, True, 2, 2), (3, 68, True, 2, 2)])
def test_norm_backward_pass(C, H, relu, gmode, omode, y16):
    """dc against autograd through [pad / parity gather] + relu + instance norm; the gradient arrives the way the
    consumer's data gradient leaves it, plus (gmode 1) the residual stream's own gradient."""
    g = torch.Generator().manual_seed(C * H + gmode * 7 + omode)
    B = 2
    raw = torch.randn(B, C, H, H, generator=g) * 2 + 0.5
    if y16:
        if omode == 2:
            pytest.skip("the stem's planes stay fp32")
        raw = raw.half().float()
    rd = raw.double().requires_grad_(True)
    a = F.instance_norm(rd, eps=1e-5)
    if relu:
        a = a.relu()
    pitch_y = H + 3
    Qy = ops._ru(H * pitch_y, 8)
    y = _pitched(raw, pitch_y, Qy).to(DEV)
    st = ops._cb_stats(y, H, H, pitch_y) if Qy <= 2048 else ops._plane_stats(raw.to(DEV))
    if Qy > 2048:
        y, pitch_y = raw.to(DEV), H
    if y16:
        y = y.half()
    base = None
    if gmode == 0:
        cot = torch.randn(B, C, H, H, generator=g)
        (a * cot.double()).sum().backward()
        gt, pitch_g = cot.to(DEV), H
    elif gmode == 1:
        cotp = torch.randn(B, C, H + 2, H + 2, generator=g)
        base = torch.randn(B, C, H, H, generator=g)
        (F.pad(a, (1, 1, 1, 1), mode="reflect") * cotp.double()).sum().backward(retain_graph=True)
        (a * base.double()).sum().backward()
        pitch_g = H + 2
        gt = _pitched(cotp, pitch_g, ops._ru((H + 2) * pitch_g, 8)).to(DEV)
    else:
        oh = (H + 1) // 2
        pitch_g = oh + 1
        Qg = ops._ru(oh * pitch_g, 8)
        cot = torch.randn(B, C, H, H, generator=g)
        (a * cot.double()).sum().backward()
        full = torch.zeros(B, C, 2 * oh, 2 * oh)
        full[:, :, :H, :H] = cot
        par = torch.full((B, 4, C, Qg), 3.0e3)
        for py in range(2):
            for px in range(2):
                par[:, py * 2 + px, :, :oh * pitch_g].reshape(B, C, oh, pitch_g)[:, :, :, :oh] = full[:, :, py::2, px::2]
        gt = par.reshape(B, 4 * C, Qg).to(DEV)
    if omode == 0:
        PP = ops._ru((H + 4) * (H + 2) + 3, 4)
    elif omode == 1:
        PP = ops._ru((H + 1) * (H + 1) + 3, 4)
    else:
        PP = 0
    dc, _, gsum, total = ops._cb_norm_bwd(gt, gmode, pitch_g, y, st, pitch_y, relu, C, H, H, omode, PP,
                                       base=None if base is None else base.to(DEV), want_sum=base is not None)
    want = rd.grad
    if omode == 2:
        assert _rel(dc, want) <= 2e-5
    else:
        got = unblock(dc, C // 8, B, PP)
        ref = (cb_zero2(want.float(), PP) if omode == 0 else cb_zero_behind(want.float(), PP))
        assert float((got - ref.double()).abs().max()) <= 2.0 ** -7 * float(ref.abs().max())
        assert torch.equal(got == 0, ref.to(torch.bfloat16) == 0) or float(((got == 0) != (ref.to(torch.bfloat16) == 0)).double().mean()) < 1e-4
    assert float(gsum.abs().max()) <= 1e-3 * float(want.abs().sum())
    if base is not None:
        folded = torch.zeros(B, C, H, H, dtype=torch.float64, requires_grad=True)
        (F.pad(folded, (1, 1, 1, 1), mode="reflect") * cotp.double()).sum().backward()
        assert _rel(total, folded.grad + base.double()) <= 2e-6


@pytest.mark.parametrize("C,H,gmode,omode", [(16, 22, 1, 0), (16, 44, 2, 1), (4, 87, 2, 2)])
def test_norm_backward_pass_with_bf16_gradient_planes(C, H, gmode, omode):
    """The data gradient arriving as bf16 planes (mdnvae_cb_conv with out16 = 2): the pass gives what it gives on the
    same values handed over as fp32 -- bit for bit, the planes are only widened on load."""
    g = torch.Generator().manual_seed(C + H + gmode)
    B = 2
    raw = (torch.randn(B, C, H, H, generator=g) * 2 + 0.5)
    stem = omode == 2
    if not stem:
        raw = raw.half().float()
    pitch_y = H if stem else H + 1
    Qy = H * H if stem else ops._ru(H * pitch_y, 8)
    y = raw.to(DEV) if stem else _pitched(raw, pitch_y, Qy).to(DEV)
    st = ops._plane_stats(raw.to(DEV)) if stem else ops._cb_stats(y, H, H, pitch_y)
    if not stem:
        y = y.half()
    if gmode == 1:
        pitch_g = H + 2
        gt = _pitched(torch.randn(B, C, H + 2, H + 2, generator=g), pitch_g, ops._ru((H + 2) * pitch_g, 8))
        base = torch.randn(B, C, H, H, generator=g).to(DEV)
    else:
        oh = (H + 1) // 2
        pitch_g = oh + 1
        gt = torch.randn(B, 4 * C, ops._ru(oh * pitch_g, 8), generator=g)
        base = None
    gt = gt.to(torch.bfloat16).to(DEV)
    PP = 0 if stem else ops._ru(((H + 4) * (H + 2) if omode == 0 else (H + 1) * (H + 1)) + 3, 4)
    outs = []
    for planes in (gt, gt.float()):
        dc, _, gsum, total = ops._cb_norm_bwd(planes, gmode, pitch_g, y, st, pitch_y, True, C, H, H, omode, PP, base=base,
                                              want_sum=base is not None)
        if not stem:
            dc = dc[:(C // 8) * B * PP * 16]             # (the buffer ends in slack for clamped reads, never written)
        outs.append((dc.clone(), gsum.clone(), None if total is None else total.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    if base is not None:
        assert torch.equal(outs[0][2], outs[1][2])
    if stem:
        # omode 3: the same result rounded to bf16, planes a multiple of 8 apart; plane sums taken before the rounding
        dc16, _, gsum16, _ = ops._cb_norm_bwd(gt, gmode, pitch_g, y, st, pitch_y, True, C, H, H, 3, 0)
        assert dc16.dtype == torch.bfloat16 and dc16.shape == (B, C, ops._ru(H * H, 8))
        assert torch.equal(dc16[:, :, :H * H], outs[0][0].reshape(B, C, H * H).to(torch.bfloat16))
        assert torch.equal(gsum16, outs[0][1])


def unblock2(buf, part_stride, groups, B, PP, scale):
    """two fp16 parts on the device -> [groups][B][PP][8] float64 (high part + what it left, scale undone)"""
    n = groups * B * PP * 8
    hi = buf[:2 * n].view(torch.float16).double()
    lo = buf[part_stride:part_stride + 2 * n].view(torch.float16).double()
    return ((hi + lo) / scale).reshape(groups, B, PP, 8).cpu()


@pytest.mark.parametrize("C,H,mode", [(16, 22, 0), (16, 44, 1), (16, 87, 1)])
def test_norm_passes_with_two_fp16_parts(C, H, mode):
    """The normalisation passes writing their operand as two fp16 parts: the forward pass with a fixed scale, the
    backward pass with the scale of the largest |dc|, which its probe run leaves in a record.  Decoded, both match
    float64 at the bars of the fp32 arithmetic of the passes (2e-6 forward, 2e-5 backward) -- the parts themselves
    lose nothing at that level."""
    g = torch.Generator().manual_seed(C * H + mode + 5)
    B = 2
    raw = torch.randn(B, C, H, H, generator=g) * 2 + 0.5
    rd = raw.double().requires_grad_(True)
    a = F.instance_norm(rd, eps=1e-5).relu()
    pitch = H + 2
    Q = ops._ru(H * pitch, 8)
    if Q <= 2048:
        y = _pitched(raw, pitch, Q).to(DEV)
        st = ops._cb_stats(y, H, H, pitch)
    else:
        y, pitch = raw.to(DEV), H
        st = ops._plane_stats(y)
    scale = ops._pow2_below(H)
    if mode == 0:
        PP = ops._ru((H + 2) * (H + 2) + 5, 4)
        act, ps, _, _ = ops._cb_norm_fwd(y, st, H, H, pitch, True, 0, PP, parts=2, scale=scale)
        got = unblock2(act, ps, C // 8, B, PP, scale)
        ref = cb_reflect(a.detach().float(), PP).double()
    else:
        oh = (H + 1) // 2
        PP = ops._ru((oh + 1) * (oh + 1) + 3, 4)
        act, ps, _, _ = ops._cb_norm_fwd(y, st, H, H, pitch, True, 1, PP, parts=2, scale=scale)
        got = unblock2(act, ps, (C // 8) * 4, B, PP, scale)
        ref = cb_parity(a.detach().float(), PP).double()
    assert float((got - ref).abs().max()) <= 2e-6 * float(ref.abs().max())

    if H > 44:
        return                                  # the stem's planes: their backward pass leaves fp32 planes, not an operand
    # backward: the gradient arrives on the grid of this layer's own consumer
    if mode == 0:
        cotp = torch.randn(B, C, H + 2, H + 2, generator=g) * 91.0
        base = torch.randn(B, C, H, H, generator=g) * 55.0
        (F.pad(a, (1, 1, 1, 1), mode="reflect") * cotp.double()).sum().backward(retain_graph=True)
        (a * base.double()).sum().backward()
        gt = _pitched(cotp, H + 2, ops._ru((H + 2) * (H + 2), 8)).to(DEV)
        PPd = ops._ru((H + 4) * (H + 2) + 3, 4)
        rec = torch.zeros(ops.ABSMAX_SLOTS, dtype=torch.int32, device=DEV)
        dc, pd, _, total = ops._cb_norm_bwd(gt, 1, H + 2, y, st, pitch, True, C, H, H, 0, PPd, base=base.to(DEV), want_sum=True,
                                            parts=2, amax_dc=rec)
        ref = cb_zero2(rd.grad.float(), PPd).double()
    else:
        cot = torch.randn(B, C, H, H, generator=g) * 0.003
        (a * cot.double()).sum().backward()
        rec = torch.zeros(ops.ABSMAX_SLOTS, dtype=torch.int32, device=DEV)
        PPd = ops._ru((H + 1) * (H + 1) + 3, 4)
        dc, pd, _, _ = ops._cb_norm_bwd(cot.to(DEV), 0, H, y, st, pitch, True, C, H, H, 1, PPd, parts=2, amax_dc=rec)
        ref = cb_zero_behind(rd.grad.float(), PPd).double()
    top = ops.absmax_value(rec)
    assert abs(top - float(rd.grad.abs().max())) <= 2e-5 * top            # the probe run's record: the largest |dc|
    got = unblock2(dc, pd, C // 8, B, PPd, 2.0 ** (14 - int(np.floor(np.log2(top)))))
    assert float((got - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    assert torch.equal(got == 0, ref == 0) or float(((got == 0) != (ref == 0)).double().mean()) < 1e-4


@pytest.mark.parametrize("variant", ["cnn_gcn_fc", "cnn_gcn_sc"])
def test_blocked_encoder_against_planar_bf16(variant):
    """Same rounding points as the planar bf16 encoder (activations and output gradients to bf16 where they enter a 3x3
    convolution, weights to bf16) but other kernels and summation orders.  Values that differ in their last fp32 bits
    round to neighbouring bf16 numbers now and then, and thirteen layers amplify that, so the two bf16 encoders are
    not compared with each other but with the fp32 encoder: the blocked one may be at most 1.5 times as far from it
    as the planar one (outputs by largest entry, gradients by L2 norm)."""
    fix = load_fixture(variant)
    x = torch.from_numpy(fix["x"]).to(DEV)
    res = {}
    for precision in ("fp32", "bf16", "bf16_planar"):
        model = M.ConvVAE(decoder="gcn", intercept="diag", final="none", precision=precision).eval()
        model.load_state_dict(fixture_params(fix), strict=False)
        model = model.to(DEV)
        mu, lv = model.encode(x)
        ((mu * torch.from_numpy(fix["eps"]).to(DEV)).sum() + (lv ** 2).sum()).backward()
        res[precision] = (mu.detach(), lv.detach(), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None})
    ref, a, b = res["fp32"], res["bf16"], res["bf16_planar"]
    for i, name in enumerate(("mu", "logvar")):
        ea, eb = _rel(a[i], ref[i]), _rel(b[i], ref[i])
        recipe.record(fix, "blocked bf16 encoder vs fp32 mode", name, ea, 1.5 * eb, note=f"planar bf16: {eb:.1e}")
        assert ea <= 1.5 * eb, (name, ea, eb)
    for name, gref in ref[2].items():
        if name.startswith("ec.") and name.endswith(".bias"):
            continue
        ea = float((a[2][name] - gref).norm() / gref.norm())
        eb = float((b[2][name] - gref).norm() / gref.norm())
        recipe.record(fix, "blocked bf16 encoder vs fp32 mode, gradient L2", name, ea, 1.5 * eb, note=f"planar bf16: {eb:.1e}")
        assert ea <= 1.5 * eb, (name, ea, eb)


@pytest.mark.parametrize("parts", [1, 2])
def test_batched_weight_packing_equals_one_layer_at_a_time(parts):
    g = torch.Generator().manual_seed(parts)
    shapes = [(96, 48, 1), (192, 96, 1), (192, 192, 0), (192, 192, 2), (96, 48, 3), (192, 96, 3)]
    ws = [(torch.randn(co, ci, 3, 3, generator=g) * (0.1 + i)).to(DEV) for i, (co, ci, _) in enumerate(shapes)]
    kinds = [k for _, _, k in shapes]
    recs = None
    if parts == 2:
        recs = []
        for w in ws:
            rec = torch.zeros(ops.ABSMAX_SLOTS, dtype=torch.int32, device=DEV)
            ops._absmax(w.reshape(w.shape[0], -1), rec, reset=False)
            recs.append(rec)
    many = ops._cb_pack_many(ws, kinds, DEV, parts, recs)
    for i, (w, k) in enumerate(zip(ws, kinds)):
        one = ops._cb_pack(w, k, DEV, parts, None if recs is None else recs[i])
        assert torch.equal(many[i], one), i
