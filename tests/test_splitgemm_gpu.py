"""The split-bf16 product of the two symmetric-input first layers (self.fc11 / self.fc12 of
function/MM_Gate.py:157-158): three bf16 parts per fp32 operand, six part products accumulated in fp32
on the bf16 matrix cores.  It must be as good as fp32 arithmetic: checked against float64 at the fp32
bars (values 1e-5, gradients 1e-4, relative to the largest entry) and, entry by entry, against the
error an fp32 accumulation is allowed (a few 1e-7 of sum |a b|)."""
import pytest
import torch

import mdnvae_amd as M
from mdnvae_amd import ops, _lib
from mdnvae_amd._lib import call, ptr, stream

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _operand(rows, k):
    return torch.empty(int(_lib.load().mdnvae_split_operand_bytes(rows, k)), dtype=torch.uint8, device=DEV)


def _up(v, m):
    return (v + m - 1) // m * m


@pytest.mark.parametrize("Mrows,N,K,tile", [(300, 512, 64, 256), (1000, 2048, 2368, 256), (257, 256, 16, 128),
                                            (3840, 512, 1000, 128)])
def test_split_product_against_float64(Mrows, N, K, tile):
    """out = a . b^T through mdnvae_split_rows / mdnvae_split_cols / mdnvae_split_gemm, with ragged row
    counts and k extents; entry-wise error against float64 relative to sum |a b|."""
    g = torch.Generator().manual_seed(Mrows + K)
    a = torch.randn(Mrows, K, generator=g) * torch.where(torch.rand(Mrows, K, generator=g) < 0.1, 30.0, 1.0)
    b = torch.randn(N, K, generator=g) * 0.02
    bias = torch.randn(N, generator=g) * 0.1
    Kp, Mp = _up(K, 16), _up(Mrows, tile)
    want = a.double() @ b.double().t()
    mag = a.double().abs() @ b.double().abs().t()
    ad, bd = a.to(DEV), b.to(DEV)
    # row-split operands need k a multiple of 16: pad the sources; the column-split form pads by itself
    a_pad = torch.zeros(Mrows, Kp, device=DEV)
    a_pad[:, :K] = ad
    b_pad = torch.zeros(N, Kp, device=DEV)
    b_pad[:, :K] = bd
    for how in ("rows", "cols"):
        oa, ob = _operand(Mp, Kp), _operand(N, Kp)
        if how == "rows":
            call("mdnvae_split_rows", ptr(a_pad), Mrows, Kp, Kp, 0, oa.data_ptr(), stream())
            call("mdnvae_split_rows", ptr(b_pad), N, Kp, Kp, 0, ob.data_ptr(), stream())
        else:                                     # from the transposed tensors [K][rows]
            at, bt = ad.t().contiguous(), bd.t().contiguous()
            ld_a, ld_b = _up(Mrows, 4), _up(N, 4)
            at_p = torch.zeros(K, ld_a, device=DEV)
            at_p[:, :Mrows] = at
            call("mdnvae_split_cols", ptr(at_p), K, Mrows, ld_a, 0, oa.data_ptr(), stream())
            call("mdnvae_split_cols", ptr(bt), K, N, ld_b, 0, ob.data_ptr(), stream())
        half = N // 2 if (N // 2) % 256 == 0 else N
        out0 = torch.full((Mrows, half), float("nan"), device=DEV)
        out1 = torch.full((Mrows, N - half), float("nan"), device=DEV) if half < N else None
        call("mdnvae_split_gemm", oa.data_ptr(), ob.data_ptr(), Mp, N, Kp, Mrows, half, None, None, 0, 0, ptr(out0), ptr(out1),
             half, half, tile, 6, stream())
        got = torch.cat([out0] + ([out1] if out1 is not None else []), dim=1).double().cpu()
        assert torch.isfinite(got).all()
        assert float(((got - want).abs() / mag).max()) <= 1e-6, how        # fp32-accumulation level, entry by entry
        assert float((got - want).abs().max() / want.abs().max()) <= 1e-5
        # three-term form (a1 b1 + a1 b2 + a2 b1): what it leaves out is at most 3 * 2^-16 of |a b| per product
        # (a bf16 part is within 2^-8 of what it rounds), a third of that in the root mean square
        call("mdnvae_split_gemm", oa.data_ptr(), ob.data_ptr(), Mp, N, Kp, Mrows, half, None, None, 0, 0, ptr(out0), ptr(out1),
             half, half, tile, 3, stream())
        got3 = torch.cat([out0] + ([out1] if out1 is not None else []), dim=1).double().cpu()
        assert float(((got3 - want).abs() / mag).max()) <= 3 * 2.0 ** -16 + 1e-6, how
        assert float((got3 - want).norm() / want.norm()) <= 3e-5, how      # gradients are held to 1e-4
    # bias + ReLU epilogue
    bd_ = bias.to(DEV)
    out0 = torch.empty(Mrows, half, device=DEV)
    out1 = torch.empty(Mrows, N - half, device=DEV) if half < N else None
    call("mdnvae_split_gemm", oa.data_ptr(), ob.data_ptr(), Mp, N, Kp, Mrows, half, ptr(bd_[:half].contiguous()),
         ptr(bd_[half:].contiguous()) if half < N else None, 1, 0, ptr(out0), ptr(out1), half, half, tile, 6, stream())
    got = torch.cat([out0] + ([out1] if out1 is not None else []), dim=1).double().cpu()
    want_r = torch.relu(want + bias.double())
    assert float((got - want_r).abs().max() / want_r.abs().max()) <= 1e-5


def _amax_slot():
    return torch.zeros(ops.ABSMAX_SLOTS, dtype=torch.int32, device=DEV)


@pytest.mark.parametrize("scale_a,scale_b", [(1.0, 0.02), (3e-6, 4e3), (7e4, 1e-7)])
@pytest.mark.parametrize("Mrows,N,K,tile", [(300, 512, 64, 256), (1000, 2048, 2368, 256), (3840, 512, 1000, 128)])
def test_two_part_fp16_product_against_float64(Mrows, N, K, tile, scale_a, scale_b):
    """The same product from two fp16 parts per operand (mdnvae_absmax / mdnvae_split_rows_f16 /
    mdnvae_split_cols_f16 / mdnvae_split_gemm_f16): three fp16 products instead of six bf16 ones.  Tensors of
    very different magnitudes (the power-of-two scaling must absorb them), a tenth of the entries 30 times
    larger than the rest; entry by entry against float64 at the bar the six-term product is held to."""
    g = torch.Generator().manual_seed(Mrows + K)
    a = torch.randn(Mrows, K, generator=g) * torch.where(torch.rand(Mrows, K, generator=g) < 0.1, 30.0, 1.0) * scale_a
    b = torch.randn(N, K, generator=g) * scale_b
    bias = torch.randn(N, generator=g) * 0.1 * scale_a * scale_b * K ** 0.5
    Kp, Mp = _up(K, 16), _up(Mrows, tile)
    want = a.double() @ b.double().t()
    mag = a.double().abs() @ b.double().abs().t()
    ad, bd = a.to(DEV), b.to(DEV)
    ma, mb = _amax_slot(), _amax_slot()
    call("mdnvae_absmax", ptr(ad), Mrows, K, K, 0.0, 1, ma.data_ptr(), stream())
    call("mdnvae_absmax", ptr(bd), N, K, K, 0.0, 1, mb.data_ptr(), stream())
    assert ops.absmax_value(ma) == float(a.abs().max()) and ops.absmax_value(mb) == float(b.abs().max())
    half = N // 2 if (N // 2) % 256 == 0 else N
    for how in ("rows", "cols"):
        oa, ob = _operand(Mp, Kp), _operand(N, Kp)
        if how == "rows":
            call("mdnvae_split_rows_f16", ptr(ad), Mrows, K, K, 0, ma.data_ptr(), oa.data_ptr(), stream())
            call("mdnvae_split_rows_f16", ptr(bd), N, K, K, 0, mb.data_ptr(), ob.data_ptr(), stream())
        else:
            at, bt = ad.t().contiguous(), bd.t().contiguous()
            call("mdnvae_split_cols_f16", ptr(at), None, K, Mrows, Mrows, 0, 0, ma.data_ptr(), oa.data_ptr(), stream())
            call("mdnvae_split_cols_f16", ptr(bt), None, K, N, N, 0, 0, mb.data_ptr(), ob.data_ptr(), stream())
        out0 = torch.full((Mrows, half), float("nan"), device=DEV)
        out1 = torch.full((Mrows, N - half), float("nan"), device=DEV) if half < N else None
        call("mdnvae_split_gemm_f16", oa.data_ptr(), ob.data_ptr(), ma.data_ptr(), mb.data_ptr(), Mp, N, Kp, Mrows, half,
             None, None, 0, 0, ptr(out0), ptr(out1), half, half, tile, -1, stream())
        got = torch.cat([out0] + ([out1] if out1 is not None else []), dim=1).double().cpu()
        assert torch.isfinite(got).all()
        assert float(((got - want).abs() / mag).max()) <= 1e-6, how
        assert float((got - want).abs().max() / want.abs().max()) <= 1e-5
    bd_ = bias.to(DEV)
    out0 = torch.empty(Mrows, half, device=DEV)
    out1 = torch.empty(Mrows, N - half, device=DEV) if half < N else None
    call("mdnvae_split_gemm_f16", oa.data_ptr(), ob.data_ptr(), ma.data_ptr(), mb.data_ptr(), Mp, N, Kp, Mrows, half,
         ptr(bd_[:half].contiguous()), ptr(bd_[half:].contiguous()) if half < N else None, 1, 0, ptr(out0), ptr(out1), half,
         half, tile, -1, stream())
    got = torch.cat([out0] + ([out1] if out1 is not None else []), dim=1).double().cpu()
    want_r = torch.relu(want + bias.double())
    assert float((got - want_r).abs().max() / want_r.abs().max()) <= 1e-5


def test_two_part_fp16_small_entries_and_shared_slots():
    """Entries far below the tensor's largest one lose low-part bits but never more than 2^-29 of the largest
    entry in absolute terms; a slot filled by several mdnvae_absmax calls (reset = 0) and a floor value holds
    the largest of all of them; an all-zero tensor gives zeros."""
    g = torch.Generator().manual_seed(5)
    Mrows, N, K = 256, 256, 512
    a = torch.randn(Mrows, K, generator=g) * torch.logspace(0, -9, K).unsqueeze(0)      # columns span nine decades
    b = torch.randn(N, K, generator=g)
    ad, bd = a.to(DEV), b.to(DEV)
    ma, mb = _amax_slot(), _amax_slot()
    call("mdnvae_absmax", ptr(ad[:100].contiguous()), 100, K, K, 0.0, 1, ma.data_ptr(), stream())
    call("mdnvae_absmax", ptr(ad[100:].contiguous()), Mrows - 100, K, K, 0.0, 0, ma.data_ptr(), stream())
    assert ops.absmax_value(ma) == float(a.abs().max())
    call("mdnvae_absmax", ptr(bd), N, K, K, 1e3, 1, mb.data_ptr(), stream())
    assert ops.absmax_value(mb) == 1e3
    oa, ob = _operand(Mrows, K), _operand(N, K)
    call("mdnvae_split_rows_f16", ptr(ad), Mrows, K, K, 0, ma.data_ptr(), oa.data_ptr(), stream())
    call("mdnvae_split_rows_f16", ptr(bd), N, K, K, 0, mb.data_ptr(), ob.data_ptr(), stream())
    out = torch.empty(Mrows, N, device=DEV)
    call("mdnvae_split_gemm_f16", oa.data_ptr(), ob.data_ptr(), ma.data_ptr(), mb.data_ptr(), Mrows, N, K, Mrows, N, None,
         None, 0, 0, ptr(out), None, N, N, 128, -1, stream())
    want = a.double() @ b.double().t()
    bound = 3 * 2.0 ** -22 * (a.double().abs() @ b.double().abs().t()) + 2.0 ** -28 * (
        float(a.abs().max()) * b.double().abs().sum(1).unsqueeze(0) + 1e3 * a.double().abs().sum(1).unsqueeze(1))
    assert bool(((out.double().cpu() - want).abs() <= bound).all())
    zero = torch.zeros(Mrows, K, device=DEV)
    call("mdnvae_absmax", ptr(zero), Mrows, K, K, 0.0, 1, ma.data_ptr(), stream())
    call("mdnvae_split_rows_f16", ptr(zero), Mrows, K, K, 0, ma.data_ptr(), oa.data_ptr(), stream())
    call("mdnvae_split_gemm_f16", oa.data_ptr(), ob.data_ptr(), ma.data_ptr(), mb.data_ptr(), Mrows, N, K, Mrows, N, None,
         None, 0, 0, ptr(out), None, N, N, 128, -1, stream())
    assert float(out.abs().max()) == 0.0


@pytest.mark.parametrize("B,R,ld,parts", [(1000, 3828, 3840, 2), (130, 2346, 2368, 3), (64, 40, 48, 2)])
def test_both_operands_of_a_batch_from_one_pass(B, R, ld, parts):
    """mdnvae_split_rows_f16_cols writes, byte for byte, what mdnvae_split_rows_f16 and mdnvae_split_cols_gated
    (with the row of ones) write from the same batch."""
    g = torch.Generator().manual_seed(B)
    src = torch.zeros(B, ld)
    src[:, :R] = torch.rand(B, R, generator=g) * 7.0
    src = src.to(DEV)
    mx = _amax_slot()
    call("mdnvae_absmax", ptr(src), B, ld, ld, 0.0, 1, mx.data_ptr(), stream())
    rows_a, rows_b = (torch.zeros(int(_lib.load().mdnvae_split_operand_bytes(_up(B, 32), ld)), dtype=torch.uint8, device=DEV)
                      for _ in range(2))
    cols_a, cols_b = (torch.zeros(int(_lib.load().mdnvae_split_operand_bytes(_up(R + 1, 32), _up(B, 16))), dtype=torch.uint8,
                                  device=DEV) for _ in range(2))
    call("mdnvae_split_rows_f16", ptr(src), B, ld, ld, 0, mx.data_ptr(), rows_a.data_ptr(), stream())
    call("mdnvae_split_cols_gated", ptr(src), None, B, R, ld, 1, parts, 0, cols_a.data_ptr(), stream())
    call("mdnvae_split_rows_f16_cols", ptr(src), B, R, ld, 1, parts, mx.data_ptr(), rows_b.data_ptr(), cols_b.data_ptr(), stream())
    assert torch.equal(rows_a, rows_b)
    assert torch.equal(cols_a, cols_b)
    # parts = 0: the column operand as two fp16 parts at the rows operand's scale, 2^14 in the unit row
    cols_a.zero_()
    cols_b.zero_()
    call("mdnvae_split_cols_f16", ptr(src), None, B, R, ld, 1, 0, mx.data_ptr(), cols_a.data_ptr(), stream())
    call("mdnvae_split_rows_f16_cols", ptr(src), B, R, ld, 1, 0, mx.data_ptr(), rows_b.data_ptr(), cols_b.data_ptr(), stream())
    assert torch.equal(rows_a, rows_b)
    assert torch.equal(cols_a, cols_b)


@pytest.mark.parametrize("scale_x,scale_g", [(1.0, 1.0), (2e-4, 3e-7), (900.0, 5e3)])
@pytest.mark.parametrize("B,R,H,accumulate", [(1000, 300, 256, 0), (4100, 2346, 512, 2)])
def test_weight_gradient_product_from_two_fp16_parts(B, R, H, accumulate, scale_x, scale_g):
    """x^T (with the unit row) times gated grad^T on the fp16 product: rows < R are the weight gradient, row R
    the column sums (bias gradient), whatever the magnitudes of the two tensors - the unit row holds 2^14, not
    ones times the scale - entry by entry against float64 at the fp32 dot-product level."""
    g = torch.Generator().manual_seed(B + R)
    ld = _up(R, 16)
    x = torch.zeros(B, ld)
    x[:, :R] = torch.rand(B, R, generator=g) * (torch.rand(B, R, generator=g) < 0.3) * scale_x
    grad = torch.randn(B, H, generator=g) * scale_g
    gate = torch.randn(B, H, generator=g)
    xd, gd, yd = x.to(DEV), grad.to(DEV), gate.to(DEV)
    mx, mg = _amax_slot(), _amax_slot()
    call("mdnvae_absmax", ptr(xd), B, ld, ld, 0.0, 1, mx.data_ptr(), stream())
    call("mdnvae_absmax", ptr(gd), B, H, H, 0.0, 1, mg.data_ptr(), stream())
    Kp = _up(B, 16)
    tile = 256 if accumulate == 2 else 128
    rows = R + 1
    Tp = _up(rows, tile)
    xb = torch.zeros(int(_lib.load().mdnvae_split_operand_bytes(Tp, Kp)), dtype=torch.uint8, device=DEV)
    pb = _operand(H, Kp)
    call("mdnvae_split_cols_f16", ptr(xd), None, B, R, ld, 1, 0, mx.data_ptr(), xb.data_ptr(), stream())
    call("mdnvae_split_cols_f16", ptr(gd), ptr(yd), B, H, H, 0, 0, mg.data_ptr(), pb.data_ptr(), stream())
    out = torch.zeros(rows, H, device=DEV)
    call("mdnvae_split_gemm_f16", xb.data_ptr(), pb.data_ptr(), mx.data_ptr(), mg.data_ptr(), Tp, H, Kp, rows, H, None, None,
         0, accumulate, ptr(out), None, H, H, tile, R, stream())
    gg = grad.double() * (gate > 0)
    want = torch.cat([x[:, :R].double().t() @ gg, gg.sum(0, keepdim=True)])
    mag = torch.cat([x[:, :R].double().abs().t() @ gg.abs(), gg.abs().sum(0, keepdim=True)])
    got = out.double().cpu()
    assert torch.isfinite(got).all()
    assert float(((got - want).abs() / mag.clamp_min(1e-300)).max()) <= 1e-6
    assert float((got[R] - want[R]).abs().max() / want[R].abs().max()) <= 2e-6          # the bias gradient row


@pytest.mark.parametrize("f16_product", [False, True])
@pytest.mark.parametrize("with_dense", [False, True])
@pytest.mark.parametrize("B,H,J", [(1000, 1024, 128), (130, 512, 64)])
def test_factored_gradient_operand_in_fp16_parts_and_its_bound(B, H, J, with_dense, f16_product):
    """mdnvae_dgrad_bound bounds |gpre . w + dense| from the factors' magnitudes (never below the true maximum,
    at most J times above it); mdnvae_split_cols_dgrad_f16 under that scale writes the operand
    mdnvae_split_cols_f16 writes from the materialised, gated product - to the rounding of the fp32 product itself."""
    g = torch.Generator().manual_seed(B + J)
    gpre = (torch.randn(B, J, generator=g) * 3e-4).to(DEV)
    w = (torch.randn(J, H, generator=g) / H ** 0.5).to(DEV)
    dense = (torch.randn(B, H, generator=g) * 1e-5).to(DEV) if with_dense else None
    gate = torch.randn(B, H, generator=g).to(DEV)
    mg, md, mb = _amax_slot(), _amax_slot(), _amax_slot()
    call("mdnvae_absmax", ptr(gpre), B, J, J, 0.0, 1, mg.data_ptr(), stream())
    if with_dense:
        call("mdnvae_absmax", ptr(dense), B, H, H, 0.0, 1, md.data_ptr(), stream())
    call("mdnvae_dgrad_bound", mg.data_ptr(), ptr(w), J, H, md.data_ptr() if with_dense else None, 1, mb.data_ptr(), stream())
    prod64 = gpre.double() @ w.double() + (dense.double() if with_dense else 0.0)
    true_max = float(prod64.abs().max())
    bound = ops.absmax_value(mb)
    assert true_max <= bound <= (J + 1) * true_max * 4
    Kp = _up(B, 16)
    got, want = _operand(H, Kp), _operand(H, Kp)
    # the product inside the pass: on the fp32 matrix instruction, or (with gpre's record) from two fp16 parts per factor
    call("mdnvae_split_cols_dgrad_f16", ptr(gpre), ptr(w), ptr(dense), ptr(gate), B, H, J, 0, mb.data_ptr(),
         mg.data_ptr() if f16_product else None, got.data_ptr(), stream())
    # the same operand from the float64 product rounded to fp32: parts differ only by the fp32 product's own rounding
    full = prod64.float().contiguous()
    call("mdnvae_split_cols_f16", ptr(full), ptr(gate), B, H, H, 0, 0, mb.data_ptr(), want.data_ptr(), stream())
    # read both operands back through the product with an identity-like probe: multiply by a random matrix
    probe = torch.randn(256, B, generator=g).to(DEV)
    mp = _amax_slot()
    call("mdnvae_absmax", ptr(probe), 256, B, B, 0.0, 1, mp.data_ptr(), stream())
    pa = _operand(256, Kp)
    call("mdnvae_split_rows_f16", ptr(probe), 256, B, B, 0, mp.data_ptr(), pa.data_ptr(), stream())
    outs = []
    for op in (got, want):
        out = torch.empty(256, H, device=DEV)
        call("mdnvae_split_gemm_f16", pa.data_ptr(), op.data_ptr(), mp.data_ptr(), mb.data_ptr(), 256, H, Kp, 256, H, None, None,
             0, 0, ptr(out), None, H, H, 128, -1, stream())
        outs.append(out.double().cpu())
    ref = probe.double().cpu() @ (prod64 * (gate > 0)).cpu()
    mag = probe.double().abs().cpu() @ (prod64.abs() * (gate > 0)).cpu()
    for out in outs:
        assert float(((out - ref).abs() / mag).max()) <= 2e-6


@pytest.mark.parametrize("wgrad", ["f16x2", "bf16x3", "bf16x6"])
@pytest.mark.parametrize("B,n", [(1024, 68), (1500, 87), (4200, 68)])
def test_first_layer_pair_against_float64_and_the_library_path(B, n, wgrad):
    """ops.sym_linear_pair (one split product for both layers, forward and weight gradients) against
    float64 on the full N*N inputs and against ops.sym_linear_act (library fp32 GEMMs).  With two fp16 parts
    (the default) or six bf16 terms the weight and bias gradients are as good as the fp32 library's; with three
    bf16 terms (Trainer(fast_wgrad=True)) they are held to 5e-5 of the largest entry, half the gradient bar."""
    with ops.arithmetic(wgrad=wgrad):
        _first_layer_pair_case(B, n, wgrad)


def _first_layer_pair_case(B, n, wgrad):
    torch.manual_seed(B)
    H, D = 1024, n * n
    x = M.data.synthetic_sc(B, n, seed=3, device=DEV)
    w = [(torch.randn(H, D) / D ** 0.5).to(DEV).requires_grad_(True) for _ in range(2)]
    b = [(torch.randn(H) * 0.05).to(DEV).requires_grad_(True) for _ in range(2)]
    up = [torch.randn(B, H, device=DEV) * (torch.rand(B, H, device=DEV) < 0.5) for _ in range(2)]
    asym = torch.zeros(1, dtype=torch.int32, device=DEV)
    xt = ops.tri_pack(x, n, asym)
    y = ops.sym_linear_pair(xt, w[0], b[0], w[1], b[1], n)
    torch.autograd.backward(list(y), up)
    assert int(asym.item()) == 0
    got = [t.detach().clone() for t in y] + [t.grad.clone() for t in w + b]
    for t in w + b:
        t.grad = None
    y_lib = [ops.sym_linear_act(xt, w[i], b[i], "relu", n) for i in range(2)]
    torch.autograd.backward(y_lib, up)
    lib = [t.detach() for t in y_lib] + [t.grad for t in w + b]
    # float64 reference with each path's ReLU gates taken from its own forward values: a pre-activation within
    # rounding of zero may land on either side, and one flipped gate moves a gradient entry by a whole term
    # (checked: the gates differ from float64's only where the float64 pre-activation is below 1e-5)
    x64 = x.double().cpu()
    w64 = [t.detach().double().cpu() for t in w]
    b64 = [t.detach().double().cpu() for t in b]
    pre64 = [x64 @ w64[i].t() + b64[i] for i in range(2)]

    def reference(outputs):
        gate = [(outputs[i].cpu() > 0) for i in range(2)]
        for i in range(2):
            flipped = gate[i] != (pre64[i] > 0)
            assert not flipped.any() or float(pre64[i][flipped].abs().max()) < 1e-5
        gpre64 = [up[i].double().cpu() * gate[i] for i in range(2)]
        return ([torch.relu(p_) for p_ in pre64] + [gpre64[i].t() @ x64 for i in range(2)]
                + [gpre64[i].sum(0) for i in range(2)])

    names = ["y1", "y2", "gw1", "gw2", "gb1", "gb2"]
    for name, g_, l_, w_, wl_ in zip(names, got, lib, reference(got), reference(lib)):
        bar = 1e-5 if name.startswith("y") else 1e-4
        err = float((g_.double().cpu() - w_).abs().max() / w_.abs().max())
        err_lib = float((l_.double().cpu() - wl_).abs().max() / wl_.abs().max())
        assert err <= bar, (name, err)
        if name.startswith("g") and wgrad == "bf16x3":      # weight and bias gradients come out of one product
            assert err <= 5e-5, (name, err)
        else:
            assert err <= max(4 * err_lib, 1e-6), (name, err, err_lib)  # no worse than the fp32 library GEMM


@pytest.mark.parametrize("Mrows,N,K", [(300, 333, 77), (513, 7569, 1000)])
def test_split_product_of_unaligned_tensors(Mrows, N, K):
    """Operands straight from tensors whose rows are neither 16-byte aligned nor a multiple of 16 long
    (the [B, N*N] connectomes with N*N odd), a ragged column count in the output, and accumulation."""
    g = torch.Generator().manual_seed(Mrows + N)
    a = torch.randn(Mrows, K, generator=g)
    b = torch.randn(N, K, generator=g) * 0.05
    want = a.double() @ b.double().t()
    ad, bd = a.to(DEV), b.to(DEV)
    Kp, Mp, Np = _up(K, 16), _up(Mrows, 128), _up(N, 256)
    oa, ob = _operand(Mp, Kp), torch.zeros(int(_lib.load().mdnvae_split_operand_bytes(Np, Kp)), dtype=torch.uint8, device=DEV)
    call("mdnvae_split_rows", ptr(ad), Mrows, K, K, 0, oa.data_ptr(), stream())
    call("mdnvae_split_rows", ptr(bd), N, K, K, 0, ob.data_ptr(), stream())
    base = torch.randn(Mrows, N, generator=g)
    out = base.to(DEV)
    call("mdnvae_split_gemm", oa.data_ptr(), ob.data_ptr(), Mp, Np, Kp, Mrows, N, None, None, 0, 1, ptr(out), None, Np, N, 128, 6, stream())
    got = out.double().cpu() - base.double()
    assert float((got - want).abs().max() / want.abs().max()) <= 1e-5
    # the column-split form of the same tensors: operand rows = columns of the [K, rows] transposes
    at, bt = ad.t().contiguous(), bd.t().contiguous()            # [K, Mrows], [K, N]: row strides 300 / 333 floats
    oa2 = torch.zeros_like(oa)
    ob2 = torch.zeros_like(ob)
    call("mdnvae_split_cols", ptr(at), K, Mrows, Mrows, 0, oa2.data_ptr(), stream())
    call("mdnvae_split_cols", ptr(bt), K, N, N, 0, ob2.data_ptr(), stream())
    out2 = torch.full((Mrows, N), float("nan"), device=DEV)
    call("mdnvae_split_gemm", oa2.data_ptr(), ob2.data_ptr(), Mp, Np, Kp, Mrows, N, None, None, 0, 0, ptr(out2), None, Np, N, 128, 6, stream())
    assert float((out2.double().cpu() - want).abs().max() / want.abs().max()) <= 1e-5


@pytest.mark.parametrize("B,n", [(1024, 68), (1100, 87)])
def test_full_first_layer_pair_against_float64_and_the_library_path(B, n):
    """ops.linear_pair: both first layers on all N*N inputs (no symmetry promise) as one split product."""
    torch.manual_seed(B)
    H, D = 1024, n * n
    x = torch.rand(B, D, device=DEV) * (torch.rand(B, D, device=DEV) < 0.3)
    w = [(torch.randn(H, D) / D ** 0.5).to(DEV).requires_grad_(True) for _ in range(2)]
    b = [(torch.randn(H) * 0.05).to(DEV).requires_grad_(True) for _ in range(2)]
    up = [torch.randn(B, H, device=DEV) * (torch.rand(B, H, device=DEV) < 0.5) for _ in range(2)]
    y = ops.linear_pair(x, w[0], b[0], w[1], b[1])
    torch.autograd.backward(list(y), up)
    got = [t.detach().clone() for t in y] + [t.grad.clone() for t in w + b]
    for t in w + b:
        t.grad = None
    y_lib = [ops.linear_act(x, w[i], b[i], "relu") for i in range(2)]
    torch.autograd.backward(y_lib, up)
    lib = [t.detach() for t in y_lib] + [t.grad for t in w + b]
    # float64 reference with the ReLU gate taken from the device's own forward values: a pre-activation
    # within rounding of zero may land on either side, and one flipped gate moves a gradient entry by a
    # whole term (checked: the gates differ only where the float64 pre-activation is below 1e-5)
    x64 = x.double().cpu()
    w64 = [t.detach().double().cpu() for t in w]
    b64 = [t.detach().double().cpu() for t in b]
    pre64 = [x64 @ w64[i].t() + b64[i] for i in range(2)]
    gate = [(got[i].cpu() > 0) for i in range(2)]
    for i in range(2):
        flipped = gate[i] != (pre64[i] > 0)
        assert float(pre64[i][flipped].abs().max()) < 1e-5 if flipped.any() else True
    gpre64 = [up[i].double().cpu() * gate[i] for i in range(2)]
    want = [torch.relu(p_) for p_ in pre64] + [gpre64[i].t() @ x64 for i in range(2)] + [gpre64[i].sum(0) for i in range(2)]
    for name, g_, l_, w_ in zip(["y1", "y2", "gw1", "gw2", "gb1", "gb2"], got, lib, want):
        bar = 1e-5 if name.startswith("y") else 1e-4
        err = float((g_.double().cpu() - w_).abs().max() / w_.abs().max())
        err_lib = float((l_.double().cpu() - w_).abs().max() / w_.abs().max())
        assert err <= bar, (name, err)
        assert err <= max(4 * err_lib, 1e-6), (name, err, err_lib)


@pytest.mark.parametrize("act", [None, "sigmoid", "relu"])
def test_large_dense_layer_on_the_split_product(act):
    """ops.linear_act sends a layer with B * n_out >= 6e6 through the split product (the N*N x N*N
    intercept of function/VAE_CNN_NN.py:187 at batch 1024; the cVAE layers at large batches): forward,
    data gradient and weight gradient against float64 at the fp32 bars, odd sizes on purpose."""
    torch.manual_seed(3)
    B, n_in, n_out = 1024, 4624, 6101
    assert B * n_out >= ops.SPLIT_LINEAR_MIN_OUTPUTS
    x = (torch.randn(B, n_in) * 0.5).to(DEV).requires_grad_(True)
    w = (torch.randn(n_out, n_in) / n_in ** 0.5).to(DEV).requires_grad_(True)
    b = (torch.randn(n_out) * 0.1).to(DEV).requires_grad_(True)
    up = torch.randn(B, n_out, device=DEV)
    calls = []
    orig = ops.call

    def spy(name, *a):
        calls.append(name)
        return orig(name, *a)

    ops.call = spy
    try:
        y = ops.linear_act(x, w, b, act)
        y.backward(up)
    finally:
        ops.call = orig
    # forward, data gradient, weight gradient (the first two on the two-part fp16 product by default)
    assert calls.count("mdnvae_split_gemm") + calls.count("mdnvae_split_gemm_f16") == 3
    x64, w64, b64 = (t.detach().double().cpu() for t in (x, w, b))
    pre = x64 @ w64.t() + b64
    want = {None: pre, "sigmoid": torch.sigmoid(pre), "relu": torch.relu(pre)}[act]
    yd = y.detach().double().cpu()
    slope = {None: torch.ones_like(pre), "sigmoid": yd * (1 - yd), "relu": (yd > 0).double()}[act]
    gpre = up.double().cpu() * slope                                   # through the device's own outputs (ReLU gates)
    checks = [("y", yd, want, 1e-5), ("gx", x.grad, gpre @ w64, 1e-4), ("gw", w.grad, gpre.t() @ x64, 1e-4),
              ("gb", b.grad, gpre.sum(0), 1e-4)]
    for name, got, ref, bar in checks:
        err = float((got.double().cpu() - ref).abs().max() / ref.abs().max())
        assert err <= bar, (name, err)


@pytest.mark.parametrize("B,n,J", [(1024, 68, 128), (1100, 87, 64)])
@pytest.mark.parametrize("wgrad", ["f16x2", "bf16x6"])
def test_next_layer_data_gradient_is_formed_inside_the_operand_split(B, n, J, wgrad, monkeypatch):
    """The dense layer after the first-layer pair leaves its data gradient as (grad_pre, weight); the pair's
    backward forms the product under the ReLU gate while it splits the operand (mdnvae_split_cols_dgrad).
    Same towers with the protocol switched off must give the same gradients to fp32 rounding, and the
    fused call must be the one that ran."""
    torch.manual_seed(B + J)
    H, D = 1024, n * n
    x = M.data.synthetic_sc(B, n, seed=5, device=DEV)
    w = [(torch.randn(H, D) / D ** 0.5).to(DEV).requires_grad_(True) for _ in range(2)]
    b = [(torch.randn(H) * 0.05).to(DEV).requires_grad_(True) for _ in range(2)]
    wa = [(torch.randn(J, H) / H ** 0.5).to(DEV).requires_grad_(True) for _ in range(2)]
    ba = [(torch.randn(J) * 0.05).to(DEV).requires_grad_(True) for _ in range(2)]
    up = [torch.randn(B, J, device=DEV) for _ in range(2)]
    extra = torch.randn(B, H, device=DEV) * 0.1              # a second consumer of the first tower's output
    asym = torch.zeros(1, dtype=torch.int32, device=DEV)
    xt = ops.tri_pack(x, n, asym)
    fused_call = "mdnvae_split_cols_dgrad_f16" if wgrad == "f16x2" else "mdnvae_split_cols_dgrad"

    def run(factored, with_extra):
        with ops.arithmetic(wgrad=wgrad, factored_dgrad=factored):
            return run_(with_extra)

    def run_(with_extra):
        for t in w + b + wa + ba:
            t.grad = None
        calls = []
        orig = ops.call

        def spy(name, *a):
            calls.append(name)
            return orig(name, *a)

        monkeypatch.setattr(ops, "call", spy)
        y = ops.sym_linear_pair(xt, w[0], b[0], w[1], b[1], n)
        h = [ops.linear_act(y[i], wa[i], ba[i], "relu") for i in range(2)]
        outs, ups = list(h), list(up)
        if with_extra:
            outs.append(y[0])
            ups.append(extra)
        torch.autograd.backward(outs, ups)
        monkeypatch.setattr(ops, "call", orig)
        return [t.grad.clone() for t in w + b + wa + ba], calls

    for with_extra in (False, True):
        got, calls = run(True, with_extra)
        want, calls_plain = run(False, with_extra)
        assert calls.count(fused_call) == 2 and fused_call not in calls_plain
        for g_, w_ in zip(got, want):
            assert float((g_ - w_).abs().max() / w_.abs().max()) <= 2e-6


def test_a_view_of_the_first_layer_output_does_not_take_the_factored_hand_off():
    """The dense layer after the pair may leave its data gradient as a product only when it consumes the pair's
    output tensor itself.  A row slice, a reshape or a detached alias starts at the same address but is another
    tensor: it must get the ordinary dense gradient (right values, no out-of-range reads), and an output that is
    observed (hook, retain_grad) must see its true gradient."""
    torch.manual_seed(9)
    B, n, H, J = 1024, 68, 1024, 128
    D = n * n
    x = M.data.synthetic_sc(B, n, seed=5, device=DEV)
    w = [(torch.randn(H, D) / D ** 0.5).to(DEV).requires_grad_(True) for _ in range(2)]
    b = [(torch.randn(H) * 0.05).to(DEV).requires_grad_(True) for _ in range(2)]
    wa = (torch.randn(J, H) / H ** 0.5).to(DEV).requires_grad_(True)
    wa_half = (torch.randn(J, H // 2) / H ** 0.5).to(DEV).requires_grad_(True)
    ba = (torch.randn(J) * 0.05).to(DEV).requires_grad_(True)
    asym = torch.zeros(1, dtype=torch.int32, device=DEV)
    xt = ops.tri_pack(x, n, asym)
    params = w + b + [wa, wa_half, ba]

    def run(consume, factored):
        for t in params:
            t.grad = None
        calls = []
        orig = ops.call

        def spy(name, *a):
            calls.append(name)
            return orig(name, *a)

        ops.call = spy
        try:
            with ops.arithmetic(factored_dgrad=factored):
                y = ops.sym_linear_pair(xt, w[0], b[0], w[1], b[1], n)
                seen = consume(y)
        finally:
            ops.call = orig
        return [None if t.grad is None else t.grad.clone() for t in params], calls, seen

    def sliced(y):                         # the first 600 rows: same address, fewer rows
        up = torch.randn(600, J, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1))
        torch.autograd.backward([ops.linear_act(y[0][:600], wa, ba, "relu"), y[1]], [up, torch.zeros_like(y[1])])

    def reshaped(y):                       # [2B, H/2]: same address, same bytes, other shape
        up = torch.randn(2 * B, J, device=DEV, generator=torch.Generator(device=DEV).manual_seed(2))
        torch.autograd.backward([ops.linear_act(y[0].view(2 * B, H // 2), wa_half, ba, "relu"), y[1]], [up, torch.zeros_like(y[1])])

    def hooked(y):                         # the output itself, but observed
        seen = []
        y[0].register_hook(lambda g: seen.append(g.detach().clone()))
        up = torch.randn(B, J, device=DEV, generator=torch.Generator(device=DEV).manual_seed(3))
        torch.autograd.backward([ops.linear_act(y[0], wa, ba, "relu"), y[1]], [up, torch.zeros_like(y[1])])
        return seen

    def retained(y):
        y[0].retain_grad()
        up = torch.randn(B, J, device=DEV, generator=torch.Generator(device=DEV).manual_seed(3))
        torch.autograd.backward([ops.linear_act(y[0], wa, ba, "relu"), y[1]], [up, torch.zeros_like(y[1])])
        return [y[0].grad.detach().clone()]

    def direct(y):                         # the case the hand-off is for
        up = torch.randn(B, J, device=DEV, generator=torch.Generator(device=DEV).manual_seed(3))
        torch.autograd.backward([ops.linear_act(y[0], wa, ba, "relu"), y[1]], [up, torch.zeros_like(y[1])])

    for consume, expect_fused in ((sliced, False), (reshaped, False), (hooked, False), (retained, False), (direct, True)):
        got, calls, seen = run(consume, True)
        want, calls_plain, seen_plain = run(consume, False)
        fused = calls.count("mdnvae_split_cols_dgrad_f16")
        assert fused == (1 if expect_fused else 0), (consume.__name__, fused)
        assert "mdnvae_split_cols_dgrad_f16" not in calls_plain
        for g_, w_ in zip(got, want):
            assert (g_ is None) == (w_ is None)
            if g_ is not None:
                assert float((g_ - w_).abs().max()) <= 2e-6 * float(w_.abs().max()), consume.__name__
        if seen:
            assert float(seen[0].abs().max()) > 0 and torch.equal(seen[0], seen_plain[0])
