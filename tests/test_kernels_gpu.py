"""Numerics of every HIP kernel against the plain PyTorch fp32/fp64 reference of the same op.

Operands are asymmetric random tensors (a swapped row/column map in an MFMA epilogue passes on
symmetric data), shapes include non-multiples of the tile sizes, and every test runs the
native extension (native.require() raises if it is missing).
"""

import pytest
import torch

from distributed_tensorflow_amd.ops import native as nv
from distributed_tensorflow_amd.ops import reference as ref

pytestmark = pytest.mark.gpu

DEV = "cuda"


def rnd(*shape, scale=1.0, seed=None):
    g = torch.Generator(device="cpu")
    g.manual_seed(seed if seed is not None else (hash(shape) & 0xFFFF) + 1)
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


def close(a, b, tol=2e-4):
    a, b = a.double().cpu(), b.double().cpu()
    denom = b.abs().max().item() + 1e-12
    err = (a - b).abs().max().item() / denom
    assert err < tol, f"relative max error {err:.3e} (tol {tol})"


def bn_state_for(x, gamma, beta):
    c = x.shape[-1]
    flat = x.reshape(-1, c).double()
    sums = torch.cat([flat.sum(0), (flat * flat).sum(0)]).to(DEV)
    return nv.BnState(nv.stats_from_totals(sums), gamma, beta, flat.shape[0])


CONV_CASES = [
    # (B, H, W, Cin, Cout, ks, stride)
    (3, 32, 32, 3, 16, 3, 1),
    (3, 32, 32, 16, 16, 3, 1),
    (3, 32, 32, 16, 32, 3, 2),
    (3, 16, 16, 32, 32, 3, 1),
    (3, 16, 16, 32, 64, 3, 2),
    (5, 8, 8, 64, 64, 3, 1),
    (2, 28, 28, 1, 32, 5, 1),
    (2, 14, 14, 32, 64, 5, 1),
    (2, 12, 20, 16, 16, 3, 1),
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fprop_plain(case):
    b, h, w, cin, cout, ks, s = case
    x = rnd(b, h, w, cin)
    wt = rnd(ks, ks, cin, cout, scale=0.1)
    want = ref.conv2d_same(x, wt, s)
    got = nv.conv_fprop(x, nv.pack_weights(wt), ks=ks, cout=cout, stride=s)
    close(got, want)


@pytest.mark.parametrize("case", CONV_CASES[1:6])
def test_conv_fprop_bn_prologue_and_stats(case):
    b, h, w, cin, cout, ks, s = case
    x = rnd(b, h, w, cin, scale=2.0) + 0.3
    gamma = rnd(cin).abs() + 0.5
    beta = rnd(cin, scale=0.2)
    wt = rnd(ks, ks, cin, cout, scale=0.1)
    a = torch.relu(ref.batch_norm(x, gamma, beta))
    want = ref.conv2d_same(a, wt, s)
    sums = nv.new_stats(cout, DEV)
    got = nv.conv_fprop(x, nv.pack_weights(wt), ks=ks, cout=cout, stride=s, in_bn=bn_state_for(x, gamma, beta), out_sums=sums)
    close(got, want)
    flat = want.reshape(-1, cout).double()
    tot = nv.stats_total(sums)
    close(tot[:cout], flat.sum(0), 1e-5)
    close(tot[cout:], (flat * flat).sum(0), 1e-5)


@pytest.mark.parametrize("case", [(3, 32, 32, 16), (3, 16, 16, 32), (5, 8, 8, 64)])
def test_conv_fprop_split_operands_forward_and_data_gradient(case):
    """Opt-in split-operand kernel (csrc/kernels/conv_fprop_split.hip): BatchNorm prologue, statistics and
    every residual mode of the forward pass, and the data-gradient form (flipped filter, ReLU mask,
    backward sums), each against the same reference and tolerance as the exact-fp32 kernel."""
    b, h, w, c = case
    x = rnd(b, h, w, c, scale=2.0) + 0.3
    gamma, beta = rnd(c).abs() + 0.5, rnd(c, scale=0.2)
    wt = rnd(3, 3, c, c, scale=0.1)
    bn = bn_state_for(x, gamma, beta)
    a = torch.relu(ref.batch_norm(x, gamma, beta))
    base = ref.conv2d_same(a, wt, 1)
    wsp = nv.pack_split_weights(wt)
    sums = nv.new_stats(c, DEV)
    close(nv.conv_fprop(x, wsp, ks=3, cout=c, in_bn=bn, out_sums=sums, split=True), base)
    tot = nv.stats_total(sums)
    flat = base.reshape(-1, c).double()
    close(tot[:c], flat.sum(0), 1e-5)
    close(tot[c:], (flat * flat).sum(0), 1e-5)
    res = rnd(b, h, w, c, seed=7)
    close(nv.conv_fprop(x, wsp, ks=3, cout=c, in_bn=bn, res_mode=nv.RES_IDENT, res=res, split=True), base + res)
    g2, b2 = rnd(c, seed=8).abs() + 0.5, rnd(c, scale=0.3, seed=9)
    want = base + torch.relu(ref.batch_norm(res, g2, b2))
    close(nv.conv_fprop(x, wsp, ks=3, cout=c, in_bn=bn, res_mode=nv.RES_BNRELU, res=res, res_bn=bn_state_for(res, g2, b2), split=True), want)
    if c > 16:
        src = rnd(b, 2 * h, 2 * w, c // 2, seed=10)
        close(nv.conv_fprop(x, wsp, ks=3, cout=c, in_bn=bn, res_mode=nv.RES_SHORTCUT_A, res=src, split=True), base + ref.shortcut_a(src))
    # data gradient: g = 1[bn(xin) > 0] * (conv(dy, W') + extra), sums of g and g * xhat
    dy = rnd(b, h, w, c, seed=11)
    extra = rnd(b, h, w, c, seed=12)
    xin = rnd(b, h, w, c, scale=1.5, seed=13)
    xbn = bn_state_for(xin, g2, b2)
    wflip = wt.flip(0, 1).permute(0, 1, 3, 2).contiguous()
    pre = ref.batch_norm(xin, g2, b2)
    want_g = (ref.conv2d_same(dy, wflip, 1) + extra) * (pre > 0)
    gs_split, gs_exact = nv.new_stats(c, DEV), nv.new_stats(c, DEV)
    got = nv.conv_fprop(dy, nv.pack_split_weights(wt, flip=True), ks=3, cout=c, epi_mode=nv.EPI_MASK, mask_src=xin, mask_bn=xbn,
                        extra=extra, out_sums=gs_split, split=True)
    exact = nv.conv_fprop(dy, nv.pack_weights(wt, flip=True), ks=3, cout=c, epi_mode=nv.EPI_MASK, mask_src=xin, mask_bn=xbn,
                          extra=extra, out_sums=gs_exact)
    close(exact, want_g)
    close(got, want_g)
    close(nv.stats_total(gs_split), nv.stats_total(gs_exact), 1e-5)


def test_conv_fprop_residual_modes():
    b, h, w, c = 4, 16, 16, 32
    x = rnd(b, h, w, c)
    wt = rnd(3, 3, c, c, scale=0.1)
    res = rnd(b, h, w, c, seed=7)
    base = ref.conv2d_same(x, wt, 1)
    wpk = nv.pack_weights(wt)
    close(nv.conv_fprop(x, wpk, ks=3, cout=c, res_mode=nv.RES_IDENT, res=res), base + res)
    gamma, beta = rnd(c).abs() + 0.5, rnd(c, scale=0.3)
    want = base + torch.relu(ref.batch_norm(res, gamma, beta))
    close(nv.conv_fprop(x, wpk, ks=3, cout=c, res_mode=nv.RES_BNRELU, res=res, res_bn=bn_state_for(res, gamma, beta)), want)
    # option-A shortcut: source at twice the resolution with half the channels
    src = rnd(b, 2 * h, 2 * w, c // 2, seed=9)
    want = base + ref.shortcut_a(src)
    close(nv.conv_fprop(x, wpk, ks=3, cout=c, res_mode=nv.RES_SHORTCUT_A, res=src), want)


def test_conv_fprop_bias_relu():
    x = rnd(2, 14, 14, 32)
    wt = rnd(5, 5, 32, 64, scale=0.05)
    bias = rnd(64, scale=0.2)
    want = torch.relu(ref.conv2d_same(x, wt, 1) + bias)
    close(nv.conv_fprop(x, nv.pack_weights(wt), ks=5, cout=64, epi_mode=nv.EPI_BIAS_RELU, bias=bias), want)


@pytest.mark.parametrize("b", [1, 3, 70, 300])
def test_conv_stem_forward_statistics_and_filter_gradient(b):
    x = rnd(b, 32, 32, 3)
    wt = rnd(3, 3, 3, 16, scale=0.3).requires_grad_(True)
    want = ref.conv2d_same(x, wt, 1)
    sums = nv.new_stats(16)
    got = nv.conv_stem_fwd(x, wt.detach(), out_sums=sums)
    close(got, want.detach())
    tot = nv.stats_total(sums)
    flat = want.detach().reshape(-1, 16).double()
    close(tot[:16], flat.sum(0), tol=1e-5)
    close(tot[16:], (flat * flat).sum(0), tol=1e-5)
    dy = rnd(b, 32, 32, 16)
    want.backward(dy)
    slots = torch.zeros(nv.STAT_SLOTS, 512, device=DEV)
    nv.conv_stem_wgrad(x, dy, slots[0, :432].view(3, 3, 3, 16), slot_stride=512)
    close(slots.sum(0)[:432].view(3, 3, 3, 16), wt.grad, tol=5e-4)
    assert float(slots[:, 432:].abs().max()) == 0.0


@pytest.mark.parametrize("b,imgs", [(1, 1), (3, 1), (3, 2), (4, 2), (37, 0), (300, 1), (1030, 0)])
def test_conv5x5_tile_forward(b, imgs):
    """Whole-image 5x5 kernel (LeNet conv2 shape): bias + ReLU epilogue, odd batches so the last
    tile is partly empty, both images-per-tile variants."""
    assert nv.conv5x5_tile_supported(14, 14, 32, 64)
    x = rnd(b, 14, 14, 32)
    wt = rnd(5, 5, 32, 64, scale=0.05)
    bias = rnd(64, scale=0.3)
    want = torch.relu(ref.conv2d_same(x, wt, 1) + bias)
    got = nv.conv5x5_tile(x, nv.conv5x5_pack(wt), bias=bias, imgs=imgs)
    close(got, want)


@pytest.mark.parametrize("split", [False, True])
@pytest.mark.parametrize("b", [1, 5, 300, 1030])
def test_conv5x5_filter_gradient(b, split):
    """split=True: the bf16x6 form on the bf16 matrix pipe, held to the exact kernel's tolerance."""
    x = rnd(b, 14, 14, 32)
    dy = rnd(b, 14, 14, 64)
    wt = rnd(5, 5, 32, 64, scale=0.05).requires_grad_(True)
    ref.conv2d_same(x, wt, 1).backward(dy)
    slots = torch.zeros(nv.STAT_SLOTS, 25 * 32 * 64 + 256, device=DEV)
    nv.conv5x5_wgrad(x, dy, slots[0, :25 * 32 * 64].view(5, 5, 32, 64), slot_stride=slots.shape[1], split=split)
    close(slots.sum(0)[:25 * 32 * 64].view(5, 5, 32, 64), wt.grad, tol=5e-4)
    dw = torch.zeros(5, 5, 32, 64, device=DEV)
    nv.conv5x5_wgrad(x, dy, dw, split=split)
    close(dw, wt.grad, tol=5e-4)
    # per-workgroup partial results (no atomics) summed by one reduction; the buffer starts dirty
    ws = torch.full((nv.conv5x5_wgrad_parts(b), 25 * 32 * 64), 7.0, device=DEV)
    dw2 = torch.ones(5, 5, 32, 64, device=DEV)
    nv.conv5x5_wgrad(x, dy, dw2, partials=ws, split=split)
    close(dw2 - 1.0, wt.grad, tol=5e-4)


@pytest.mark.parametrize("b,imgs", [(1, 1), (3, 2), (7, 1), (300, 1), (515, 0)])
def test_conv5x5_forward_with_fused_pool(b, imgs):
    x = rnd(b, 14, 14, 32)
    wt = rnd(5, 5, 32, 64, scale=0.05)
    bias = rnd(64, scale=0.3)
    wpk = nv.conv5x5_pack(wt)
    act = nv.conv5x5_tile(x, wpk, bias=bias, imgs=imgs)
    want_p, want_arg = nv.maxpool2x2_fwd(act)
    p = torch.full((b, 7, 7, 64), -1.0, device=DEV)
    arg = torch.full((b, 7, 7, 64), 9, device=DEV, dtype=torch.uint8)
    nv.conv5x5_tile(x, wpk, bias=bias, imgs=imgs, pool_out=(p, arg))
    assert torch.equal(p, want_p) and torch.equal(arg, want_arg)
    # a second tile-loop iteration must see clean halos again: same result when run twice
    nv.conv5x5_tile(x, wpk, bias=bias, imgs=imgs, pool_out=(p, arg))
    assert torch.equal(p, want_p)


@pytest.mark.parametrize("split", [False, True])
@pytest.mark.parametrize("b", [1, 6, 270])
def test_conv5x5_backward_from_the_pooled_gradient(b, split):
    """Data and filter gradient of the second LeNet convolution computed straight from the
    gradient at the max pool (pool + ReLU backward formed while staging) equal the unfused chain."""
    x = rnd(b, 14, 14, 32)
    wt = rnd(5, 5, 32, 64, scale=0.05)
    bias = rnd(64, scale=0.3)
    act = nv.conv5x5_tile(x, nv.conv5x5_pack(wt), bias=bias)
    pooled, arg = nv.maxpool2x2_fwd(act)
    dp = rnd(b, 7, 7, 64)
    dy = nv.maxpool2x2_bwd(dp, arg, pooled, (b, 14, 14, 64), relu_mask=True)
    want_dx = nv.conv5x5_tile(dy, wt, flip=True)
    got_dx = nv.conv5x5_tile(None, wt, flip=True, pooled=(dp, arg, pooled))
    close(got_dx, want_dx, tol=1e-6)
    want_dw = torch.zeros(5, 5, 32, 64, device=DEV)
    nv.conv5x5_wgrad(x, dy, want_dw, split=split)
    ws = torch.empty(nv.conv5x5_wgrad_parts(b), 25 * 32 * 64, device=DEV)
    got_dw = torch.zeros(5, 5, 32, 64, device=DEV)
    nv.conv5x5_wgrad(x, None, got_dw, partials=ws, pooled=(dp, arg, pooled), split=split)
    close(got_dw, want_dw, tol=1e-5)


@pytest.mark.parametrize("b", [1, 5, 300])
def test_conv5x5_tile_data_gradient(b):
    dy = rnd(b, 14, 14, 64)
    wt = rnd(5, 5, 32, 64, scale=0.05)
    x = rnd(b, 14, 14, 32).requires_grad_(True)
    ref.conv2d_same(x, wt, 1).backward(dy)
    got = nv.conv5x5_tile(dy, wt, flip=True)
    close(got, x.grad)


@pytest.mark.parametrize("b", [1, 3, 37, 300, 1030])
def test_conv5x5_split_forward_data_gradient_and_fusions(b):
    """Split-operand (bf16x6) form of the whole-image 5x5 kernel: forward with bias + ReLU, forward with the
    fused max pool, data gradient, data gradient staged from the pooled gradient - each against the exact
    kernel's reference at the exact kernel's tolerance."""
    x = rnd(b, 14, 14, 32)
    wt = rnd(5, 5, 32, 64, scale=0.05)
    bias = rnd(64, scale=0.3)
    wf, wb = nv.conv5x5_split_pack(wt), nv.conv5x5_split_pack(wt, flip=True)
    want = torch.relu(ref.conv2d_same(x, wt, 1) + bias)
    act = nv.conv5x5_tile(x, wf, bias=bias, split=True)
    close(act, want)
    want_p, want_arg = nv.maxpool2x2_fwd(act)
    p = torch.full((b, 7, 7, 64), -1.0, device=DEV)
    arg = torch.full((b, 7, 7, 64), 9, device=DEV, dtype=torch.uint8)
    for _ in range(2):   # the second run must see clean halos again
        nv.conv5x5_tile(x, wf, bias=bias, pool_out=(p, arg), split=True)
        assert torch.equal(p, want_p) and torch.equal(arg, want_arg)
    dy = rnd(b, 14, 14, 64)
    xg = x.clone().requires_grad_(True)
    ref.conv2d_same(xg, wt, 1).backward(dy)
    close(nv.conv5x5_tile(dy, wb, flip=True, split=True), xg.grad)
    dp = rnd(b, 7, 7, 64)
    dy2 = nv.maxpool2x2_bwd(dp, arg, p, (b, 14, 14, 64), relu_mask=True)
    want_dx = nv.conv5x5_tile(dy2, wb, flip=True, split=True)
    got_dx = nv.conv5x5_tile(None, wb, flip=True, pooled=(dp, arg, p), split=True)
    close(got_dx, want_dx, tol=1e-6)


@pytest.mark.parametrize("case", [c for c in CONV_CASES if c[3] >= 16])
def test_conv_dgrad(case):
    b, h, w, cin, cout, ks, s = case
    x = rnd(b, h, w, cin).requires_grad_(True)
    wt = rnd(ks, ks, cin, cout, scale=0.1)
    y = ref.conv2d_same(x, wt, s)
    dy = rnd(*y.shape, seed=11)
    (want,) = torch.autograd.grad(y, x, dy)
    _, pb, _ = ref.same_pad(h, ks, s)
    wpk_t = nv.pack_weights(wt, flip=True)
    got = nv.conv_fprop(dy, wpk_t, ks=ks, cout=cin, stride=1, upsample=(s == 2), pad=ks - 1 - pb, out_hw=(h, w))
    close(got, want)


@pytest.mark.parametrize("shape", [(5, 32, 16, 32), (3, 16, 32, 64)])
@pytest.mark.parametrize("with_extra", [False, True])
def test_conv_dgrad_stride2_mask_epilogue(shape, with_extra):
    # data gradient of a stride-2 layer with the ReLU mask and BatchNorm-backward sums (parity-decomposed kernel)
    b, h, cin, cout = shape
    x_pre = rnd(b, h, h, cin, scale=1.5)
    gamma, beta = rnd(cin).abs() + 0.5, rnd(cin, scale=0.3)
    wt = rnd(3, 3, cin, cout, scale=0.1)
    a = torch.relu(ref.batch_norm(x_pre, gamma, beta)).requires_grad_(True)
    y = ref.conv2d_same(a, wt, 2)
    dy = rnd(*y.shape, seed=23)
    extra = rnd(b, h, h, cin, seed=29) if with_extra else None
    (da,) = torch.autograd.grad(y, a, dy)
    bn_x = ref.batch_norm(x_pre, gamma, beta)
    want = torch.where(bn_x > 0, da + (extra if with_extra else 0), torch.zeros_like(da))
    st = bn_state_for(x_pre, gamma, beta)
    gs = nv.new_stats(cin, DEV)
    _, pb, _ = ref.same_pad(h, 3, 2)
    got = nv.conv_fprop(dy, nv.pack_weights(wt, flip=True), ks=3, cout=cin, stride=1, upsample=True, pad=2 - pb, out_hw=(h, h),
                        epi_mode=nv.EPI_MASK, mask_src=x_pre, mask_bn=st, extra=extra, out_sums=gs)
    close(got, want)
    mean, var = ref.batch_moments(x_pre)
    xhat = (x_pre - mean) * torch.rsqrt(var + ref.BN_EPSILON)
    gt = nv.stats_total(gs)
    close(gt[:cin], want.reshape(-1, cin).double().sum(0), 1e-4)
    close(gt[cin:], (want * xhat).reshape(-1, cin).double().sum(0), 1e-4)


@pytest.mark.parametrize("shape", [(5, 32, 16), (7, 16, 32), (9, 8, 64), (64, 8, 64)])
@pytest.mark.parametrize("res_mode", ["ident", "bnrelu", "none"])
def test_conv_pingpong_forward(shape, res_mode):
    # 8-wave ping-pong kernel, forced: odd tile counts leave one wave group with fewer tiles
    b, h, c = shape
    x = rnd(b, h, h, c, scale=1.5) + 0.1
    gamma, beta = rnd(c).abs() + 0.5, rnd(c, scale=0.3)
    wt = rnd(3, 3, c, c, scale=0.1)
    a = torch.relu(ref.batch_norm(x, gamma, beta))
    want = ref.conv2d_same(a, wt, 1)
    kw = {}
    if res_mode == "ident":
        res = rnd(b, h, h, c, seed=31)
        want = want + res
        kw = dict(res_mode=nv.RES_IDENT, res=res)
    elif res_mode == "bnrelu":
        res = rnd(b, h, h, c, seed=31)
        g2, b2 = rnd(c, seed=32).abs() + 0.5, rnd(c, scale=0.3, seed=33)
        want = want + torch.relu(ref.batch_norm(res, g2, b2))
        kw = dict(res_mode=nv.RES_BNRELU, res=res, res_bn=bn_state_for(res, g2, b2))
    sums = nv.new_stats(c, DEV)
    got = nv.conv_fprop(x, nv.pack_weights(wt), ks=3, cout=c, in_bn=bn_state_for(x, gamma, beta), out_sums=sums, debug_skip=64, **kw)
    close(got, want)
    flat = want.reshape(-1, c).double()
    tot = nv.stats_total(sums)
    close(tot[:c], flat.sum(0), 1e-5)
    close(tot[c:], (flat * flat).sum(0), 1e-5)


@pytest.mark.parametrize("shape", [(5, 32, 16), (7, 16, 32), (9, 8, 64)])
@pytest.mark.parametrize("with_extra", [False, True])
def test_conv_pingpong_masked_dgrad(shape, with_extra):
    b, h, c = shape
    x_pre = rnd(b, h, h, c, scale=1.5)
    gamma, beta = rnd(c).abs() + 0.5, rnd(c, scale=0.3)
    wt = rnd(3, 3, c, c, scale=0.1)
    a = torch.relu(ref.batch_norm(x_pre, gamma, beta)).requires_grad_(True)
    y = ref.conv2d_same(a, wt, 1)
    dy = rnd(*y.shape, seed=41)
    extra = rnd(b, h, h, c, seed=43) if with_extra else None
    (da,) = torch.autograd.grad(y, a, dy)
    want = torch.where(ref.batch_norm(x_pre, gamma, beta) > 0, da + (extra if with_extra else 0), torch.zeros_like(da))
    gs = nv.new_stats(c, DEV)
    got = nv.conv_fprop(dy, nv.pack_weights(wt, flip=True), ks=3, cout=c, pad=1, out_hw=(h, h), epi_mode=nv.EPI_MASK,
                        mask_src=x_pre, mask_bn=bn_state_for(x_pre, gamma, beta), extra=extra, out_sums=gs, debug_skip=64)
    close(got, want)
    mean, var = ref.batch_moments(x_pre)
    xhat = (x_pre - mean) * torch.rsqrt(var + ref.BN_EPSILON)
    gt = nv.stats_total(gs)
    close(gt[:c], want.reshape(-1, c).double().sum(0), 1e-4)
    close(gt[c:], (want * xhat).reshape(-1, c).double().sum(0), 1e-4)


def _bwd_filter(w, split, c):
    """Data-gradient packing of w for the exact-fp32 kernel or its split-operand (bf16x6) twin."""
    if not split:
        return nv.pack_weights(w.detach(), flip=True)
    if not nv.hip().conv_bwd_split_supported(1, 8, 8, c) and not nv.hip().conv_bwd_split_supported(1, 16, 16, c) \
            and not nv.hip().conv_bwd_split_supported(1, 32, 32, c):
        pytest.skip("no split-operand backward kernel for this channel count")
    return nv.pack_split_weights(w.detach(), flip=True)


@pytest.mark.parametrize("split", [False, True])
@pytest.mark.parametrize("shape", [(5, 32, 16), (7, 16, 32), (9, 8, 64), (70, 8, 64), (300, 32, 16), (700, 16, 32)])
@pytest.mark.parametrize("with_extra", [False, True])
def test_conv_bwd_fused(shape, with_extra, split):
    # one kernel: masked data gradient + BatchNorm-backward sums + filter gradient (per-XCD copies)
    b, h, c = shape
    x_pre = rnd(b, h, h, c, scale=1.5) + 0.1
    gamma, beta = rnd(c).abs() + 0.5, rnd(c, scale=0.3)
    wt = rnd(3, 3, c, c, scale=0.1).requires_grad_(True)
    a = torch.relu(ref.batch_norm(x_pre, gamma, beta)).requires_grad_(True)
    y = ref.conv2d_same(a, wt, 1)
    dy = rnd(*y.shape, seed=51)
    extra = rnd(b, h, h, c, seed=53) if with_extra else None
    da, dw_want = torch.autograd.grad(y, (a, wt), dy)
    want = torch.where(ref.batch_norm(x_pre, gamma, beta) > 0, da + (extra if with_extra else 0), torch.zeros_like(da))
    n = dw_want.numel()
    slots = torch.zeros(nv.STAT_SLOTS, n + 64, device=DEV)
    gs = nv.new_stats(c, DEV)
    assert nv.conv_bwd_fused_supported(x_pre)
    # dynamic band scheduling on the larger case, static striding otherwise
    ctr = torch.zeros(nv.hip().BWD_TILE_COUNTERS, device=DEV, dtype=torch.int32) if b > 32 and not split else None
    got = nv.conv_bwd_fused(dy, x_pre, bn_state_for(x_pre, gamma, beta), _bwd_filter(wt, split, c),
                            slots[0, :n].view(3, 3, c, c), gs, slot_stride=n + 64, extra=extra, tile_counters=ctr, split=split)
    close(got, want)
    assert slots[:, n:].abs().max().item() == 0
    close(slots[:, :n].sum(0).view(dw_want.shape), dw_want)
    mean, var = ref.batch_moments(x_pre)
    xhat = (x_pre - mean) * torch.rsqrt(var + ref.BN_EPSILON)
    gt = nv.stats_total(gs)
    close(gt[:c], want.reshape(-1, c).double().sum(0), 1e-4)
    close(gt[c:], (want * xhat).reshape(-1, c).double().sum(0), 1e-4)


@pytest.mark.parametrize("split", [False, True])
@pytest.mark.parametrize("shape", [(5, 32, 16), (6, 16, 32), (9, 8, 64)])
def test_conv_bwd_fused_with_bn_backward_prologue(shape, split):
    # x -> bn1/relu -> conv1 -> y -> bn2/relu -> conv2 -> z. The backward of conv1 takes the masked
    # gradient at bn2's output and applies bn2's backward while staging it.
    b, h, c = shape
    x = rnd(b, h, h, c, scale=1.5) + 0.1
    g1, b1 = rnd(c).abs() + 0.5, rnd(c, scale=0.3)
    g2 = (rnd(c, seed=61).abs() + 0.5).requires_grad_(True)
    b2 = rnd(c, scale=0.3, seed=62).requires_grad_(True)
    w1 = rnd(3, 3, c, c, scale=0.1).requires_grad_(True)
    w2 = rnd(3, 3, c, c, scale=0.1, seed=63)
    a1 = torch.relu(ref.batch_norm(x, g1, b1)).requires_grad_(True)
    y = ref.conv2d_same(a1, w1, 1)
    u2 = ref.batch_norm(y, g2, b2)
    u2.retain_grad()
    z = ref.conv2d_same(torch.relu(u2), w2, 1)
    dz = rnd(*z.shape, seed=64)
    da1, dw1, dg2, db2 = torch.autograd.grad(z, (a1, w1, g2, b2), dz, retain_graph=True)
    (du2,) = torch.autograd.grad(z, u2, dz)                      # already masked by relu'
    want_g = torch.where(ref.batch_norm(x, g1, b1) > 0, da1, torch.zeros_like(da1))
    yd = y.detach()
    mean, var = ref.batch_moments(yd)
    yhat = (yd - mean) * torch.rsqrt(var + ref.BN_EPSILON)
    gs2 = nv.stats_from_totals(torch.cat([du2.reshape(-1, c).double().sum(0), (du2 * yhat).reshape(-1, c).double().sum(0)]))
    n = dw1.numel()
    slots = torch.zeros(nv.STAT_SLOTS, n, device=DEV)
    gs1 = nv.new_stats(c, DEV)
    dgam, dbet = torch.zeros(c, device=DEV), torch.zeros(c, device=DEV)
    got = nv.conv_bwd_fused(du2.contiguous(), x, bn_state_for(x, g1, b1), _bwd_filter(w1, split, c),
                            slots[0].view(3, 3, c, c), gs1, slot_stride=n, split=split,
                            dy_bn=(yd, bn_state_for(yd, g2.detach(), b2.detach()), gs2), dy_dgamma=dgam, dy_dbeta=dbet)
    close(got, want_g, 2e-4)
    close(slots.sum(0).view(dw1.shape), dw1, 2e-4)
    close(dgam, dg2, 1e-4)
    close(dbet, db2, 1e-4)


@pytest.mark.parametrize("split", [False, True])
@pytest.mark.parametrize("option_a", [False, True])
@pytest.mark.parametrize("shape", [(5, 32, 16), (6, 16, 32), (9, 8, 64), (300, 32, 16), (260, 16, 32)])
def test_conv_bwd_fused_prologue_with_shortcut_gradient(shape, option_a, split):
    # as above, and y also feeds an identity shortcut whose gradient `res` arrives from the block
    # after: the kernel stages dy = bn2_backward(du2) + res and writes it out as the block input gradient
    b, h, c = shape
    x = rnd(b, h, h, c, scale=1.5) + 0.1
    g1, b1 = rnd(c).abs() + 0.5, rnd(c, scale=0.3)
    g2 = (rnd(c, seed=61).abs() + 0.5).requires_grad_(True)
    b2 = rnd(c, scale=0.3, seed=62).requires_grad_(True)
    w1 = rnd(3, 3, c, c, scale=0.1).requires_grad_(True)
    w2 = rnd(3, 3, c, c, scale=0.1, seed=63)
    a1 = torch.relu(ref.batch_norm(x, g1, b1)).requires_grad_(True)
    y = ref.conv2d_same(a1, w1, 1)
    y.retain_grad()
    u2 = ref.batch_norm(y, g2, b2)
    z = ref.conv2d_same(torch.relu(u2), w2, 1)
    dz = rnd(*z.shape, seed=64)
    if option_a:
        if c == 64:
            pytest.skip("the 64-channel stage is the last one")
        # y also feeds the resolution-halving shortcut: 2x2 average pool, C/2 zero channels each side
        res = rnd(b, h // 2, h // 2, 2 * c, seed=65)
        pooled = torch.nn.functional.avg_pool2d(y.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1)
        short = torch.nn.functional.pad(pooled, (c // 2, c // 2))
        loss = (z * dz).sum() + (short * res).sum()
    else:
        res = rnd(*y.shape, seed=65)
        loss = (z * dz).sum() + (y * res).sum()
    da1, dw1, dg2, db2, dy_total = torch.autograd.grad(loss, (a1, w1, g2, b2, y), retain_graph=True)
    (du2,) = torch.autograd.grad((z * dz).sum(), u2)
    want_g = torch.where(ref.batch_norm(x, g1, b1) > 0, da1, torch.zeros_like(da1))
    yd = y.detach()
    mean, var = ref.batch_moments(yd)
    yhat = (yd - mean) * torch.rsqrt(var + ref.BN_EPSILON)
    gs2 = nv.stats_from_totals(torch.cat([du2.reshape(-1, c).double().sum(0), (du2 * yhat).reshape(-1, c).double().sum(0)]))
    n = dw1.numel()
    slots = torch.zeros(nv.STAT_SLOTS, n, device=DEV)
    gs1 = nv.new_stats(c, DEV)
    dgam, dbet = torch.zeros(c, device=DEV), torch.zeros(c, device=DEV)
    dy_out = torch.full_like(yd, 7.0)
    got = nv.conv_bwd_fused(du2.contiguous(), x, bn_state_for(x, g1, b1), _bwd_filter(w1, split, c),
                            slots[0].view(3, 3, c, c), gs1, slot_stride=n, split=split,
                            dy_bn=(yd, bn_state_for(yd, g2.detach(), b2.detach()), gs2), dy_dgamma=dgam, dy_dbeta=dbet,
                            dy_res=res, dy_out=dy_out, dy_res_option_a=option_a)
    close(dy_out, dy_total, 2e-4)
    close(got, want_g, 2e-4)
    close(slots.sum(0).view(dw1.shape), dw1, 2e-4)
    close(dgam, dg2, 1e-4)
    close(dbet, db2, 1e-4)


def test_conv_dgrad_mask_epilogue():
    b, h, w, cin, cout = 4, 16, 16, 32, 32
    x_pre = rnd(b, h, w, cin, scale=1.5)          # BN input of the layer below
    gamma, beta = rnd(cin).abs() + 0.5, rnd(cin, scale=0.3)
    wt = rnd(3, 3, cin, cout, scale=0.1)
    dy = rnd(b, h, w, cout, seed=3)
    extra = rnd(b, h, w, cin, seed=5)
    a = torch.relu(ref.batch_norm(x_pre, gamma, beta)).requires_grad_(True)
    y = ref.conv2d_same(a, wt, 1)
    (da,) = torch.autograd.grad(y, a, dy)
    bnv = ref.batch_norm(x_pre, gamma, beta)
    want = torch.where(bnv > 0, da + extra, torch.zeros_like(da))
    st = bn_state_for(x_pre, gamma, beta)
    gs = nv.new_stats(cin, DEV)
    got = nv.conv_fprop(dy, nv.pack_weights(wt, flip=True), ks=3, cout=cin, pad=1, out_hw=(h, w),
                        epi_mode=nv.EPI_MASK, mask_src=x_pre, mask_bn=st, extra=extra, out_sums=gs)
    close(got, want)
    mean, var = ref.batch_moments(x_pre)
    xhat = (x_pre - mean) * torch.rsqrt(var + ref.BN_EPSILON)
    gt = nv.stats_total(gs)
    close(gt[:cin], want.reshape(-1, cin).double().sum(0), 1e-4)
    close(gt[cin:], (want * xhat).reshape(-1, cin).double().sum(0), 1e-4)


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_wgrad(case):
    b, h, w, cin, cout, ks, s = case
    x = rnd(b, h, w, cin)
    wt = rnd(ks, ks, cin, cout, scale=0.1).requires_grad_(True)
    y = ref.conv2d_same(x, wt, s)
    dy = rnd(*y.shape, seed=13)
    (want,) = torch.autograd.grad(y, wt, dy)
    dw = torch.zeros(want.shape, device=DEV)
    nv.conv_wgrad(x, dy, dw, stride=s)
    close(dw, want)
    # several tiles per block exercises the persistent accumulation
    dw2 = torch.zeros(want.shape, device=DEV)
    nv.conv_wgrad(x, dy, dw2, stride=s, tiles_per_block=3)
    close(dw2, want)


@pytest.mark.parametrize("case", [c for c in CONV_CASES if c[5] == 3 and c[3] >= 16])
@pytest.mark.parametrize("co_tiles,tpb", [(0, 0), (1, 1), (2, 5)])
def test_conv_wgrad_xcd_slots(case, co_tiles, tpb):
    b, h, w, cin, cout, ks, s = case
    if co_tiles and (cout // 16) % co_tiles:
        pytest.skip("co_tiles does not divide the output tiles")
    b = 9
    x = rnd(b, h, w, cin, scale=1.5) + 0.2
    gamma, beta = rnd(cin).abs() + 0.5, rnd(cin, scale=0.3)
    a = torch.relu(ref.batch_norm(x, gamma, beta))
    wt = rnd(ks, ks, cin, cout, scale=0.1).requires_grad_(True)
    y = ref.conv2d_same(a, wt, s)
    dy = rnd(*y.shape, seed=19)
    (want,) = torch.autograd.grad(y, wt, dy)
    n = want.numel()
    slots = torch.zeros(nv.STAT_SLOTS, n + 64, device=DEV)
    dw = slots[0, :n].view(want.shape)
    nv.conv_wgrad(x, dy, dw, stride=s, in_bn=bn_state_for(x, gamma, beta), slot_stride=n + 64, co_tiles=co_tiles, tiles_per_block=tpb)
    assert slots[:, n:].abs().max().item() == 0
    close(slots[:, :n].sum(0).view(want.shape), want)


def test_conv_wgrad_bn_prologue():
    b, h, w, cin, cout = 4, 16, 16, 32, 32
    x = rnd(b, h, w, cin, scale=1.5) + 0.2
    gamma, beta = rnd(cin).abs() + 0.5, rnd(cin, scale=0.3)
    a = torch.relu(ref.batch_norm(x, gamma, beta))
    wt = rnd(3, 3, cin, cout, scale=0.1).requires_grad_(True)
    y = ref.conv2d_same(a, wt, 1)
    dy = rnd(*y.shape, seed=17)
    (want,) = torch.autograd.grad(y, wt, dy)
    dw = torch.zeros(want.shape, device=DEV)
    nv.conv_wgrad(x, dy, dw, stride=1, in_bn=bn_state_for(x, gamma, beta))
    close(dw, want)


def test_bn_stats_apply_and_backward():
    x = rnd(6, 16, 16, 32, scale=2.0) + 0.5
    c = 32
    gamma, beta = rnd(c).abs() + 0.5, rnd(c, scale=0.3)
    sums = nv.bn_stats(x)
    flat = x.reshape(-1, c).double()
    tot = nv.stats_total(sums)
    close(tot[:c], flat.sum(0), 1e-6)
    close(tot[c:], (flat * flat).sum(0), 1e-6)
    st = nv.BnState(sums, gamma, beta, flat.shape[0])
    close(nv.bn_apply(x, st, relu=True), torch.relu(ref.batch_norm(x, gamma, beta)))
    close(nv.bn_apply(x, st, relu=False), ref.batch_norm(x, gamma, beta))

    xr = x.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    y = torch.relu(ref.batch_norm(xr, gr, br))
    dy = rnd(*x.shape, seed=23)
    dx_want, dg_want, db_want = torch.autograd.grad(y, (xr, gr, br), dy)
    gs = nv.new_stats(c, DEV)
    g = nv.bn_relu_bwd_reduce(dy, x, st, gs, relu=True)
    dgamma, dbeta = torch.empty(c, device=DEV), torch.empty(c, device=DEV)
    res = rnd(*x.shape, seed=29)
    dx = nv.bn_bwd_apply(g, x, st, gs, dgamma=dgamma, dbeta=dbeta, res_mode=nv.BWD_RES_IDENT, res=res)
    close(dx, dx_want + res, 5e-4)
    close(dgamma, dg_want, 5e-4)
    close(dbeta, db_want, 5e-4)


def test_bn_bwd_apply_shortcut_a():
    b, h, w, c = 3, 16, 16, 16
    x = rnd(b, h, w, c)
    gamma, beta = rnd(c).abs() + 0.5, rnd(c, scale=0.3)
    st = bn_state_for(x, gamma, beta)
    dy_block = rnd(b, h // 2, w // 2, 2 * c, seed=31)
    xs = x.clone().requires_grad_(True)
    (want_sc,) = torch.autograd.grad(ref.shortcut_a(xs), xs, dy_block)
    gs = nv.new_stats(c, DEV)
    g = nv.bn_relu_bwd_reduce(torch.zeros_like(x), x, st, gs)
    dx = nv.bn_bwd_apply(g, x, st, gs, res_mode=nv.BWD_RES_SHORTCUT_A, res=dy_block)
    close(dx, want_sc)
    close(nv.shortcut_a_fwd(x), ref.shortcut_a(x))
    close(nv.shortcut_a_bwd(dy_block, tuple(x.shape)), want_sc)


@pytest.mark.parametrize("slots", [1, 8])
def test_head_fwd_bwd(slots):
    b, hw, c, k = 9, 8, 64, 10
    x = rnd(b, hw, hw, c, scale=1.5)
    gamma, beta = rnd(c).abs() + 0.5, rnd(c, scale=0.3)
    fc_w, fc_b = rnd(c, k, scale=0.2), rnd(k, scale=0.1)
    labels = torch.randint(0, k, (b,), device=DEV, dtype=torch.int32)
    xr = x.clone().requires_grad_(True)
    wr, br_ = fc_w.clone().requires_grad_(True), fc_b.clone().requires_grad_(True)
    gr, be = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    feat_ref = torch.relu(ref.batch_norm(xr, gr, be)).mean(dim=(1, 2))
    logits_ref = feat_ref @ wr + br_
    loss_ref = ref.softmax_xent_mean(logits_ref, labels)
    dx_w, dw_w, db_w, dg_w, dbe_w = torch.autograd.grad(loss_ref, (xr, wr, br_, gr, be))

    st = bn_state_for(x, gamma, beta)
    feat = torch.empty(b, c, device=DEV)
    logits = torch.empty(b, k, device=DEV)
    dlogits = torch.empty(b, k, device=DEV)
    loss = torch.zeros(slots, device=DEV)
    correct = torch.zeros(slots, device=DEV, dtype=torch.int32)
    nv.head_fwd(x, st, fc_w, fc_b, labels, feat, logits, dlogits, loss, correct)
    close(logits, logits_ref)
    close(loss.sum().reshape(1), loss_ref.reshape(1))
    assert int(correct.sum().item()) == ref.top1_hits(logits_ref, labels)

    g = torch.empty_like(x)
    gs = nv.new_stats(c, DEV)
    d_w, d_b = torch.zeros_like(fc_w), torch.zeros_like(fc_b)
    nv.head_bwd(x, st, fc_w, feat, dlogits, g, gs, d_w, d_b)
    dgamma, dbeta = torch.empty(c, device=DEV), torch.empty(c, device=DEV)
    dx = nv.bn_bwd_apply(g, x, st, gs, dgamma=dgamma, dbeta=dbeta)
    close(dx, dx_w, 5e-4)
    close(d_w, dw_w, 5e-4)
    close(d_b, db_w, 5e-4)
    close(dgamma, dg_w, 5e-4)
    close(dbeta, dbe_w, 5e-4)


def test_softmax_xent_and_top1():
    b, k = 300, 10
    logits = rnd(b, k, scale=3.0)
    labels = torch.randint(0, k, (b,), device=DEV, dtype=torch.int32)
    lr = logits.clone().requires_grad_(True)
    want = ref.softmax_xent_mean(lr, labels)
    (dwant,) = torch.autograd.grad(want, lr)
    loss = torch.zeros(1, device=DEV)
    dl = torch.empty_like(logits)
    correct = torch.zeros(1, device=DEV, dtype=torch.int32)
    nv.softmax_xent(logits, labels, loss, dl, correct)
    close(loss, want.reshape(1))
    close(dl, dwant)
    hits = ref.top1_hits(logits, labels)
    assert int(correct.item()) == hits
    c2 = torch.zeros(1, device=DEV, dtype=torch.int32)
    nv.top1_count(logits, labels, c2)
    assert int(c2.item()) == hits


@pytest.mark.parametrize("n", [1024, 1663370, 269722 + 3])
def test_fused_adam_and_momentum(n):
    p, g = rnd(n), rnd(n, seed=2)
    m, v = rnd(n, scale=0.1, seed=3), rnd(n, scale=0.1, seed=4).abs()
    lr_t = ref.adam_lr_t(1e-3, 7)
    hyper = torch.tensor([lr_t, 0.9, 0.999, 1e-8, 0.125, 0.9, 2e-4, 0.0], device=DEV)
    pr, mr, vr = p.clone(), m.clone(), v.clone()
    ref.adam_update(pr, g * 0.125, mr, vr, lr_t)
    nv.fused_adam(p, g, m, v, hyper)
    close(p, pr, 1e-5)
    close(m, mr, 1e-5)
    close(v, vr, 1e-5)
    p2, m2 = rnd(n, seed=5), rnd(n, scale=0.1, seed=6)
    pr2, mr2 = p2.clone(), m2.clone()
    ref.momentum_update(pr2, g * 0.125, mr2, lr_t, 0.9, 2e-4)
    nv.fused_momentum_sgd(p2, g, m2, hyper)
    close(p2, pr2, 1e-5)
    close(m2, mr2, 1e-5)


def test_flat_ops_and_elementwise():
    n = 100003
    x = rnd(n)
    want = x * 0.25
    close(nv.flat_scale(x.clone(), 0.25), want)
    sd = torch.tensor([0.25], device=DEV)
    close(nv.flat_scale(x.clone(), scale_dev=sd), want)
    y = rnd(n, seed=8)
    close(nv.flat_axpy(y.clone(), x, -0.5), y - 0.5 * x)
    a, b = rnd(n, seed=9), rnd(n, seed=10)
    close(nv.add_relu_fwd(a, b, True), torch.relu(a) + b)
    close(nv.add_relu_fwd(a, b, False), torch.relu(a + b))
    d = rnd(n, seed=12)
    da, db = nv.add_relu_bwd(d, a, b, True)
    close(da, d * (a > 0))
    close(db, d)
    da, db = nv.add_relu_bwd(d, a, b, False)
    close(da, d * ((a + b) > 0))
    assert abs(nv.zero_fraction(torch.relu(a)) - float((torch.relu(a) == 0).float().mean())) < 1e-6


@pytest.mark.parametrize("b", [1, 3, 70, 260, 520])
def test_conv1_pool_fused_forward_and_backward(b):
    """Fused first LeNet layer: pooled maximum / winner against conv + bias + ReLU + max-pool, and
    filter / bias gradients against autograd of the unfused chain."""
    x = rnd(b, 28, 28, 1)
    wt = rnd(5, 5, 1, 32, scale=0.3)
    bias = rnd(32, scale=0.2)
    xr = x.clone().requires_grad_(False)
    wr, br = wt.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    act = torch.relu(ref.conv2d_same(xr, wr, 1) + br)
    pooled = torch.nn.functional.max_pool2d(act.permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1)
    p, arg = nv.conv1_pool_fwd(x, wt, bias)
    close(p, pooled)
    # winner positions agree with the unfused kernels wherever the maximum is positive
    _, arg_ref = nv.maxpool2x2_fwd(act.detach().contiguous())
    pos = pooled > 1e-6
    assert torch.equal(arg[pos], arg_ref[pos])
    dp = rnd(b, 14, 14, 32)
    pooled.backward(dp)
    slots = torch.zeros(nv.STAT_SLOTS, 1024, device=DEV)
    dw = slots[0, :800].view(5, 5, 1, 32)
    db = slots[0, 800:832]
    nv.conv1_pool_bwd(x, dp, arg, p, dw, db, slot_stride=1024)
    total = slots.sum(0)
    close(total[:800].view(5, 5, 1, 32), wr.grad, tol=5e-4)
    close(total[800:832], br.grad, tol=5e-4)
    # without slots the sums land in place
    dw2, db2 = torch.zeros(5, 5, 1, 32, device=DEV), torch.zeros(32, device=DEV)
    nv.conv1_pool_bwd(x, dp, arg, p, dw2, db2)
    close(dw2, wr.grad, tol=5e-4)
    close(db2, br.grad, tol=5e-4)


def test_tf_histogram_matches_host_statement():
    from distributed_tensorflow_amd.utils import summary as sm

    x = rnd(3, 17, 9, 16, scale=3.0)
    x = torch.relu(x)            # many exact zeros, like an activation
    x.view(-1)[5] = 1e-13        # below the smallest positive limit
    x.view(-1)[6] = -2.5
    got = nv.tf_histogram(x)
    want = sm.histogram_of(x.cpu().numpy())
    assert got["bucket_limit"] == want["bucket_limit"] and got["bucket"] == want["bucket"]
    assert got["num"] == want["num"] and got["min"] == want["min"] and got["max"] == want["max"]
    assert abs(got["sum"] - want["sum"]) < 1e-6 * abs(want["sum"]) and abs(got["sum_squares"] - want["sum_squares"]) < 1e-6 * want["sum_squares"]


def test_maxpool_same():
    for shape in [(3, 28, 28, 32), (3, 14, 14, 64), (2, 7, 7, 8)]:
        x = rnd(*shape)
        xr = x.clone().requires_grad_(True)
        want = ref.max_pool_2x2_same(xr)
        y, arg = nv.maxpool2x2_fwd(x)
        close(y, want)
        dy = rnd(*want.shape, seed=41)
        (dx_want,) = torch.autograd.grad(want, xr, dy)
        close(nv.maxpool2x2_bwd(dy, arg, y, tuple(x.shape)), dx_want)


def test_dropout_statistics_and_replay():
    n = 1 << 20
    x = torch.ones(n, device=DEV)
    y = nv.dropout(x, 0.5, 66478, 0)
    keep = float((y != 0).float().mean())
    assert abs(keep - 0.5) < 5e-3
    assert torch.all((y == 0) | (y == 2.0))
    # same (seed, offset) regenerates the same mask (backward pass), another offset does not
    assert torch.equal(y, nv.dropout(x, 0.5, 66478, 0))
    assert not torch.equal(y, nv.dropout(x, 0.5, 66478, n // 4))


@pytest.mark.parametrize("b,keep,train", [(5, 1.0, True), (64, 0.5, True), (1030, 0.5, True), (2051, 0.5, True), (77, 0.5, False)])
def test_fc_head_forward_and_backward(b, keep, train):
    """One launch = bias + ReLU + dropout + output layer + loss + the way back; checked against the
    same chain in plain fp32 PyTorch with the mask the stand-alone dropout kernel draws."""
    h_raw, b1 = rnd(b, 512, seed=1), rnd(512, scale=0.3, seed=2)
    w2, b2 = rnd(512, 10, scale=0.1, seed=3), rnd(10, scale=0.1, seed=4)
    labels = torch.randint(0, 10, (b,), generator=torch.Generator().manual_seed(5)).to(DEV, torch.int32)
    seed, offset = 66478, 12
    step = torch.tensor([7], device=DEV, dtype=torch.int64)
    mask = nv.dropout(torch.ones(b, 512, device=DEV), keep, seed, offset + 7) if keep < 1 else torch.ones(b, 512, device=DEV)

    leaves = [t.clone().requires_grad_(True) for t in (h_raw, b1, w2, b2)]
    hd = torch.relu(leaves[0] + leaves[1]) * mask
    logits_ref = hd @ leaves[2] + leaves[3]
    loss_ref = torch.nn.functional.cross_entropy(logits_ref, labels.long())
    g_h, g_b1, g_w2, g_b2 = torch.autograd.grad(loss_ref, leaves)

    logits = torch.empty(b, 10, device=DEV)
    loss = torch.zeros(1, device=DEV)
    correct = torch.zeros(1, device=DEV, dtype=torch.int32)
    kw = dict(keep_prob=keep, seed=seed, offset=offset, offset_dev=step)
    if train:
        # gradient outputs accumulate: start them from something that is not zero
        dh = torch.full((b, 512), 9.0, device=DEV)
        dw2, db2, db1 = rnd(512, 10, seed=6), rnd(10, seed=7), rnd(512, seed=8)
        dw2_0, db2_0, db1_0 = dw2.clone(), db2.clone(), db1.clone()
        kw.update(dh=dh, dw2=dw2, db2=db2, db1=db1)
    nv.fc_head(h_raw, b1, w2, b2, labels, logits, loss, correct, **kw)
    close(logits, logits_ref)
    assert abs(loss.item() - loss_ref.item()) < 1e-4 * max(1.0, abs(loss_ref.item()))
    assert correct.item() == int((logits_ref.argmax(1) == labels.long()).sum())
    if train:
        close(dh, g_h)
        close(dw2 - dw2_0, g_w2, tol=1e-3)   # the difference of two accumulated values: rounding of the larger operand
        close(db2 - db2_0, g_b2, tol=1e-3)
        close(db1 - db1_0, g_b1, tol=1e-3)
    with pytest.raises(Exception):
        nv.fc_head(h_raw, b1, w2, b2, labels, logits, loss, correct, dh=torch.empty(b, 512, device=DEV))


@pytest.mark.parametrize("shape", [(256, 3136, 512), (100, 512, 10), (37, 70, 129), (1024, 512, 3136), (132, 260, 388), (520, 1028, 772)])
def test_gemm_variants(shape):
    m, k, n = shape
    a, b = rnd(m, k), rnd(k, n, seed=3)
    bias = rnd(n, seed=4)
    close(nv.gemm(a, b), a @ b)
    close(nv.gemm(a, b, bias=bias, relu=True), torch.relu(a @ b + bias))
    close(nv.gemm(a.t().contiguous(), b, trans_a=True), a @ b)
    close(nv.gemm(a, b.t().contiguous(), trans_b=True), a @ b)
    close(nv.gemm(a.t().contiguous(), b.t().contiguous(), trans_a=True, trans_b=True), a @ b)
    acc = rnd(m, n, seed=5)
    want = acc + a @ b
    close(nv.gemm(a, b, out=acc, accumulate=True), want)
    out = torch.zeros(n, device=DEV)
    nv.col_sum(b, out)
    close(out, b.sum(0))
    # with a ReLU mask: the masked tensor is written (in place here) and its column sums added
    y = rnd(k, n, seed=6)
    want_masked = b * (y > 0)
    out2 = torch.ones(n, device=DEV)
    g = b.clone()
    nv.col_sum(g, out2, relu_of=y, masked=g)
    assert torch.equal(g, want_masked)
    close(out2 - 1, want_masked.sum(0))


@pytest.mark.parametrize("variant", [2, 3, 4, 5, 6, 7, 8, 9])
def test_gemm_tile_variants_and_splits(variant):
    """Every vector-load tile shape, with and without a k split whose last step is short, in all
    four operand layouts; the work-item-to-XCD mapping must cover every tile exactly once."""
    for (m, k, n), splits in (((256, 3136, 512), 1), ((256, 3136, 512), 7), ((132, 260, 388), 3), ((520, 1028, 772), 2)):
        a, b = rnd(m, k), rnd(k, n, seed=3)
        want = a @ b
        kw = dict(variant=variant, splits=splits)
        close(nv.gemm(a, b, **kw), want)
        close(nv.gemm(a.t().contiguous(), b, trans_a=True, **kw), want)
        close(nv.gemm(a, b.t().contiguous(), trans_b=True, **kw), want)
        close(nv.gemm(a.t().contiguous(), b.t().contiguous(), trans_a=True, trans_b=True, **kw), want)
        bias = rnd(n, seed=4)
        close(nv.gemm(a, b, bias=bias, relu=True, **kw), torch.relu(want + bias))
        acc = rnd(m, n, seed=5)
        close(nv.gemm(a, b, out=acc.clone(), accumulate=True, **kw), acc + want)


def test_crop_whiten():
    src = rnd(10, 36, 36, 3, scale=40.0) + 100
    idx = torch.tensor([3, 1, 7, 7], device=DEV, dtype=torch.int32)
    offs = torch.tensor([[0, 0], [4, 2], [1, 3], [2, 2]], device=DEV, dtype=torch.int32)
    got = nv.crop_whiten(src, idx, offs, (32, 32))
    crops = torch.stack([src[int(i), int(o[0]):int(o[0]) + 32, int(o[1]):int(o[1]) + 32] for i, o in zip(idx, offs)])
    close(got, ref.per_image_whiten(crops), 1e-3)


def test_program_record_run_and_graph_replay():
    n = 4096
    x = torch.ones(n, device=DEV)
    prog = nv.hip().Program()
    prog.begin_record()
    nv.flat_scale(x, 2.0)
    prog.mark_segment()
    nv.flat_scale(x, 3.0)
    prog.end_record()
    assert prog.num_records == 2 and prog.num_segments == 2
    torch.cuda.synchronize()
    assert float(x[0]) == 1.0  # recording does not launch
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        prog.run(s.cuda_stream)
        s.synchronize()
        assert float(x[5]) == 6.0
        prog.run(s.cuda_stream, 1)
        s.synchronize()
        assert float(x[5]) == 18.0
        prog.instantiate(s.cuda_stream)
        prog.replay(s.cuda_stream)
        prog.replay(s.cuda_stream)
        s.synchronize()
    assert float(x[n - 1]) == 18.0 * 36.0


def test_conv_stem_filter_gradient_with_bn_backward_prologue():
    # images -> stem conv -> y -> bn/relu -> (rest): the stem's filter gradient takes the masked
    # gradient at the BatchNorm output and applies the BatchNorm backward while staging it
    b, c = 37, 16
    x = rnd(b, 32, 32, 3)
    w = rnd(3, 3, 3, c, scale=0.3).requires_grad_(True)
    gam = (rnd(c, seed=71).abs() + 0.5).requires_grad_(True)
    bet = rnd(c, scale=0.3, seed=72).requires_grad_(True)
    y = ref.conv2d_same(x, w, 1)
    u = ref.batch_norm(y, gam, bet)
    u.retain_grad()
    r = torch.relu(u)
    dr = rnd(*r.shape, seed=73)
    dw_want, dg, db = torch.autograd.grad((r * dr).sum(), (w, gam, bet), retain_graph=True)
    (du,) = torch.autograd.grad((r * dr).sum(), u)          # masked by relu'
    yd = y.detach()
    mean, var = ref.batch_moments(yd)
    yhat = (yd - mean) * torch.rsqrt(var + ref.BN_EPSILON)
    gs = nv.stats_from_totals(torch.cat([du.reshape(-1, c).double().sum(0), (du * yhat).reshape(-1, c).double().sum(0)]))
    slots = torch.zeros(nv.STAT_SLOTS, 512, device=DEV)
    dgam, dbet = torch.zeros(c, device=DEV), torch.zeros(c, device=DEV)
    nv.conv_stem_wgrad(x, du.contiguous(), slots[0, :432].view(3, 3, 3, c), slot_stride=512,
                       dy_bn=(yd, bn_state_for(yd, gam.detach(), bet.detach()), gs), dy_dgamma=dgam, dy_dbeta=dbet)
    close(slots[:, :432].sum(0).view(3, 3, 3, c), dw_want, 2e-4)
    close(dgam, dg, 1e-4)
    close(dbet, db, 1e-4)


def test_zero_multi_clears_every_buffer_and_nothing_else():
    sizes = [4, 20, 1 << 16, 272476 // 4 * 4, 8]
    guard = 8
    bufs, views = [], []
    for i, n in enumerate(sizes):
        dt = torch.float64 if i == 1 else (torch.int32 if i == 4 else torch.float32)
        full = torch.ones(n + 2 * guard, device="cuda", dtype=dt)
        bufs.append(full)
        views.append(full[guard:guard + n])
    views = [v for v in views if v.data_ptr() % 16 == 0 and (v.numel() * v.element_size()) % 16 == 0]
    assert len(views) >= 4
    nv.zero_multi(views)
    torch.cuda.synchronize()
    for v in views:
        assert int((v != 0).sum()) == 0
    for full in bufs:
        assert bool((full[:guard] == 1).all()) and bool((full[-guard:] == 1).all())
    with pytest.raises(Exception):
        nv.zero_multi([bufs[0][1:5]])
