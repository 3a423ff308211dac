"""Numerics at the sizes the benchmarks run at.

The small-shape tests give a persistent convolution kernel fewer tiles than workgroups, so its
steady state - write the prefetched tile, prefetch the next, run the MFMA phase, loop - never takes
a second iteration under an oracle. Here every workgroup of the band kernels takes several tiles and
the last round is ragged (B = 300, 1030), and whole training steps run at batch 512 / 1024 (ResNet-20)
and 256 (LeNet). The reference is plain PyTorch in fp64 ON THE GPU (convolutions as unfold + matmul,
which needs no fp64 convolution library); the same code in fp32 calibrates how much rounding alone
moves each gradient, as in test_resnet_gpu.py.
"""
import pytest
import torch
import torch.nn.functional as F

from distributed_tensorflow_amd.ops import native as nv
from distributed_tensorflow_amd.ops import reference as ref

pytestmark = pytest.mark.gpu

DEV = "cuda"


def conv2d_same_unfold(x, w, stride=1):
    """TensorFlow-SAME convolution as im2col + matmul: any dtype, NHWC / HWIO (twin of ref.conv2d_same)."""
    kh, kw, cin, cout = w.shape
    b = x.shape[0]
    ho, pt, pb = ref.same_pad(x.shape[1], kh, stride)
    wo, pl, pr = ref.same_pad(x.shape[2], kw, stride)
    xn = F.pad(x.permute(0, 3, 1, 2), (pl, pr, pt, pb))
    cols = F.unfold(xn, (kh, kw), stride=stride)                       # [B, Cin*kh*kw, Ho*Wo], channel-major rows
    wm = w.permute(2, 0, 1, 3).reshape(cin * kh * kw, cout)            # rows in the same (cin, kh, kw) order
    return (cols.transpose(1, 2) @ wm).reshape(b, ho, wo, cout)


@pytest.fixture
def unfold_convolutions(monkeypatch):
    monkeypatch.setattr(ref, "conv2d_same", conv2d_same_unfold)


def rnd(*shape, scale=1.0, seed=1):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


def rel(a, b):
    a, b = a.double(), b.double()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


def bn_state_for(x, gamma, beta):
    c = x.shape[-1]
    flat = x.reshape(-1, c).double()
    sums = torch.cat([flat.sum(0), (flat * flat).sum(0)])
    return nv.BnState(nv.stats_from_totals(sums), gamma, beta, flat.shape[0])


BAND_SHAPES = [(32, 16, 16, 1), (16, 32, 32, 1), (8, 64, 64, 1), (32, 16, 32, 2), (16, 32, 64, 2)]   # (H = W, Cin, Cout, stride)


@pytest.mark.parametrize("b", [300, 1030])
@pytest.mark.parametrize("split", [False, True], ids=["fp32", "bf16x6"])
@pytest.mark.parametrize("res_mode", ["none", "ident", "bnrelu", "shortcut_a"])
@pytest.mark.parametrize("shape", BAND_SHAPES)
def test_band_forward_steady_state_against_fp64(shape, res_mode, split, b):
    """split: the opt-in split-operand kernel (bf16 matrix pipe, three-way split operands) under the
    same oracle and the same tolerance as the exact-fp32 kernel."""
    hw, cin, cout, s = shape
    if s == 2 and res_mode != "none":
        pytest.skip("the stride-2 layers of the network are conv1 of a block: no residual epilogue")
    if split and (s == 2 or (res_mode == "shortcut_a" and cin == 16)):
        pytest.skip("the split-operand kernel covers the stride-1 layers (no option-A shortcut at 16 channels)")
    x = rnd(b, hw, hw, cin, scale=2.0, seed=3) + 0.3
    gamma, beta = rnd(cin, seed=4).abs() + 0.5, rnd(cin, scale=0.2, seed=5)
    wt = rnd(3, 3, cin, cout, scale=0.1, seed=6)
    ho = hw // s
    kw = {}
    xd = x.double()
    a = torch.relu(ref.batch_norm(xd, gamma.double(), beta.double()))
    want = conv2d_same_unfold(a, wt.double(), s)
    if res_mode == "ident":
        res = rnd(b, ho, ho, cout, seed=7)
        want = want + res.double()
        kw = dict(res_mode=nv.RES_IDENT, res=res)
    elif res_mode == "bnrelu":
        res = rnd(b, ho, ho, cout, seed=7)
        g2, b2 = rnd(cout, seed=8).abs() + 0.5, rnd(cout, scale=0.3, seed=9)
        want = want + torch.relu(ref.batch_norm(res.double(), g2.double(), b2.double()))
        kw = dict(res_mode=nv.RES_BNRELU, res=res, res_bn=bn_state_for(res, g2, b2))
    elif res_mode == "shortcut_a":
        src = rnd(b, 2 * ho, 2 * ho, cout // 2, seed=10)
        want = want + ref.shortcut_a(src.double())
        kw = dict(res_mode=nv.RES_SHORTCUT_A, res=src)
    sums = nv.new_stats(cout, DEV)
    wpk = nv.pack_split_weights(wt) if split else nv.pack_weights(wt)
    got = nv.conv_fprop(x, wpk, ks=3, cout=cout, stride=s, in_bn=bn_state_for(x, gamma, beta), out_sums=sums, split=split, **kw)
    torch.cuda.synchronize()
    assert rel(got, want) < 2e-5, rel(got, want)
    flat = want.reshape(-1, cout)
    tot = nv.stats_total(sums)
    assert rel(tot[:cout], flat.sum(0)) < 1e-5 and rel(tot[cout:], (flat * flat).sum(0)) < 1e-5


def _check_gradients(arena, grads_ref, grads32):
    """At these batch sizes a gradient tensor is a sum over up to a million pixels, and which way
    an activation within rounding of zero falls decides whether its ReLU passes a gradient: every
    such flip moves a filter gradient by one pixel's product, about 1 / sqrt(pixels) of the tensor's
    scale. Plain fp32 PyTorch is 1e-3 .. 1e-2 away from fp64 for that reason (`noise`), a different
    draw for every tensor and every evaluation order; an indexing mistake in a kernel's steady state
    moves a tensor by O(1). So: every tensor within 3e-2 (max) and 1.5e-2 (L2) of fp64, nine in ten
    within four times the fp32 yardstick of the same tensor, and the root mean square over all
    tensors within four times the yardstick's."""
    errs, noises, calibrated = [], [], 0
    for name, gref in grads_ref.items():
        got = arena.grad(name)
        err, noise = rel(got, gref), rel(grads32[name], gref)
        l2 = float((got.double() - gref).norm() / (gref.norm() + 1e-30))
        assert err < 3e-2 and l2 < 1.5e-2, f"{name}: gradient error max {err:.2e} L2 {l2:.2e} (fp32 noise {noise:.2e})"
        calibrated += int(err < max(2e-3, 4.0 * noise))
        errs.append(err)
        noises.append(noise)
    assert calibrated >= 0.9 * len(errs), (calibrated, len(errs))
    rms = lambda v: (sum(x * x for x in v) / len(v)) ** 0.5   # noqa: E731
    assert rms(errs) < max(2e-3, 4.0 * rms(noises)), (rms(errs), rms(noises))


def _reference_grads(loss_fn, state, images, labels, dtype):
    params = {k: v.to(DEV, dtype).requires_grad_(True) for k, v in state.items()}
    loss = loss_fn(params, images.to(DEV, dtype), labels.to(DEV))
    grads = torch.autograd.grad(loss, list(params.values()))
    return loss.item(), {k: g.detach() for k, g in zip(params, grads)}


@pytest.mark.parametrize("matmul", ["fp32", "bf16x6"])
@pytest.mark.parametrize("batch", [512, 1024])
def test_resnet20_step_at_benchmark_batch_matches_autograd(batch, matmul, unfold_convolutions):
    from distributed_tensorflow_amd.models.resnet import resnet_loss
    from distributed_tensorflow_amd.models.resnet_hip import ResNetHipStep

    n = 3
    step = ResNetHipStep(n, batch, matmul=matmul)
    assert len(step._split_fwd) == (16 if matmul == "bf16x6" else 0)
    g = torch.Generator().manual_seed(1234)
    for s in step.arena.specs:
        if s.name.endswith("gamma"):
            step.arena.param(s.name).copy_(1.0 + 0.2 * torch.randn(s.shape, generator=g))
        if s.name.endswith("beta"):
            step.arena.param(s.name).copy_(0.1 * torch.randn(s.shape, generator=g))
    images = torch.randn(batch, 32, 32, 3, generator=g)
    labels = torch.randint(0, 10, (batch,), generator=g)
    step.load_batch(images.cuda(), labels.cuda())
    step.set_hyper(lr_t=0.0)
    step.instantiate()
    step.run(graph=True)          # the replayed hipGraph, as bench.py runs it
    torch.cuda.synchronize()
    state = step.arena.state_dict()
    loss_fn = lambda p, x, y: resnet_loss(p, x, y, n)   # noqa: E731
    loss_ref, grads_ref = _reference_grads(loss_fn, state, images, labels, torch.float64)
    _, grads32 = _reference_grads(loss_fn, state, images, labels, torch.float32)
    assert abs(step.loss_value() - loss_ref) < 2e-4 * max(1.0, abs(loss_ref))
    _check_gradients(step.arena, grads_ref, grads32)


@pytest.mark.parametrize("matmul", ["fp32", "bf16x6"])
def test_lenet_step_at_batch_256_matches_autograd(unfold_convolutions, matmul):
    from distributed_tensorflow_amd.models.lenet import lenet_loss
    from distributed_tensorflow_amd.models.lenet_hip import LeNetHipStep

    b = 256
    step = LeNetHipStep(b, dropout=False, matmul=matmul)
    g = torch.Generator().manual_seed(3)
    images = torch.rand(b, 28, 28, 1, generator=g) - 0.5
    labels = torch.randint(0, 10, (b,), generator=g)
    step.load_batch(images.cuda(), labels.cuda())
    step.set_hyper(lr_t=0.0)
    step.instantiate()
    step.run(graph=True)
    torch.cuda.synchronize()
    state = step.arena.state_dict()
    loss_fn = lambda p, x, y: lenet_loss(p, x, y, train=False)   # noqa: E731
    loss_ref, grads_ref = _reference_grads(loss_fn, state, images, labels, torch.float64)
    _, grads32 = _reference_grads(loss_fn, state, images, labels, torch.float32)
    assert abs(step.loss_value() - loss_ref) < 1e-4 * max(1.0, abs(loss_ref))
    _check_gradients(step.arena, grads_ref, grads32)
