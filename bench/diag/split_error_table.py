"""Error of the split-operand ("bf16x6") path next to the exact-fp32 path, both against fp64 on the
same inputs: (1) every stride-1 layer shape of the network, forward kernel with BatchNorm prologue,
identity residual and statistics, batch 1024, and the fused backward kernel (data gradient and filter
gradient) at batch 256; (2) a whole ResNet-20 training step at batch 256 and
1024: loss and per-tensor gradient error. The rule the split path is held to: max error <= 2 x the
exact path's. (3) The LeNet's second convolution: forward, data gradient and filter gradient kernels."""
import sys

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import test_headline_sizes_gpu as T  # noqa: E402
from distributed_tensorflow_amd.models.resnet import resnet_loss  # noqa: E402
from distributed_tensorflow_amd.models.resnet_hip import ResNetHipStep  # noqa: E402
from distributed_tensorflow_amd.ops import native as nv  # noqa: E402
from distributed_tensorflow_amd.ops import reference as ref  # noqa: E402

print("1. forward kernel, batch 1024: max |error| / max |reference| against fp64 (five seeds each)")
print("   shape        exact fp32    bf16x6      ratio")
worst_ratio = 0.0
for hw, c in [(32, 16), (16, 32), (8, 64)]:
    e32, esp = 0.0, 0.0
    for seed in range(5):
        x = T.rnd(1024, hw, hw, c, scale=2.0, seed=3 + 10 * seed) + 0.3
        gamma, beta = T.rnd(c, seed=4 + 10 * seed).abs() + 0.5, T.rnd(c, scale=0.2, seed=5 + 10 * seed)
        wt = T.rnd(3, 3, c, c, scale=0.1, seed=6 + 10 * seed)
        res = T.rnd(1024, hw, hw, c, seed=7 + 10 * seed)
        a = torch.relu(ref.batch_norm(x.double(), gamma.double(), beta.double()))
        want = T.conv2d_same_unfold(a, wt.double(), 1) + res.double()
        bn = T.bn_state_for(x, gamma, beta)
        e32 = max(e32, T.rel(nv.conv_fprop(x, nv.pack_weights(wt), ks=3, cout=c, in_bn=bn, res_mode=nv.RES_IDENT, res=res), want))
        esp = max(esp, T.rel(nv.conv_fprop(x, nv.pack_split_weights(wt), ks=3, cout=c, in_bn=bn, res_mode=nv.RES_IDENT, res=res, split=True), want))
    worst_ratio = max(worst_ratio, esp / e32)
    print(f"   {hw}x{hw}x{c:<3d}     {e32:.3e}     {esp:.3e}   {esp / e32:.2f}")

print("1b. fused backward kernel, batch 256: data gradient g and filter gradient dW, same measure (three seeds each)")
print("   shape        g exact     g bf16x6   ratio    dW exact    dW bf16x6  ratio")
for hw, c in [(32, 16), (16, 32), (8, 64)]:
    if not nv.hip().conv_bwd_split_supported(256, hw, hw, c):
        continue
    eg32 = egsp = ew32 = ewsp = 0.0
    for seed in range(3):
        b = 256
        x = T.rnd(b, hw, hw, c, scale=2.0, seed=3 + 10 * seed) + 0.3
        gamma, beta = T.rnd(c, seed=4 + 10 * seed).abs() + 0.5, T.rnd(c, scale=0.2, seed=5 + 10 * seed)
        wt = T.rnd(3, 3, c, c, scale=0.1, seed=6 + 10 * seed)
        dy = T.rnd(b, hw, hw, c, seed=7 + 10 * seed)
        u = ref.batch_norm(x.double(), gamma.double(), beta.double())
        a = torch.relu(u).requires_grad_(True)
        wd = wt.double().requires_grad_(True)
        da, dw = torch.autograd.grad(T.conv2d_same_unfold(a, wd, 1), (a, wd), dy.double())
        want_g = torch.where(u > 0, da, torch.zeros_like(da))
        bn = T.bn_state_for(x, gamma, beta)
        n = wt.numel()
        for split in (False, True):
            slots = torch.zeros(nv.STAT_SLOTS, n, device="cuda")
            wpk = nv.pack_split_weights(wt, flip=True) if split else nv.pack_weights(wt, flip=True)
            got = nv.conv_bwd_fused(dy, x, bn, wpk, slots[0].view(3, 3, c, c), nv.new_stats(c, "cuda"), slot_stride=n, split=split)
            eg, ew = T.rel(got, want_g), T.rel(slots.double().sum(0).view(dw.shape), dw)
            if split:
                egsp, ewsp = max(egsp, eg), max(ewsp, ew)
            else:
                eg32, ew32 = max(eg32, eg), max(ew32, ew)
    worst_ratio = max(worst_ratio, egsp / eg32, ewsp / ew32)
    print(f"   {hw}x{hw}x{c:<3d}     {eg32:.3e}   {egsp:.3e}  {egsp / eg32:.2f}     {ew32:.3e}   {ewsp:.3e}  {ewsp / ew32:.2f}")

ref.conv2d_same = T.conv2d_same_unfold
print("2. ResNet-20 training step against fp64 autograd: loss error, and over the 65 parameter tensors the worst and the")
print("   root-mean-square of (max |gradient error| / max |gradient|); 'torch fp32' = plain PyTorch in fp32, the yardstick")
print("   batch   path        |loss err|   worst tensor   rms over tensors")
for batch in (256, 1024):
    rows = {}
    for matmul in ("fp32", "bf16x6"):
        step = ResNetHipStep(3, batch, matmul=matmul)
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
        step.run(graph=False)
        torch.cuda.synchronize()
        state = step.arena.state_dict()
        fn = lambda p, x, y: resnet_loss(p, x, y, 3)   # noqa: E731
        loss_ref, gr = T._reference_grads(fn, state, images, labels, torch.float64)
        errs = [T.rel(step.arena.grad(k), v) for k, v in gr.items()]
        rows[matmul] = (abs(step.loss_value() - loss_ref), max(errs), (sum(e * e for e in errs) / len(errs)) ** 0.5)
        if matmul == "fp32":
            l32, g32 = T._reference_grads(fn, state, images, labels, torch.float32)
            e = [T.rel(g32[k], v) for k, v in gr.items()]
            rows["torch fp32"] = (abs(l32 - loss_ref), max(e), (sum(x * x for x in e) / len(e)) ** 0.5)
    for k in ("fp32", "bf16x6", "torch fp32"):
        print(f"   {batch:<6d}  {k:<10s}  {rows[k][0]:.2e}     {rows[k][1]:.2e}       {rows[k][2]:.2e}")
    worst_ratio = max(worst_ratio, rows["bf16x6"][1] / rows["fp32"][1], rows["bf16x6"][2] / rows["fp32"][2])
print("3. LeNet second convolution (5x5, 14x14, 32 -> 64), batch 1024: forward with bias + ReLU, data gradient, filter gradient (three seeds)")
print("   kernel            exact fp32    bf16x6      ratio")
e = {k: [0.0, 0.0] for k in ("forward", "data gradient", "filter gradient")}
for seed in range(3):
    b = 1024
    x = T.rnd(b, 14, 14, 32, seed=3 + 10 * seed)
    wt = T.rnd(5, 5, 32, 64, scale=0.05, seed=4 + 10 * seed)
    bias = T.rnd(64, scale=0.3, seed=5 + 10 * seed)
    dy = T.rnd(b, 14, 14, 64, seed=6 + 10 * seed)
    xd, wd = x.double().requires_grad_(True), wt.double().requires_grad_(True)
    pre = T.conv2d_same_unfold(xd, wd, 1)
    want_y = torch.relu(pre + bias.double())
    want_dx, want_dw = torch.autograd.grad(pre, (xd, wd), dy.double())
    for i, split in enumerate((False, True)):
        wf = nv.conv5x5_split_pack(wt) if split else nv.conv5x5_pack(wt)
        wb = nv.conv5x5_split_pack(wt, flip=True) if split else wt
        e["forward"][i] = max(e["forward"][i], T.rel(nv.conv5x5_tile(x, wf, bias=bias, split=split), want_y))
        e["data gradient"][i] = max(e["data gradient"][i], T.rel(nv.conv5x5_tile(dy, wb, flip=True, split=split), want_dx))
        ws = torch.empty(nv.conv5x5_wgrad_parts(b), 25 * 32 * 64, device="cuda")
        dw = torch.zeros(5, 5, 32, 64, device="cuda")
        nv.conv5x5_wgrad(x, dy, dw, partials=ws, split=split)
        e["filter gradient"][i] = max(e["filter gradient"][i], T.rel(dw, want_dw))
for k, (e32, esp) in e.items():
    worst_ratio = max(worst_ratio, esp / e32)
    print(f"   {k:<16s}  {e32:.3e}     {esp:.3e}   {esp / e32:.2f}")
print(f"worst bf16x6 / exact ratio over everything above: {worst_ratio:.2f}  (rule: <= 2)")
