"""Per-tensor gradient error of the fused ResNet-20 step against fp64 autograd on the GPU, next to the
error of plain fp32 PyTorch (the rounding-noise yardstick), for several batch sizes."""
import sys

import torch

sys.path.insert(0, "tests")
sys.path.insert(0, ".")
import test_headline_sizes_gpu as T  # noqa: E402
from distributed_tensorflow_amd.models.resnet import resnet_loss  # noqa: E402
from distributed_tensorflow_amd.models.resnet_hip import ResNetHipStep  # noqa: E402
from distributed_tensorflow_amd.ops import reference as ref  # noqa: E402

ref.conv2d_same = T.conv2d_same_unfold
n = 3
for batch in [int(x) for x in sys.argv[1:]] or [256, 512, 1024]:
    step = ResNetHipStep(n, batch)
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
    g1 = step.arena.grads.clone()
    step.run(graph=False)
    torch.cuda.synchronize()
    g2 = step.arena.grads.clone()
    state = step.arena.state_dict()
    fn = lambda p, x, y: resnet_loss(p, x, y, n)   # noqa: E731
    loss_ref, gr = T._reference_grads(fn, state, images, labels, torch.float64)
    _, g32 = T._reference_grads(fn, state, images, labels, torch.float32)
    rows = []
    for name, gref in gr.items():
        off = step.arena.offsets[name]
        a = g1[off:off + gref.numel()].view(gref.shape)
        b = g2[off:off + gref.numel()].view(gref.shape)
        l2 = lambda u, v: float((u.double() - v.double()).norm() / (v.double().norm() + 1e-30))   # noqa: E731
        rows.append((T.rel(a, gref), T.rel(g32[name], gref), T.rel(a, b), float(gref.abs().max()), name, l2(a, gref), l2(g32[name], gref)))
    rows.sort(reverse=True)
    print(f"batch {batch}: loss {step.loss_value():.6f} ref {loss_ref:.6f}")
    print("  err(ours)  err(fp32 torch)  run-to-run   max|g|     tensor")
    for r in rows[:8]:
        print("  %.2e   %.2e         %.2e     %.2e  %s   L2: ours %.2e torch32 %.2e" % r)
    import math
    print("  worst L2 ours %.2e torch32 %.2e | rms of max-errors ours %.2e torch32 %.2e | worst ratio max %.1f L2 %.1f" % (
        max(r[5] for r in rows), max(r[6] for r in rows), math.sqrt(sum(r[0] ** 2 for r in rows) / len(rows)),
        math.sqrt(sum(r[1] ** 2 for r in rows) / len(rows)), max(r[0] / max(r[1], 5e-4) for r in rows), max(r[5] / max(r[6], 2.5e-4) for r in rows)))
