"""Training on both matrix paths from the same start on the same stream of batches: ResNet-20 (batch 256) and
the LeNet (batch 256) on a learnable synthetic task - ten fixed smooth class patterns under heavy per-pixel noise, 16 384 training
images, a held-out set of 2 048 - with the loss every 50 steps and the held-out loss and accuracy at the end.
The split-operand path is held to the exact path's curve, not to its bits."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from distributed_tensorflow_amd.models.lenet_hip import LeNetHipStep  # noqa: E402
from distributed_tensorflow_amd.models.resnet_hip import ResNetHipStep  # noqa: E402
from distributed_tensorflow_amd.ops import reference as ref  # noqa: E402

STEPS = int(os.environ.get("STEPS", "400"))
BATCH = 256


def dataset(shape, n, seed, noise):
    g = torch.Generator().manual_seed(seed)
    protos = torch.randn(10, *shape, generator=torch.Generator().manual_seed(7))
    # smooth the patterns so that crops of them stay recognisable
    protos = torch.nn.functional.avg_pool2d(protos.permute(0, 3, 1, 2), 5, 1, 2).permute(0, 2, 3, 1) * 3.0
    y = torch.randint(0, 10, (n,), generator=g)
    x = protos[y] + noise * torch.randn(n, *shape, generator=g)
    return x.cuda(), y.cuda()


def run(make_step, shape, lr, noise):
    train_x, train_y = dataset(shape, 16384, 1, noise)
    test_x, test_y = dataset(shape, 2048, 2, noise)
    rows = {}
    for matmul in ("fp32", "bf16x6"):
        torch.manual_seed(0)
        step = make_step(matmul, True)
        order = torch.Generator().manual_seed(3)
        curve = []
        for t in range(1, STEPS + 1):
            idx = torch.randint(0, train_x.shape[0], (BATCH,), generator=order).cuda()
            step.load_batch(train_x[idx], train_y[idx])
            step.set_hyper(lr_t=ref.adam_lr_t(lr, t), step=t)
            step.run()
            if t % 50 == 0:
                curve.append(step.loss_value())
        ev = make_step(matmul, False, step.arena)
        loss = correct = 0.0
        for i in range(0, test_x.shape[0], BATCH):
            ev.load_batch(test_x[i:i + BATCH], test_y[i:i + BATCH])
            ev.run(graph=False)
            loss += ev.loss_value()
            correct += ev.correct_count()
        rows[matmul] = (curve, loss / (test_x.shape[0] // BATCH), correct / test_x.shape[0])
    return rows


def report(name, rows):
    print(name)
    print("   step        " + " ".join("%7d" % (50 * (i + 1)) for i in range(len(rows["fp32"][0]))))
    for k in ("fp32", "bf16x6"):
        print("   %-10s  " % k + " ".join("%7.4f" % v for v in rows[k][0]))
    for k in ("fp32", "bf16x6"):
        print("   %-10s  held-out loss %.4f, accuracy %.4f" % (k, rows[k][1], rows[k][2]))


# per-pixel noise against patterns of unit scale: hard enough to leave errors on the held-out set
report("ResNet-20, batch %d, Adam 1e-3, %d steps, noise 8" % (BATCH, STEPS),
       run(lambda m, tr, arena=None: ResNetHipStep(3, BATCH, arena, matmul=m, training=tr), (32, 32, 3), 1e-3, 8.0))
report("LeNet, batch %d, Adam 1e-3, %d steps, noise 4" % (BATCH, STEPS),
       run(lambda m, tr, arena=None: LeNetHipStep(BATCH, arena, matmul=m, training=tr), (28, 28, 1), 1e-3, 4.0))
