#!/usr/bin/env python3
"""A few LeNet training steps at one batch size, for rocprofv3 --kernel-trace."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.models.lenet_hip import LeNetHipStep  # noqa: E402
from distributed_tensorflow_amd.ops import reference as ref  # noqa: E402

b = int(sys.argv[1]) if len(sys.argv) > 1 else 256
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    step = LeNetHipStep(b)
    step.load_batch(torch.rand(b, 28, 28, 1, device="cuda") - 0.5, torch.randint(0, 10, (b,), device="cuda"))
    for t in range(1, 9):
        step.set_hyper(lr_t=ref.adam_lr_t(1e-3, t), step=t)
        step.run(graph=False)
    s.synchronize()
print("loss", step.loss_value())
