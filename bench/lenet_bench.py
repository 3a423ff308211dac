#!/usr/bin/env python3
"""LeNet (MNIST) training-step throughput on one GPU: hipGraph replay vs direct issue."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.models.lenet_hip import LeNetHipStep  # noqa: E402
from distributed_tensorflow_amd.ops import reference as ref  # noqa: E402


def main():
    for b in (128, 256, 1024, 4096):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            step = LeNetHipStep(b)
            step.load_batch(torch.rand(b, 28, 28, 1, device="cuda") - 0.5, torch.randint(0, 10, (b,), device="cuda"))
            step.instantiate()
            out = {}
            for graph in (True, False):
                for t in range(1, 11):
                    step.set_hyper(lr_t=ref.adam_lr_t(1e-3, t), step=t)
                    step.run(graph=graph)
                s.synchronize()
                t0 = time.perf_counter()
                n = 100
                for t in range(11, 11 + n):
                    step.set_hyper(lr_t=ref.adam_lr_t(1e-3, t), step=t)
                    step.run(graph=graph)
                s.synchronize()
                out[graph] = (time.perf_counter() - t0) / n
        print(f"LeNet batch {b}: graph {out[True] * 1e3:.3f} ms/step ({b / out[True]:.0f} img/s), "
              f"direct {out[False] * 1e3:.3f} ms/step ({b / out[False]:.0f} img/s), {step.kernel_count} launches/step")


if __name__ == "__main__":
    main()
