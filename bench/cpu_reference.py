#!/usr/bin/env python3
"""CPU reference throughput of the ResNet CIFAR-10 training step (the recorded floor).

The reference repository publishes no numbers, so the comparison floor for bench.py is this
measurement: the same model (pre-activation ResNet-(6n+2), batch statistics BN, Adam, fp32,
synthetic CIFAR-shaped data, random-init weights) trained with plain PyTorch ops on the build
sandbox's CPU. Writes bench/cpu_floor.json.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from distributed_tensorflow_amd.models.resnet import resnet_loss, resnet_param_specs  # noqa: E402
from distributed_tensorflow_amd.ops import reference as ref  # noqa: E402
from distributed_tensorflow_amd.runtime.arena import ParamArena  # noqa: E402


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=3, help="residual blocks per stage (3 = ResNet-20)")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--out", default=os.path.join(os.path.dirname(__file__), "cpu_floor.json"))
    args = ap.parse_args()

    arena = ParamArena(resnet_param_specs(args.n), "cpu", ("m", "v"))
    arena.initialize()
    g = torch.Generator().manual_seed(0)
    images = torch.randn(args.batch, 32, 32, 3, generator=g)
    labels = torch.randint(0, 10, (args.batch,), generator=g)
    names = arena.names()

    def one_step(t: int) -> float:
        params = {k: arena.param(k).detach().requires_grad_(True) for k in names}
        loss = resnet_loss(params, images, labels, args.n)
        grads = torch.autograd.grad(loss, [params[k] for k in names])
        lr_t = ref.adam_lr_t(1e-3, t)
        for k, gr in zip(names, grads):
            ref.adam_update(arena.param(k), gr, arena.slot("m", k), arena.slot("v", k), lr_t)
        return float(loss)

    for t in range(args.warmup):
        one_step(t + 1)
    t0 = time.perf_counter()
    for t in range(args.steps):
        loss = one_step(args.warmup + t + 1)
    dt = (time.perf_counter() - t0) / args.steps
    res = {
        "model": f"resnet{6 * args.n + 2}", "batch": args.batch, "dtype": "fp32", "optimizer": "adam",
        "hardware": f"build sandbox CPU ({os.cpu_count()} cores), torch {torch.__version__}",
        "sec_per_step": dt, "images_per_sec": args.batch / dt, "final_loss": loss,
    }
    with open(args.out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
