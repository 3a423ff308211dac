"""Synthetic inputs of the datasets' shapes, resident on the training device (benchmarks, smoke
tests and every environment without dataset files)."""
from __future__ import annotations

from typing import Tuple

import torch


def synthetic_batch(kind: str, batch: int, device: torch.device | str = "cpu", seed: int = 0) -> Tuple[torch.Tensor, torch.Tensor]:
    g = torch.Generator(device="cpu").manual_seed(seed)
    if kind == "mnist":
        x = torch.rand(batch, 28, 28, 1, generator=g) - 0.5
    elif kind == "cifar10":
        x = torch.randn(batch, 32, 32, 3, generator=g)
    else:
        raise ValueError(f"unknown dataset kind {kind}")
    y = torch.randint(0, 10, (batch,), generator=g)
    return x.to(device), y.to(device)
