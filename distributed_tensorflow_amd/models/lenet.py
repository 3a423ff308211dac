"""LeNet-style CNN for MNIST (reference: DistributedLeNet/src/mnist.py:76-159).

conv 5x5x1->32 SAME + bias + ReLU, max-pool 2x2/2 SAME, conv 5x5x32->64 + bias + ReLU, max-pool,
flatten 7*7*64, FC 3136->512 + ReLU, dropout keep 0.5 (training only), FC 512->10.
Weights are truncated-normal(0, 0.1); conv1 bias 0, every other bias 0.1. The reference's
variables are unnamed, so TensorFlow calls them Variable, Variable_1, ... Variable_7 in creation
order; both those names and descriptive ones are carried. 1,663,370 parameters.

This module is the parameter table plus the plain-PyTorch forward (CPU path + numerics oracle);
the MI355X execution is models/lenet_hip.py.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional

import torch
import torch.nn.functional as F

from ..ops import reference as ref
from ..runtime.arena import ParamSpec

IMAGE_SIZE = 28
NUM_CHANNELS = 1
NUM_LABELS = 10
DROPOUT_KEEP = 0.5
SEED = 66478  # reference: mnist.py:30


def lenet_param_specs() -> List[ParamSpec]:
    table = [
        ("conv1_weights", (5, 5, NUM_CHANNELS, 32), "trunc_normal"),
        ("conv1_biases", (32,), "zeros"),
        ("conv2_weights", (5, 5, 32, 64), "trunc_normal"),
        ("conv2_biases", (64,), "const:0.1"),
        ("fc1_weights", (IMAGE_SIZE // 4 * IMAGE_SIZE // 4 * 64, 512), "trunc_normal"),
        ("fc1_biases", (512,), "const:0.1"),
        ("fc2_weights", (512, NUM_LABELS), "trunc_normal"),
        ("fc2_biases", (NUM_LABELS,), "const:0.1"),
    ]
    return [ParamSpec(n, s, i, "Variable" if k == 0 else f"Variable_{k}") for k, (n, s, i) in enumerate(table)]


def lenet_forward(params: Dict[str, torch.Tensor] | Callable[[str], torch.Tensor], images: torch.Tensor,
                  train: bool = False, dropout_gen: Optional[torch.Generator] = None) -> torch.Tensor:
    """Logits [B,10] from NHWC images [B,28,28,1] in [-0.5, 0.5]."""
    p = params if callable(params) else params.__getitem__
    x = torch.relu(ref.conv2d_same(images, p("conv1_weights"), 1) + p("conv1_biases"))
    x = ref.max_pool_2x2_same(x)
    x = torch.relu(ref.conv2d_same(x, p("conv2_weights"), 1) + p("conv2_biases"))
    x = ref.max_pool_2x2_same(x)
    x = x.reshape(x.shape[0], -1)
    h = torch.relu(x @ p("fc1_weights") + p("fc1_biases"))
    if train:
        if dropout_gen is not None:
            mask = (torch.rand(h.shape, generator=dropout_gen) < DROPOUT_KEEP).to(h.dtype).to(h.device)
            h = h * mask / DROPOUT_KEEP
        else:
            h = F.dropout(h, 1.0 - DROPOUT_KEEP, training=True)
    return h @ p("fc2_weights") + p("fc2_biases")


def lenet_loss(params, images: torch.Tensor, labels: torch.Tensor, train: bool = True,
               dropout_gen: Optional[torch.Generator] = None) -> torch.Tensor:
    return ref.softmax_xent_mean(lenet_forward(params, images, train, dropout_gen), labels)
