"""Pre-activation ResNet-(6n+2) for CIFAR-10 (reference: DistributedResNet/src/resnet.py:171-218).

Structure: conv 3x3x3->16 + BN + ReLU; three stages of n residual blocks at 16/32/64 channels and
32/16/8 pixels; the first block of stages 2 and 3 uses stride 2 and the parameter-free option-A
shortcut (2x2 average pool + zero-padded channels); BN + ReLU + global average pool + FC 64->10.
BatchNorm always uses batch statistics, eps 0.001 (reference: resnet.py:10,65-84).

Variables are created in the reference's order so that the TensorFlow auto-generated names
(`Variable`, `Variable_1`, ...) of its active model file line up; every tensor also carries the
scoped name used by the reference's alternative model file
(reference: DistributedResNet/src/back_up.py:41-48,78-82,148-156,183-216).

This module holds the parameter table and the plain-PyTorch forward (CPU path + numerics
oracle). The fused MI355X execution lives in models/resnet_hip.py.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Dict, List, Optional

import torch

from ..ops import reference as ref
from ..runtime.arena import ParamSpec

STAGE_CHANNELS = (16, 32, 64)
NUM_CLASSES = 10


@dataclass(frozen=True)
class ConvLayer:
    """One 3x3 convolution and the BatchNorm (if any) applied to its INPUT."""

    name: str            # scope, e.g. "conv2_1/conv1_in_block"
    cin: int
    cout: int
    stride: int
    in_hw: int
    weight: str          # parameter names
    bn_beta: Optional[str]
    bn_gamma: Optional[str]


@dataclass(frozen=True)
class BlockLayout:
    stage: int
    index: int
    conv1: ConvLayer
    conv2: ConvLayer
    first_block: bool
    increase_dim: bool


def resnet_param_specs(n: int, num_classes: int = NUM_CLASSES) -> List[ParamSpec]:
    """Parameters in TensorFlow creation order with auto and scoped names."""
    specs: List[ParamSpec] = []

    def add(name, shape, init):
        idx = len(specs)
        tf_name = "Variable" if idx == 0 else f"Variable_{idx}"
        specs.append(ParamSpec(name, tuple(shape), init, tf_name))

    # conv_bn_relu_layer: filter, then beta, gamma (reference: resnet.py:97-100)
    add("conv0/conv", (3, 3, 3, 16), "trunc_normal")
    add("conv0/beta", (16,), "zeros")
    add("conv0/gamma", (16,), "ones")
    cin = 16
    for stage, cout in enumerate(STAGE_CHANNELS, start=1):
        for i in range(n):
            scope = f"conv{stage}_{i}"
            first = stage == 1 and i == 0
            if not first:  # bn_relu_conv_layer: beta, gamma, filter (reference: resnet.py:117-121)
                add(f"{scope}/conv1_in_block/beta", (cin,), "zeros")
                add(f"{scope}/conv1_in_block/gamma", (cin,), "ones")
            add(f"{scope}/conv1_in_block/conv", (3, 3, cin, cout), "trunc_normal")
            add(f"{scope}/conv2_in_block/beta", (cout,), "zeros")
            add(f"{scope}/conv2_in_block/gamma", (cout,), "ones")
            add(f"{scope}/conv2_in_block/conv", (3, 3, cout, cout), "trunc_normal")
            cin = cout
    add("fc/beta", (cin,), "zeros")
    add("fc/gamma", (cin,), "ones")
    add("fc/fc_weights", (cin, num_classes), "trunc_normal")
    # the reference ignores its zeros initializer, fc_bias is random too (resnet.py:41-46,59)
    add("fc/fc_bias", (num_classes,), "trunc_normal")
    return specs


def resnet_blocks(n: int) -> List[BlockLayout]:
    blocks: List[BlockLayout] = []
    cin, hw = 16, 32
    for stage, cout in enumerate(STAGE_CHANNELS, start=1):
        for i in range(n):
            scope = f"conv{stage}_{i}"
            first = stage == 1 and i == 0
            inc = cin * 2 == cout
            stride = 2 if inc else 1
            c1 = ConvLayer(
                f"{scope}/conv1_in_block", cin, cout, stride, hw, f"{scope}/conv1_in_block/conv",
                None if first else f"{scope}/conv1_in_block/beta",
                None if first else f"{scope}/conv1_in_block/gamma",
            )
            hw_out = hw // stride
            c2 = ConvLayer(
                f"{scope}/conv2_in_block", cout, cout, 1, hw_out, f"{scope}/conv2_in_block/conv",
                f"{scope}/conv2_in_block/beta", f"{scope}/conv2_in_block/gamma",
            )
            blocks.append(BlockLayout(stage, i, c1, c2, first, inc))
            cin, hw = cout, hw_out
    return blocks


def num_layers(n: int) -> int:
    return 6 * n + 2


def resnet_forward(params: Dict[str, torch.Tensor] | Callable[[str], torch.Tensor], images: torch.Tensor, n: int) -> torch.Tensor:
    """Logits [B,10] from NHWC images [B,32,32,3]; `params` maps scoped names to tensors."""
    p = params if callable(params) else params.__getitem__
    x = ref.conv2d_same(images, p("conv0/conv"), 1)
    x = torch.relu(ref.batch_norm(x, p("conv0/gamma"), p("conv0/beta")))
    for blk in resnet_blocks(n):
        inp = x
        if blk.first_block:
            c1 = ref.conv2d_same(inp, p(blk.conv1.weight), 1)
        else:
            a = torch.relu(ref.batch_norm(inp, p(blk.conv1.bn_gamma), p(blk.conv1.bn_beta)))
            c1 = ref.conv2d_same(a, p(blk.conv1.weight), blk.conv1.stride)
        a2 = torch.relu(ref.batch_norm(c1, p(blk.conv2.bn_gamma), p(blk.conv2.bn_beta)))
        c2 = ref.conv2d_same(a2, p(blk.conv2.weight), 1)
        x = c2 + (ref.shortcut_a(inp) if blk.increase_dim else inp)
    x = torch.relu(ref.batch_norm(x, p("fc/gamma"), p("fc/beta")))
    feat = x.mean(dim=(1, 2))
    return feat @ p("fc/fc_weights") + p("fc/fc_bias")


def resnet_loss(params, images: torch.Tensor, labels: torch.Tensor, n: int) -> torch.Tensor:
    """Mean sparse softmax cross-entropy (reference: DistributedResNet/src/distributed_train.py:119-129)."""
    return ref.softmax_xent_mean(resnet_forward(params, images, n), labels)
