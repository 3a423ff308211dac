"""Plain PyTorch fp32 reference implementations of every op, with TensorFlow semantics.

Tensors are NHWC, filters HWIO. These functions are (a) the numerics oracle for the HIP kernels
in tests, and (b) the CPU execution path of the models, so the whole framework runs without a
GPU. They are autograd-differentiable.

TensorFlow `SAME` padding is asymmetric when the total padding is odd (the extra row/column goes
at the bottom/right), which matters for the stride-2 convolutions of the ResNet
(reference: DistributedResNet/src/resnet.py:99,121) and the 2x2/2 max pools of LeNet on odd
sizes (reference: DistributedLeNet/src/mnist.py:115-127).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

BN_EPSILON = 0.001  # reference: DistributedResNet/src/resnet.py:10


def same_pad(size: int, k: int, stride: int) -> tuple[int, int, int]:
    """Returns (out_size, pad_before, pad_after) of TensorFlow SAME padding."""
    out = (size + stride - 1) // stride
    total = max((out - 1) * stride + k - size, 0)
    before = total // 2
    return out, before, total - before


def conv2d_same(x: torch.Tensor, w: torch.Tensor, stride: int = 1) -> torch.Tensor:
    """x [B,H,W,Cin] NHWC, w [kh,kw,Cin,Cout] HWIO -> [B,Ho,Wo,Cout]."""
    kh, kw = w.shape[0], w.shape[1]
    _, pt, pb = same_pad(x.shape[1], kh, stride)
    _, pl, pr = same_pad(x.shape[2], kw, stride)
    xn = x.permute(0, 3, 1, 2)
    xn = F.pad(xn, (pl, pr, pt, pb))
    y = F.conv2d(xn, w.permute(3, 2, 0, 1), stride=stride)
    return y.permute(0, 2, 3, 1).contiguous()


def batch_moments(x: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    """Per-channel mean and biased variance over all leading axes (tf.nn.moments axes=[0,1,2])."""
    dims = tuple(range(x.dim() - 1))
    mean = x.mean(dim=dims)
    var = ((x - mean) ** 2).mean(dim=dims)
    return mean, var


def batch_norm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = BN_EPSILON) -> torch.Tensor:
    """BatchNorm with batch statistics (always; the reference keeps no moving averages)."""
    mean, var = batch_moments(x)
    return (x - mean) * torch.rsqrt(var + eps) * gamma + beta


def shortcut_a(x: torch.Tensor) -> torch.Tensor:
    """avg_pool 2x2/2 VALID then zero-pad channels by C/2 on each side (C -> 2C)."""
    c = x.shape[-1]
    xn = x.permute(0, 3, 1, 2)
    p = F.avg_pool2d(xn, 2, 2).permute(0, 2, 3, 1)
    return F.pad(p, (c // 2, c // 2)).contiguous()


def max_pool_2x2_same(x: torch.Tensor) -> torch.Tensor:
    h, w = x.shape[1], x.shape[2]
    xn = x.permute(0, 3, 1, 2)
    xn = F.pad(xn, (0, w % 2, 0, h % 2), value=float("-inf"))
    return F.max_pool2d(xn, 2, 2).permute(0, 2, 3, 1).contiguous()


def softmax_xent_mean(logits: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    return F.cross_entropy(logits, labels.long(), reduction="mean")


def top1_hits(logits: torch.Tensor, labels: torch.Tensor) -> int:
    return int((logits.argmax(dim=1) == labels.long()).sum().item())


def per_image_whiten(x: torch.Tensor) -> torch.Tensor:
    """(x - mean) / max(std, 1/sqrt(numel)) per image (reference: cifar10_input.py:111-122)."""
    b = x.shape[0]
    flat = x.reshape(b, -1)
    mean = flat.mean(dim=1, keepdim=True)
    std = flat.std(dim=1, unbiased=False, keepdim=True)
    std = torch.clamp(std, min=1.0 / math.sqrt(flat.shape[1]))
    return ((flat - mean) / std).reshape_as(x)


def adam_update(p, g, m, v, lr_t: float, beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8):
    """tf.train.AdamOptimizer update with the bias-corrected step size already folded into lr_t."""
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    p.sub_(lr_t * m / (v.sqrt() + eps))


def momentum_update(p, g, m, lr: float, momentum: float, weight_decay: float = 0.0):
    g = g + weight_decay * p
    m.mul_(momentum).add_(g)
    p.sub_(lr * m)


def adam_lr_t(lr: float, step: int, beta1: float = 0.9, beta2: float = 0.999) -> float:
    """Step size of update number `step` (1-based)."""
    return lr * math.sqrt(1.0 - beta2 ** step) / (1.0 - beta1 ** step)
