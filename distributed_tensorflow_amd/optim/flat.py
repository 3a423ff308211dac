"""Optimizers over a flat parameter arena, usable without the training driver.

`FlatAdam` is tf.train.AdamOptimizer as the reference configures it (beta1 0.9, beta2 0.999,
epsilon 1e-8, bias correction folded into the step size; reference:
DistributedLeNet/src/distributed_train.py:164), `FlatMomentumSGD` is momentum SGD with L2 weight
decay (the `weight_decay` hyper-parameter of DistributedResNet/src/hyper_parameters.py:29).

One `step()` is ONE kernel launch over the whole arena on a GPU (csrc/kernels/optim.hip); the
averaging of an aggregated gradient (`grad_scale` = 1 / contributors) happens inside that launch,
so a summed gradient never needs a separate scaling pass. On a CPU arena the same arithmetic runs
in PyTorch (ops/reference.py), which is also what the kernels are tested against.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from ..ops import reference as ref
from ..runtime.arena import ParamArena


class _FlatOptimizer:
    slots: Tuple[str, ...] = ()

    def __init__(self, arena: ParamArena):
        for s in self.slots:
            if s not in arena.slots:
                raise ValueError(f"arena was built without optimizer slot '{s}' (pass slots={self.slots})")
        self.arena = arena
        self.steps = 0
        self._gpu = arena.params.is_cuda
        if self._gpu:
            from ..ops import native as nv

            self._nv = nv
            self._hyper = torch.zeros(nv.HYPER_SLOTS, device=arena.params.device, dtype=torch.float32)

    def _stage(self, lr_t, beta1, beta2, eps, grad_scale, momentum, weight_decay) -> None:
        # by value with the launch: steps queued without a host synchronisation keep their own values
        self._nv.set_scalars(self._hyper, (lr_t, beta1, beta2, eps, grad_scale, momentum, weight_decay))

    def _range(self, lo: Optional[int], hi: Optional[int]) -> Tuple[int, int]:
        lo = 0 if lo is None else lo
        hi = self.arena.total if hi is None else hi
        if not (0 <= lo <= hi <= self.arena.total):
            raise ValueError(f"range [{lo}, {hi}) outside the arena of {self.arena.total} elements")
        return lo, hi

    def zero_grad(self) -> None:
        self.arena.grads.zero_()


class FlatAdam(_FlatOptimizer):
    slots = ("m", "v")

    def __init__(self, arena: ParamArena, beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8):
        super().__init__(arena)
        self.beta1, self.beta2, self.eps = beta1, beta2, eps

    def step(self, lr: float, grad_scale: float = 1.0, lo: Optional[int] = None, hi: Optional[int] = None) -> None:
        """Updates params[lo:hi] (default: everything) from grads[lo:hi] * grad_scale."""
        lo, hi = self._range(lo, hi)
        a = self.arena
        t = self.steps + 1
        lr_t = lr * (1.0 - self.beta2 ** t) ** 0.5 / (1.0 - self.beta1 ** t)
        if hi > lo:
            if self._gpu:
                self._stage(lr_t, self.beta1, self.beta2, self.eps, grad_scale, 0.0, 0.0)
                self._nv.fused_adam(a.params[lo:hi], a.grads[lo:hi], a.slots["m"][lo:hi], a.slots["v"][lo:hi], self._hyper)
            else:
                g = a.grads[lo:hi] * grad_scale if grad_scale != 1.0 else a.grads[lo:hi]
                ref.adam_update(a.params[lo:hi], g, a.slots["m"][lo:hi], a.slots["v"][lo:hi], lr_t,
                                self.beta1, self.beta2, self.eps)
        self.steps = t


class FlatMomentumSGD(_FlatOptimizer):
    slots = ("m",)

    def __init__(self, arena: ParamArena, momentum: float = 0.9, weight_decay: float = 0.0):
        super().__init__(arena)
        self.momentum, self.weight_decay = momentum, weight_decay

    def step(self, lr: float, grad_scale: float = 1.0, lo: Optional[int] = None, hi: Optional[int] = None) -> None:
        lo, hi = self._range(lo, hi)
        a = self.arena
        if hi > lo:
            if self._gpu:
                self._stage(lr, 0.0, 0.0, 0.0, grad_scale, self.momentum, self.weight_decay)
                self._nv.fused_momentum_sgd(a.params[lo:hi], a.grads[lo:hi], a.slots["m"][lo:hi], self._hyper)
            else:
                g = a.grads[lo:hi] * grad_scale if grad_scale != 1.0 else a.grads[lo:hi]
                ref.momentum_update(a.params[lo:hi], g, a.slots["m"][lo:hi], lr, self.momentum, self.weight_decay)
        self.steps += 1
