"""Common machinery of recorded GPU training steps (shared by the ResNet and LeNet programs).

A step program owns: the flat parameter arena, static input buffers (the batch is copied in,
the program always reads the same addresses so it can be replayed as a hipGraph), a small device
buffer of step-dependent scalars, the scalar outputs (loss, top-1 count), and the recorded
`_hip.Program`. Segments of the program end where a gradient bucket becomes final.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch

from ..ops import native as nv


class StepProgramBase:
    SCALAR_SLOTS = 64
    # set by subclasses before _init_common(): self.arena, self.device, self.batch, self.training, self.optimizer
    def _init_common(self, image_shape: Tuple[int, ...]) -> None:
        dev, B = self.device, self.batch
        f32 = dict(device=dev, dtype=torch.float32)
        self.images = torch.zeros((B,) + tuple(image_shape), **f32)
        self.labels = torch.zeros(B, device=dev, dtype=torch.int32)
        # loss and top-1 count are sums over SCALAR_SLOTS elements: kernels that add one value per
        # image spread them, since atomics on a single address serialise
        self.scalars = torch.zeros(self.SCALAR_SLOTS, **f32)
        self.loss = self.scalars[0:1]
        self._correct_buf = torch.zeros(self.SCALAR_SLOTS, device=dev, dtype=torch.int32)
        self.correct = self._correct_buf[0:1]
        self.hyper = torch.zeros(nv.HYPER_SLOTS, **f32)
        self.step_counter = torch.zeros(2, device=dev, dtype=torch.int64)
        self.bucket_ranges: List[Tuple[int, int]] = []
        self.program = nv.hip().Program()
        self.num_segments = 1

    def _record_prologue(self, also_zero=()) -> None:
        """Clears everything a step accumulates into (loss, top-1 count, gradients, `also_zero`) in
        one launch; buffers that are not 16-byte granular get a memset of their own."""
        bufs = [self.scalars, self._correct_buf] + list(also_zero)
        if self.training:
            bufs.append(self.arena.grads)
        batch = []
        for t in bufs:
            if t.numel() == 0:
                continue
            if t.data_ptr() % 16 == 0 and (t.numel() * t.element_size()) % 16 == 0:
                batch.append(t)
            else:
                nv.memset_zero(t)
        for i in range(0, len(batch), nv.ZERO_MULTI_MAX):
            nv.zero_multi(batch[i:i + nv.ZERO_MULTI_MAX])

    def _record_optimizer(self) -> None:
        a = self.arena
        if self.optimizer == "adam":
            nv.fused_adam(a.params, a.grads, a.slots["m"], a.slots["v"], self.hyper)
        else:
            nv.fused_momentum_sgd(a.params, a.grads, a.slots["m"], self.hyper)

    # ------------------------------------------------------------------------------------------
    def set_hyper(self, lr_t: float, beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8,
                  grad_scale: float = 1.0, momentum: float = 0.9, weight_decay: float = 0.0,
                  step: Optional[int] = None) -> None:
        """Writes the step-dependent scalars in stream order, before the next replay. The values
        ride in the launch itself: the host may run any number of steps ahead of the GPU without
        one step's values leaking into another's (a shared pinned staging buffer would allow that).
        step: also sets the device-visible step number (see set_step_counter) in the same launch."""
        values = (lr_t, beta1, beta2, eps, grad_scale, momentum, weight_decay)
        if step is None:
            nv.set_scalars(self.hyper, values)
        else:
            nv.set_scalars(self.hyper, values, u_dst=self.step_counter, u_values=(int(step) * (1 << 20),))

    def set_step_counter(self, step: int) -> None:
        """Device-visible step number (dropout draws a fresh mask per step from it)."""
        nv.set_scalars(u_dst=self.step_counter, u_values=(int(step) * (1 << 20),))

    def load_batch(self, images: torch.Tensor, labels: torch.Tensor) -> None:
        self.images.copy_(images.reshape(self.images.shape), non_blocking=True)
        self.labels.copy_(labels.to(torch.int32), non_blocking=True)

    def instantiate(self, per_segment: bool = False) -> None:
        s = torch.cuda.current_stream().cuda_stream
        if per_segment:
            for i in range(self.num_segments):
                self.program.instantiate(s, i)
        else:
            self.program.instantiate(s, -1)

    def run(self, seg: int = -1, graph: bool = True) -> None:
        """Enqueues the whole step (seg < 0) or one segment on the current stream."""
        s = torch.cuda.current_stream().cuda_stream
        if graph and self.program.instantiated(seg):
            self.program.replay(s, seg)
        else:
            self.program.run(s, seg)

    def run_traced(self, seg: int = -1, base_event: int = 0):
        """Runs the whole step (seg < 0) or one segment launch by launch, each bracketed by timing
        events, and returns [(kernel name, lane, phase name, start us, duration us, blocks, threads)];
        start is relative to `base_event` (a raw hipEvent handle recorded earlier on this stream)."""
        s = torch.cuda.current_stream().cuda_stream
        phases = self.program.phase_names()
        return [(n, lane, phases[ph] if ph >= 0 else "", ts, dur, blocks, threads)
                for n, lane, ph, ts, dur, blocks, threads in self.program.run_traced(s, seg, base_event)]

    def zero_grad_range(self, lo: int, hi: int) -> None:
        """Zeroes grads[lo:hi] on the current stream (short-circuited backward)."""
        if hi > lo:
            nv.memset_zero(self.arena.grads[lo:hi])

    def loss_value(self) -> float:
        return float(self.scalars.sum().item())

    def correct_count(self) -> int:
        return int(self._correct_buf.sum().item())

    @property
    def kernel_count(self) -> int:
        return self.program.num_records
