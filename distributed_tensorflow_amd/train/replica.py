"""One model replica: parameters + forward/backward + optimizer behind a device-neutral interface.

`TorchReplica` runs the plain-PyTorch model (CPU path: plumbing tests, machines without a GPU).
`HipReplica` runs the recorded MI355X step program. Both expose the same contract to the
synchronisation strategies:

  forward_backward(on_bucket, should_abort) fills arena.grads; calls on_bucket(lo, hi) whenever a
      contiguous gradient range is final (so its collective can start while backward continues);
      polls should_abort() at layer/segment boundaries and, if it fires, stops computing and
      zero-fills the gradients not yet produced (short-circuited backward, reference:
      DistributedLeNet/src/short_circuit_gradients.py:487-545).
  apply_update(lr, grad_scale) runs the optimizer over the flat arena (Adam like the reference's
      tf.train.AdamOptimizer, or momentum SGD + weight decay).
"""
from __future__ import annotations

import contextlib
from typing import Callable, List, Optional, Tuple

import torch

from ..models import lenet as lenet_mod
from ..models import resnet as resnet_mod
from ..ops import reference as ref
from ..runtime.arena import ParamArena

BucketFn = Optional[Callable[[int, int], None]]
AbortFn = Optional[Callable[[], bool]]


class _Aborted(Exception):
    pass


class ReplicaBase:
    arena: ParamArena
    model: str
    batch_size: int
    optimizer: str
    updates_applied: int = 0
    last_aborted: bool = False
    timeline = None            # a train.timeline.StepTimeline while steps are being traced

    def span(self, name: str):
        """Timeline span around a phase of the step (no-op unless a timeline is attached)."""
        tl = self.timeline
        if tl is None or not tl.enabled:
            return contextlib.nullcontext()
        return tl.span(name, getattr(self, "stream", None))

    def adam_lr_t(self, lr: float) -> float:
        return ref.adam_lr_t(lr, self.updates_applied + 1)

    def stream_ctx(self):
        """Context in which collectives on this replica's buffers must be issued."""
        return contextlib.nullcontext()

    def num_examples_per_step(self) -> int:
        return self.batch_size


def make_specs(model: str, num_residual_blocks: int = 18):
    if model == "lenet":
        return lenet_mod.lenet_param_specs()
    if model == "resnet":
        return resnet_mod.resnet_param_specs(num_residual_blocks)
    raise ValueError(f"unknown model {model}")


class TorchReplica(ReplicaBase):
    """Reference-op execution with autograd; gradients are copied into the flat arena."""

    def __init__(self, model: str, batch_size: int, *, num_residual_blocks: int = 18, optimizer: str = "adam",
                 device: str = "cpu", momentum: float = 0.9, weight_decay: float = 0.0, seed: int = 66478,
                 num_buckets: int = 1):
        self.model, self.batch_size, self.optimizer = model, batch_size, optimizer
        self.n = num_residual_blocks
        self.momentum, self.weight_decay = momentum, weight_decay
        slots = ("m", "v") if optimizer == "adam" else ("m",)
        self.arena = ParamArena(make_specs(model, num_residual_blocks), device, slots)
        self.arena.initialize(seed)
        self.device = torch.device(device)
        self._images: Optional[torch.Tensor] = None
        self._labels: Optional[torch.Tensor] = None
        self._loss = float("nan")
        self._dropout_gen = torch.Generator().manual_seed(lenet_mod.SEED)
        self.bucket_ranges: List[Tuple[int, int]] = [(0, self.arena.total)]

    def load_batch(self, images, labels) -> None:
        self._images = torch.as_tensor(images, dtype=torch.float32, device=self.device)
        self._labels = torch.as_tensor(labels, device=self.device).long()

    def _loss_fn(self, params, train: bool):
        if self.model == "lenet":
            return lenet_mod.lenet_loss(params, self._images, self._labels, train, self._dropout_gen if train else None)
        return resnet_mod.resnet_loss(params, self._images, self._labels, self.n)

    def forward_backward(self, on_bucket: BucketFn = None, should_abort: AbortFn = None) -> bool:
        a = self.arena
        names = a.names()
        leaves = {k: a.param(k).detach().requires_grad_(True) for k in names}
        self.last_aborted = False

        def getter(name: str) -> torch.Tensor:
            t = leaves[name]
            if should_abort is not None and t.dim() > 1:
                # Poll the abort signal when backprop reaches this layer's weights.
                def hook(g):
                    if should_abort():
                        raise _Aborted()
                    return g
                t = t.clone()
                t.register_hook(hook)
            return t

        a.grads.zero_()
        loss = self._loss_fn(getter, True)
        self._loss = float(loss.detach())
        if should_abort is not None and should_abort():
            self.last_aborted = True
        else:
            try:
                loss.backward()
            except _Aborted:
                self.last_aborted = True
        # Whatever backprop produced before an abort stays; the rest is zero.
        for k in names:
            g = leaves[k].grad
            if g is not None:
                a.grad(k).copy_(g)
        if on_bucket is not None:
            on_bucket(0, a.total)
        return not self.last_aborted

    def apply_update(self, lr: float, grad_scale: float = 1.0) -> None:
        a = self.arena
        g = a.grads * grad_scale if grad_scale != 1.0 else a.grads
        if self.optimizer == "adam":
            ref.adam_update(a.params, g, a.slots["m"], a.slots["v"], self.adam_lr_t(lr))
        else:
            ref.momentum_update(a.params, g, a.slots["m"], lr, self.momentum, self.weight_decay)
        self.updates_applied += 1

    def apply_update_shard(self, lr: float, grad_scale: float, lo: int, hi: int) -> None:
        """Optimizer over the flat range [lo, hi) only (sharded-optimizer mode: this replica owns
        that range of parameters and optimizer slots)."""
        a = self.arena
        p, gr, m = a.full("params")[lo:hi], a.full("grads")[lo:hi], a.full("slot:m")[lo:hi]
        g = gr * grad_scale if grad_scale != 1.0 else gr
        if self.optimizer == "adam":
            ref.adam_update(p, g, m, a.full("slot:v")[lo:hi], self.adam_lr_t(lr))
        else:
            ref.momentum_update(p, g, m, lr, self.momentum, self.weight_decay)
        self.updates_applied += 1

    def zero_grads(self) -> None:
        self.arena.grads.zero_()

    def scale_grads(self, s: float) -> None:
        self.arena.grads.mul_(s)

    def loss_value(self) -> float:
        return self._loss

    def synchronize(self) -> None:
        pass

    @torch.no_grad()
    def evaluate(self, images, labels) -> Tuple[float, int]:
        x = torch.as_tensor(images, dtype=torch.float32, device=self.device)
        y = torch.as_tensor(labels, device=self.device).long()
        params = {k: self.arena.param(k) for k in self.arena.names()}
        if self.model == "lenet":
            logits = lenet_mod.lenet_forward(params, x, train=False)
        else:
            logits = resnet_mod.resnet_forward(params, x, self.n)
        return float(ref.softmax_xent_mean(logits, y)), ref.top1_hits(logits, y)


class HipReplica(ReplicaBase):
    """Recorded MI355X step program (hipGraph replay per segment)."""

    def __init__(self, model: str, batch_size: int, *, num_residual_blocks: int = 18, optimizer: str = "adam",
                 momentum: float = 0.9, weight_decay: float = 0.0, seed: int = 66478, num_buckets: int = 1,
                 use_graph: bool = True, abortable: bool = False, inference_only: bool = False, matmul: Optional[str] = None):
        from ..models.lenet_hip import LeNetHipStep
        from ..models.resnet_hip import ResNetHipStep

        self.model, self.batch_size, self.optimizer = model, batch_size, optimizer
        self.momentum, self.weight_decay = momentum, weight_decay
        self.n = num_residual_blocks
        self.device = torch.device("cuda", torch.cuda.current_device())
        slots = ("m", "v") if optimizer == "adam" else ("m",)
        self.arena = ParamArena(make_specs(model, num_residual_blocks), self.device, slots)
        self.arena.initialize(seed)
        self.stream = torch.cuda.Stream()
        self._eval_steps = {}
        self._device_step = -1      # step number last written to the step program's device counter
        self.ctl_block = 0
        self.step = None
        self.use_graph = False
        self.bucket_ranges = [(0, self.arena.total)]
        if inference_only:
            # an evaluator needs the parameters and an inference program per batch size (built on
            # first use by evaluate()): no activations kept for backward, no gradient buffers
            return
        # Short-circuiting (reference: DistributedLeNet/src/short_circuit_gradients.py:487-545): the
        # kernels of the step are recorded with the addresses of two words of a control block that
        # the update log adopts later; once "updates complete" rises above "updates applied" every
        # kernel of the step returns at once. Without it the kernels carry null pointers and pay nothing.
        self.ctl_block = 0
        if abortable:
            from ..ops import native as nv
            from ..parallel.update_log_hip import abort_words, alloc_control_block

            self.ctl_block = alloc_control_block()
            nv.hip().set_abort_ref(*abort_words(self.ctl_block))
        try:
            with torch.cuda.stream(self.stream):
                if model == "lenet":
                    self.step = LeNetHipStep(batch_size, self.arena, optimizer=optimizer, num_buckets=num_buckets, matmul=matmul)
                else:
                    # a step whose kernels can return early keeps the atomic gradient flush: behind a
                    # short-circuited kernel the optimizer must find the zeros of the step prologue
                    self.step = ResNetHipStep(num_residual_blocks, batch_size, self.arena, optimizer=optimizer, num_buckets=num_buckets,
                                              deterministic=False if abortable else None, matmul=matmul)
        finally:
            if abortable:
                nv.hip().set_abort_ref(0, 0)
        with torch.cuda.stream(self.stream):
            # A step that trails its filter gradients on a side stream is issued from C++ on two
            # real streams: hipGraph replay of the forked graph was measured slower (cross-queue
            # dependencies cost ~10 us each).
            self.use_graph = use_graph and not getattr(self.step, "side_stream", False)
            if self.use_graph:
                self.step.instantiate(per_segment=True)
        self.stream.synchronize()
        self.bucket_ranges = list(self.step.bucket_ranges)

    def stream_ctx(self):
        return torch.cuda.stream(self.stream)

    def load_batch(self, images, labels) -> None:
        if isinstance(images, torch.Tensor) and images.is_cuda:
            # device tensors may still be being produced on the caller's stream
            self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            x = torch.as_tensor(images, dtype=torch.float32)
            y = torch.as_tensor(labels)
            self.step.load_batch(x.to(self.device, non_blocking=True), y.to(self.device, non_blocking=True))

    def forward_backward(self, on_bucket: BucketFn = None, should_abort: AbortFn = None) -> bool:
        """Enqueues forward + backward, segment by segment (a segment ends where a gradient bucket is
        final). Short-circuiting is not a host decision here: a replica built with `abortable=True`
        records every kernel with an abort guard that reads the update log's words on the device, so
        `should_abort` is not consulted and the host never waits for the GPU to find out how far the
        pass got. Always returns True."""
        st = self.step
        self.last_aborted = False
        nseg = st.num_segments
        with torch.cuda.stream(self.stream):
            # the optimizer launch of the previous update normally left the step number on the device
            # already (apply_update); a restore or an update applied by the update log's kernels did not
            if self._device_step != self.updates_applied:
                st.set_step_counter(self.updates_applied)
                self._device_step = self.updates_applied
            for i in range(nseg - 1):
                self._run_segment(i)
                if on_bucket is not None:
                    on_bucket(*st.bucket_ranges[i])
        return True

    def _run_segment(self, seg: int) -> None:
        """Replays one segment; while a timeline is attached, launch by launch with per-kernel events."""
        tl = self.timeline
        if tl is not None and tl.enabled:
            tl.add_kernel_events(self.step.run_traced(seg, tl.base_event_handle))
        else:
            self.step.run(seg, self.use_graph)

    def apply_update(self, lr: float, grad_scale: float = 1.0) -> None:
        st = self.step
        with torch.cuda.stream(self.stream):
            lr_t = self.adam_lr_t(lr) if self.optimizer == "adam" else lr
            st.set_hyper(lr_t=lr_t, grad_scale=grad_scale, momentum=self.momentum, weight_decay=self.weight_decay,
                         step=self.updates_applied + 1)
            self._run_segment(st.num_segments - 1)
        self.updates_applied += 1
        self._device_step = self.updates_applied

    def apply_update_shard(self, lr: float, grad_scale: float, lo: int, hi: int) -> None:
        """Fused optimizer launch over the flat range [lo, hi) only (sharded-optimizer mode)."""
        from ..ops import native as nv

        st, a = self.step, self.arena
        with torch.cuda.stream(self.stream):
            lr_t = self.adam_lr_t(lr) if self.optimizer == "adam" else lr
            st.set_hyper(lr_t=lr_t, grad_scale=grad_scale, momentum=self.momentum, weight_decay=self.weight_decay,
                         step=self.updates_applied + 1)
            if hi > lo:
                p, g, m = a.full("params")[lo:hi], a.full("grads")[lo:hi], a.full("slot:m")[lo:hi]
                if self.optimizer == "adam":
                    nv.fused_adam(p, g, m, a.full("slot:v")[lo:hi], st.hyper)
                else:
                    nv.fused_momentum_sgd(p, g, m, st.hyper)
        self.updates_applied += 1
        self._device_step = self.updates_applied

    def zero_grads(self) -> None:
        with torch.cuda.stream(self.stream):
            self.arena.grads.zero_()

    def scale_grads(self, s: float) -> None:
        from ..ops import native as nv

        with torch.cuda.stream(self.stream):
            nv.flat_scale(self.arena.grads, s)

    def loss_value(self) -> float:
        self.stream.synchronize()
        return self.step.loss_value()

    def synchronize(self) -> None:
        self.stream.synchronize()

    def evaluate(self, images, labels) -> Tuple[float, int]:
        """Inference-only program over the same arena (BatchNorm uses the eval batch's statistics,
        like the reference's evaluator)."""
        from ..models.lenet_hip import LeNetHipStep
        from ..models.resnet_hip import ResNetHipStep

        b = len(labels)
        with torch.cuda.stream(self.stream):
            ev = self._eval_steps.get(b)
            if ev is None:
                if self.model == "lenet":
                    ev = LeNetHipStep(b, self.arena, training=False)
                else:
                    ev = ResNetHipStep(self.n, b, self.arena, training=False)
                self._eval_steps[b] = ev
            ev.load_batch(torch.as_tensor(images, dtype=torch.float32).to(self.device), torch.as_tensor(labels).to(self.device))
            ev.run(-1, False)
        self.stream.synchronize()
        return ev.loss_value(), ev.correct_count()


def make_replica(model: str, batch_size: int, device: str = "auto", **kw) -> ReplicaBase:
    use_gpu = device == "cuda" or (device == "auto" and torch.cuda.is_available())
    if use_gpu:
        kw.pop("device", None)
        return HipReplica(model, batch_size, **kw)
    kw.pop("use_graph", None)
    kw.pop("abortable", None)
    kw.pop("inference_only", None)
    kw.pop("matmul", None)
    return TorchReplica(model, batch_size, device="cpu", **kw)
