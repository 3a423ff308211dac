"""MI355X training/eval step of the MNIST LeNet, recorded as a step program.

forward   conv1 + bias + ReLU + pool = one kernel (its 28x28x32 activation never reaches memory),
          conv2 = whole-image 5x5 MFMA kernel (input planes resident in LDS, filter streamed from
          L2) with fused bias + ReLU + max pool, pools that
          keep a 2-bit argmax, FC1 = split-K MFMA GEMM; FC1's bias + ReLU, Philox dropout, FC2,
          softmax cross-entropy and their backward passes are one kernel (fc_head) that keeps
          the hidden activations in registers.
backward  GEMMs (NT / TN forms) for FC1, ReLU masks (the pooled maximum doubles as the ReLU mask of
          its winner), max-pool scatter, conv data/filter gradients, column sums for the biases;
          gradients land in the flat arena, one fused Adam launch.

Reference behaviour reproduced: DistributedLeNet/src/mnist.py:76-164 (model, loss, top-1 count)
and DistributedLeNet/src/distributed_train.py:164 (Adam).
"""
from __future__ import annotations

from typing import Optional

import torch

from ..ops import native as nv
from ..runtime.arena import ParamArena
from ..runtime.step_program import StepProgramBase
from .lenet import DROPOUT_KEEP, NUM_LABELS, SEED, lenet_param_specs


class LeNetHipStep(StepProgramBase):
    def __init__(self, batch: int, arena: Optional[ParamArena] = None, *, optimizer: str = "adam", num_buckets: int = 1,
                 training: bool = True, device: str | torch.device = "cuda", dropout: bool = True, matmul: Optional[str] = None):
        nv.require()
        import os

        # matmul="bf16x6" (or DTF_MATMUL): the second convolution's forward pass, data gradient and filter
        # gradient run on the bf16 matrix pipe with fp32 operands split three ways and fp32 accumulation
        # (csrc/kernels/conv5x5_split.hip); opt-in, everything else stays on the exact-fp32 instruction
        self.matmul = matmul or os.environ.get("DTF_MATMUL", "fp32")
        if self.matmul not in ("fp32", "bf16x6"):
            raise ValueError("matmul must be 'fp32' or 'bf16x6'")
        self.batch, self.training, self.optimizer = batch, training, optimizer
        self.device = torch.device(device)
        self.dropout = dropout and training
        if arena is None:
            slots = ("m", "v") if optimizer == "adam" else ("m",)
            arena = ParamArena(lenet_param_specs(), self.device, slots if training else ())
            arena.initialize()
        self.arena = arena
        B, dev = batch, self.device
        f32 = dict(device=dev, dtype=torch.float32)
        self._init_common((28, 28, 1))
        a = arena
        split = self.matmul == "bf16x6"
        nw = int(nv.hip().conv5x5_split_packed_floats()) if split else 25 * 32 * 64
        self.wpk2 = torch.empty(nw, **f32)   # [tap][out][in], the whole-image kernel's operand (or its split-operand packing)
        self.wpk2_t = torch.empty(nw, **f32) if split and training else None   # data-gradient packing
        self.p1 = torch.empty((B, 14, 14, 32), **f32)
        self.arg1 = torch.empty((B, 14, 14, 32), device=dev, dtype=torch.uint8)
        self.p2 = torch.empty((B, 7, 7, 64), **f32)
        self.arg2 = torch.empty((B, 7, 7, 64), device=dev, dtype=torch.uint8)
        self.h = torch.empty((B, 512), **f32)
        self.logits = torch.empty((B, NUM_LABELS), **f32)
        if training:
            self.dh = torch.empty((B, 512), **f32)
            self.dp2 = torch.empty((B, 7 * 7 * 64), **f32)
            self.dp1 = torch.empty((B, 14, 14, 32), **f32)
        self._record(num_buckets)
        self.set_hyper(lr_t=0.0)
        self.set_step_counter(0)

    def _record(self, num_buckets: int) -> None:
        a = self.arena
        p = self.program
        B = self.batch
        p.begin_record()
        try:
            p.mark_phase("prologue")
            # FC1 is a k-split product that accumulates into its output: cleared with everything else
            self._record_prologue(also_zero=[self.h])
            p.mark_phase("forward")
            split = self.matmul == "bf16x6"
            if split:
                nv.conv5x5_split_pack(a.param("conv2_weights"), out=self.wpk2)
                if self.training:
                    nv.conv5x5_split_pack(a.param("conv2_weights"), flip=True, out=self.wpk2_t)
            else:
                nv.conv5x5_pack(a.param("conv2_weights"), self.wpk2)
            # ---------------- forward
            # conv1 + bias + ReLU + pool in one kernel: the 28x28x32 activation is never written
            nv.conv1_pool_fwd(self.images, a.param("conv1_weights"), a.param("conv1_biases"), out=self.p1, arg=self.arg1)
            # conv2 + bias + ReLU + pool: the 14x14x64 activation stays in LDS
            nv.conv5x5_tile(self.p1, self.wpk2, bias=a.param("conv2_biases"), pool_out=(self.p2, self.arg2), split=split)
            flat = self.p2.view(B, -1)
            # FC1's raw product; its bias, ReLU and dropout, FC2, the loss and the backward pass down
            # to FC1's pre-activation are one kernel, so the hidden activations never reach memory
            nv.gemm(flat, a.param("fc1_weights"), out=self.h, accumulate=True)
            keep = DROPOUT_KEEP if self.dropout else 1.0
            head = dict(keep_prob=keep, seed=SEED, offset=0, offset_dev=self.step_counter[0:1])
            if self.training:
                # the arena's gradients were zeroed by the step prologue: accumulating into them
                # spares the split-K products their own clearing pass
                head.update(dh=self.dh, dw2=a.grad("fc2_weights"), db2=a.grad("fc2_biases"), db1=a.grad("fc1_biases"))
            nv.fc_head(self.h, a.param("fc1_biases"), a.param("fc2_weights"), a.param("fc2_biases"), self.labels,
                       self.logits, self.loss, self.correct, **head)
            if self.training:
                # ---------------- backward
                p.mark_phase("backward")
                nv.gemm(flat, self.dh, trans_a=True, out=a.grad("fc1_weights"), accumulate=True)
                # the FC block holds 97 % of the parameters: its bucket is final here
                if num_buckets > 1:
                    lo = a.offsets["fc1_weights"]
                    self.bucket_ranges.append((lo, a.total))
                    p.mark_segment()
                else:
                    lo = a.total
                nv.gemm(self.dh, a.param("fc1_weights"), trans_b=True, out=self.dp2)
                # The gradient at conv2's output is the pool + ReLU backward of dp2; both conv2 backward
                # kernels form it while staging, so it is never written. Its per-channel sum (the bias
                # gradient) is the sum of dp2 over the cells whose maximum survived the ReLU.
                dp2 = self.dp2.view(B, 7, 7, 64)
                # ReLU mask of the pooled maximum and conv2's bias gradient in one pass over dp2
                nv.col_sum(dp2.view(-1, 64), a.grad("conv2_biases"), relu_of=self.p2, masked=dp2)
                pooled2 = (dp2, self.arg2, self.p2)
                # filter gradients accumulate in per-XCD copies of the conv part of the arena
                conv_end = a.offsets["fc1_weights"]
                if not hasattr(self, "_wslots"):
                    self._wslots = torch.zeros(nv.STAT_SLOTS, conv_end, device=self.device, dtype=torch.float32)

                def slot_view(name):
                    off, spec = a.offsets[name], a.spec(name)
                    return self._wslots[0, off:off + spec.numel].view(spec.shape)

                # conv2 filter gradient: per-workgroup partial results, one reduction into the arena
                if not hasattr(self, "_w2_parts"):
                    self._w2_parts = torch.empty(nv.conv5x5_wgrad_parts(B), 25 * 32 * 64, device=self.device, dtype=torch.float32)
                nv.conv5x5_wgrad(self.p1, None, a.grad("conv2_weights"), partials=self._w2_parts, pooled=pooled2, split=split)
                # data gradient: the same kernel over the unpacked filter, taps walked backwards
                if split:
                    nv.conv5x5_tile(None, self.wpk2_t, flip=True, out=self.dp1, pooled=pooled2, split=True)
                else:
                    nv.conv5x5_tile(None, a.param("conv2_weights"), flip=True, out=self.dp1, pooled=pooled2)
                # pool backward + ReLU mask + filter and bias gradients of conv1 in one kernel
                nv.conv1_pool_bwd(self.images, self.dp1, self.arg1, self.p1, slot_view("conv1_weights"),
                                  slot_view("conv1_biases"), slot_stride=conv_end)
                nv.slot_reduce(a.grads, self._wslots, 0, conv_end)
                self.bucket_ranges.append((0, lo))
                p.mark_segment()
                p.mark_phase("optimizer")
                self._record_optimizer()
        finally:
            p.end_record()
        self.num_segments = p.num_segments
