"""Fused MI355X training/eval step of the CIFAR ResNet, recorded as a step program.

The whole step is a flat list of kernel launches over pre-allocated buffers (51 for ResNet-20):

  forward   one implicit-GEMM conv kernel per layer; the producer's BatchNorm+ReLU is applied
            while the input tile is staged into LDS, the residual add (identity / option-A /
            BN+ReLU'd stem) and the NEXT BatchNorm's batch statistics are fused into the
            epilogue. No normalised activation is ever written to HBM.
  head      BN+ReLU+global-average-pool+FC+softmax-cross-entropy in one kernel.
  backward  one kernel per stride-1 conv: masked data gradient + BatchNorm-backward's two
            reductions + filter gradient; the BatchNorm backward of the layer after it and the
            shortcut gradient are applied while that kernel stages its output gradient (the block
            input gradient is written once, by the convolution that consumes it). Stride-2 convs
            use separate data- / filter-gradient kernels and an elementwise BatchNorm-backward
            pass; the stem's filter-gradient kernel applies BN0's backward in its prologue.
  update    one fused Adam (or momentum-SGD + weight decay) launch over the flat arena.

The program is recorded once (no launches), then replayed either from C++ or as a hipGraph.
Segment marks split the backward pass at stage boundaries so the gradient buckets of the
already finished layers can be all-reduced while the rest of backward still runs.

Reference behaviour reproduced: DistributedResNet/src/resnet.py (model),
DistributedResNet/src/distributed_train.py:119-129,237,256 (loss, Adam, gradients).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import os

import numpy as np
import torch

from ..ops import native as nv
from ..ops.reference import BN_EPSILON, same_pad
from ..runtime.arena import ParamArena
from ..runtime.step_program import StepProgramBase
from .resnet import NUM_CLASSES, resnet_blocks, resnet_param_specs

PACK_DTYPE = np.dtype(
    [("w_off", "<i8"), ("out_off", "<i8"), ("ks", "<i4"), ("Cin", "<i4"), ("Cout", "<i4"), ("flip", "<i4"),
     ("n_floats", "<i4"), ("first_block", "<i4")]
)


@dataclass
class _Bn:
    fwd: nv.BnState        # statistics of the BN input tensor
    gsums: torch.Tensor    # float64 statistics block of the backward reductions
    gamma_name: str
    beta_name: str


class ResNetHipStep(StepProgramBase):
    """Owns the buffers and the recorded program of one replica's training step."""

    def __init__(self, n: int, batch: int, arena: Optional[ParamArena] = None, *, optimizer: str = "adam",
                 num_buckets: int = 1, num_classes: int = NUM_CLASSES, training: bool = True,
                 device: str | torch.device = "cuda", deterministic: Optional[bool] = None, matmul: Optional[str] = None):
        nv.require()
        if optimizer not in ("adam", "momentum"):
            raise ValueError("optimizer must be 'adam' or 'momentum'")
        self.n, self.batch, self.num_classes = n, batch, num_classes
        self.device = torch.device(device)
        self.training = training
        self.optimizer = optimizer
        # "fp32": every convolution on the exact-fp32 matrix instruction (the default and the headline).
        # "bf16x6": opt-in; the forward pass of the stride-1 layers runs on the bf16 matrix pipe with both
        # operands split three ways and fp32 accumulation (csrc/kernels/conv_fprop_split.hip) - same
        # results to within fp32 rounding, measured against fp64 in profiles/.
        self.matmul = matmul or os.environ.get("DTF_MATMUL", "fp32")
        if self.matmul not in ("fp32", "bf16x6"):
            raise ValueError("matmul must be 'fp32' or 'bf16x6'")
        if arena is None:
            slots = ("m", "v") if optimizer == "adam" else ("m",)
            arena = ParamArena(resnet_param_specs(n, num_classes), self.device, slots if training else ())
            arena.initialize()
        self.arena = arena
        self.blocks = resnet_blocks(n)
        dev, B = self.device, batch
        f32 = dict(device=dev, dtype=torch.float32)

        self._init_common((32, 32, 3))
        self.feat = torch.empty((B, 64), **f32)
        self.logits = torch.empty((B, num_classes), **f32)
        self.dlogits = torch.empty((B, num_classes), **f32)

        # ---- activations kept for backward: raw conv outputs / block sums only
        self.c0 = torch.empty((B, 32, 32, 16), **f32)
        self.c1: List[torch.Tensor] = []
        self.y: List[torch.Tensor] = []
        for blk in self.blocks:
            hw = blk.conv2.in_hw
            self.c1.append(torch.empty((B, hw, hw, blk.conv1.cout), **f32))
            self.y.append(torch.empty((B, hw, hw, blk.conv2.cout), **f32))

        # ---- BatchNorm statistics arena (fp64), zeroed once per step
        bn_channels = [16] + [c for blk in self.blocks for c in (([] if blk.first_block else [blk.conv1.cin]) + [blk.conv2.cin])] + [64]
        self.stats = torch.zeros(sum(2 * nv.STAT_SLOTS * 2 * c for c in bn_channels), device=dev, dtype=torch.float64)
        self._stat_off = 0
        a = arena
        self.bn0 = self._make_bn(16, B * 32 * 32, "conv0/gamma", "conv0/beta")
        self.bn1: List[Optional[_Bn]] = []
        self.bn2: List[_Bn] = []
        for blk in self.blocks:
            if blk.first_block:
                self.bn1.append(None)
            else:
                hw = blk.conv1.in_hw
                self.bn1.append(self._make_bn(blk.conv1.cin, B * hw * hw, blk.conv1.bn_gamma, blk.conv1.bn_beta))
            hw = blk.conv2.in_hw
            self.bn2.append(self._make_bn(blk.conv2.cin, B * hw * hw, blk.conv2.bn_gamma, blk.conv2.bn_beta))
        self.bnf = self._make_bn(64, B * 8 * 8, "fc/gamma", "fc/beta")

        # ---- packed filters (forward order, and flipped/transposed for the data gradient)
        entries = []
        off = 0
        self._pk_fwd: Dict[str, Tuple[int, int]] = {}
        self._pk_bwd: Dict[str, Tuple[int, int]] = {}
        convs = [("conv0/conv", 3, 16, 1)] + [(c.weight, c.cin, c.cout, c.stride) for blk in self.blocks for c in (blk.conv1, blk.conv2)]
        in_hw = {c.weight: c.in_hw for blk in self.blocks for c in (blk.conv1, blk.conv2)}
        # the split-operand backward kernels need the fused backward pass and static band scheduling
        self._split_backward = (os.environ.get("DTF_SPLIT_BWD", "1") != "0" and os.environ.get("DTF_BWD_FUSED", "1") != "0"
                                and os.environ.get("DTF_DYNAMIC_TILES", "0") == "0")
        self._split_fwd = set()     # layers whose forward pass runs on the split-operand kernel
        self._split_bwd = set()     # layers whose fused backward pass does
        for name, cin, cout, stride in convs:
            if self.matmul == "bf16x6" and name != "conv0/conv" and nv.split_supported(cin, cout, stride):
                nfl = int(nv.hip().packed_split_floats(cin))
                self._split_fwd.add(name)
                entries.append((a.offsets[name], off, 3, cin, cout, 2, nfl))
            else:
                nfl = nv.packed_floats(3, cin, cout, False)
                entries.append((a.offsets[name], off, 3, cin, cout, 0, nfl))
            self._pk_fwd[name] = (off, nfl)
            off += nfl
            if training and name != "conv0/conv":
                hw = in_hw[name]
                if name in self._split_fwd and self._split_backward and nv.hip().conv_bwd_split_supported(B, hw, hw, cin):
                    nfl = int(nv.hip().packed_split_floats(cin))
                    self._split_bwd.add(name)
                    entries.append((a.offsets[name], off, 3, cin, cout, 3, nfl))
                else:
                    nfl = nv.packed_floats(3, cin, cout, True)
                    entries.append((a.offsets[name], off, 3, cin, cout, 1, nfl))
                self._pk_bwd[name] = (off, nfl)
                off += nfl
        self.packed = torch.empty(off, **f32)
        table = np.zeros(len(entries), dtype=PACK_DTYPE)
        blk0 = 0
        for i, (w_off, out_off, ks, cin, cout, flip, nfl) in enumerate(entries):
            table[i] = (w_off, out_off, ks, cin, cout, flip, nfl, blk0)
            blk0 += (nfl + 255) // 256
        assert PACK_DTYPE.itemsize == nv.hip().PACK_ENTRY_BYTES
        self._pack_blocks = blk0
        self._pack_table = torch.from_numpy(table.view(np.uint8).copy()).to(dev)
        self._pack_n = len(entries)

        # ---- backward scratch: one gradient buffer per tensor, never reused within a step, so the
        # filter-gradient kernels can trail the data-gradient chain on a side stream without
        # write-after-read hazards (HBM is plentiful: ~2x the activations)
        if training:
            self._d_out = [torch.empty_like(y) for y in self.y]      # gradient of each block output
            self._d_c1 = [torch.empty_like(c) for c in self.c1]       # gradient through conv2 / BN2
            self._d_c0 = torch.empty_like(self.c0)
            # stride-1 layers run data gradient + filter gradient + BatchNorm sums as ONE kernel
            self.fused_backward = os.environ.get("DTF_BWD_FUSED", "1") != "0"
            # work counters of the fused backward kernels (dynamic band scheduling), one block of
            # counters per layer, zeroed with the statistics at the start of every step. Opt-in:
            # alone on the GPU the static walk is 3 % faster (3.06 vs 3.15 ms/step at batch 1024);
            # dynamic hand-out is for CUs shared with a collective or another process.
            self.dynamic_tiles = os.environ.get("DTF_DYNAMIC_TILES", "0") != "0"
            self._tile_ctr = torch.zeros(2 * len(self.blocks) * nv.hip().BWD_TILE_COUNTERS, device=dev, dtype=torch.int32)
            self._tile_ctr_used = 0
            self._g_raw = {}
            # With separate filter-gradient kernels a side stream hides them behind the data-gradient
            # chain (1.40 -> 1.24 ms/step at batch 256); with the fused kernels only three filter
            # gradients remain and the second stream measures slightly slower (1.23 vs 1.19 ms).
            self.side_stream = os.environ.get("DTF_SIDE_STREAM", "0" if self.fused_backward else "1") != "0"
            # per-XCD copies of the filter gradients (see ConvWgradArgs::dw_slot_stride), summed into
            # the gradient arena at the end of every bucket segment
            self._wslots = torch.zeros(nv.STAT_SLOTS, arena.total, **f32)
            # Deterministic gradients: every workgroup stores its part of a filter gradient into a copy
            # of its own and one launch per bucket adds the copies in a fixed order (nv.GradientParts),
            # instead of float atomics in arrival order. Default on; DTF_DETERMINISTIC=0 (or
            # deterministic=False: replicas whose kernels short-circuit, which must find zeros, not
            # last step's copies, behind a kernel that returned early) keeps the atomics.
            if deterministic is None:
                deterministic = os.environ.get("DTF_DETERMINISTIC", "1") != "0"
            self.deterministic = bool(deterministic)
            self._parts = nv.GradientParts(self._wslots) if self.deterministic else None

        self._record(num_buckets)
        self.set_hyper(lr_t=0.0)

    # ------------------------------------------------------------------------------------------
    def _make_bn(self, c: int, count: int, gamma: str, beta: str) -> _Bn:
        o, n = self._stat_off, nv.STAT_SLOTS * 2 * c
        self._stat_off += 2 * n
        st = nv.BnState(self.stats[o:o + n], self.arena.param(gamma), self.arena.param(beta), count, BN_EPSILON)
        return _Bn(st, self.stats[o + n:o + 2 * n], gamma, beta)

    def _summary_items(self):
        items = [("conv0", self.c0, self.bn0)]
        for k in range(len(self.blocks)):
            items.append((f"block{k}/conv1", self.c1[k], self.bn2[k]))
            nxt = self.bn1[k + 1] if k + 1 < len(self.blocks) else self.bnf
            items.append((f"block{k}/out", self.y[k], nxt))
        return items

    def activation_sparsity(self) -> Dict[str, float]:
        """Zero fraction of every ReLU output (tf.nn.zero_fraction in the reference's
        activation_summary, DistributedResNet/src/resnet.py:13-20). The normalised activations
        are never materialised by the step, so they are recomputed here (summary cadence only)."""
        return {f"{name}/sparsity": nv.zero_fraction(nv.bn_apply(t, bn.fwd, relu=True)) for name, t, bn in self._summary_items()}

    def activation_histograms(self) -> Dict[str, dict]:
        """tf.summary.histogram of the same activations (the other half of activation_summary)."""
        return {f"{name}/activations": nv.tf_histogram(nv.bn_apply(t, bn.fwd, relu=True)) for name, t, bn in self._summary_items()}

    def _wgrad(self, x: torch.Tensor, dout: torch.Tensor, name: str, stride: int, in_bn=None) -> None:
        a = self.arena
        off, spec = a.offsets[name], a.spec(name)
        dw = self._wslots[0, off:off + spec.numel].view(spec.shape)
        nv.conv_wgrad(x, dout, dw, stride=stride, in_bn=in_bn, slot_stride=a.total, parts=self._parts)

    def _wpk(self, name: str, bwd: bool = False) -> torch.Tensor:
        off, n = (self._pk_bwd if bwd else self._pk_fwd)[name]
        return self.packed[off:off + n]

    def _view(self, buf: torch.Tensor, shape) -> torch.Tensor:
        n = 1
        for s in shape:
            n *= s
        return buf[:n].view(shape)

    # ------------------------------------------------------------------------------------------
    def _record_forward(self) -> None:
        a = self.arena
        nv.hip().pack_multi(a.params.data_ptr(), self.packed.data_ptr(), self._pack_table.data_ptr(), self._pack_n,
                            self._pack_blocks, nv.stream())
        # stem: dedicated 3 -> 16 kernel over the plain filter (K = 27)
        nv.conv_stem_fwd(self.images, a.param("conv0/conv"), out=self.c0, out_sums=self.bn0.fwd.sums)
        x, x_bn = self.c0, self.bn0
        for k, blk in enumerate(self.blocks):
            bn1 = self.bn0 if blk.first_block else self.bn1[k]
            bn2 = self.bn2[k]
            nxt = self.bn1[k + 1] if k + 1 < len(self.blocks) else self.bnf
            nv.conv_fprop(x, self._wpk(blk.conv1.weight), ks=3, cout=blk.conv1.cout, stride=blk.conv1.stride,
                          in_bn=bn1.fwd, out=self.c1[k], out_sums=bn2.fwd.sums, split=blk.conv1.weight in self._split_fwd)
            if blk.first_block:
                res = dict(res_mode=nv.RES_BNRELU, res=x, res_bn=self.bn0.fwd)
            elif blk.increase_dim:
                res = dict(res_mode=nv.RES_SHORTCUT_A, res=x)
            else:
                res = dict(res_mode=nv.RES_IDENT, res=x)
            nv.conv_fprop(self.c1[k], self._wpk(blk.conv2.weight), ks=3, cout=blk.conv2.cout, in_bn=bn2.fwd,
                          out=self.y[k], out_sums=nxt.fwd.sums, split=blk.conv2.weight in self._split_fwd, **res)
            x = self.y[k]
        nv.head_fwd(x, self.bnf.fwd, a.param("fc/fc_weights"), a.param("fc/fc_bias"), self.labels, self.feat,
                    self.logits, self.dlogits, self.scalars, self._correct_buf)

    def _record_backward(self, num_buckets: int) -> None:
        a = self.arena
        blocks = self.blocks
        # bucket split points (float offsets into the flat arena), by parameter volume
        marks: Dict[int, int] = {}
        if num_buckets > 1:
            total = a.total
            targets = [total * i // num_buckets for i in range(1, num_buckets)]
            for tgt in targets:
                # the block whose first parameter offset is closest to the target from above
                best = None
                for k, blk in enumerate(blocks):
                    first = blk.conv1.bn_beta or blk.conv1.weight
                    o = a.offsets[first]
                    if o >= tgt and (best is None or o < best[1]):
                        best = (k, o)
                if best is not None and best[0] > 0:
                    marks[best[0]] = best[1]
        hi = a.total

        p = self.program
        side = self.side_stream

        def wgrad(x, dout, name, stride, in_bn=None):
            # filter gradients are leaves of the dependency graph: they run on lane 1 behind the
            # kernel that produced `dout`, overlapping the data-gradient chain on lane 0
            if side:
                p.sync_lanes(0, 1)
                p.set_lane(1)
            self._wgrad(x, dout, name, stride, in_bn)
            if side:
                p.set_lane(0)

        def dw_slot_view(name):
            off, spec = a.offsets[name], a.spec(name)
            return self._wslots[0, off:off + spec.numel].view(spec.shape)

        def fusable(x, conv):
            return self.fused_backward and conv.stride == 1 and conv.cin == conv.cout and nv.conv_bwd_fused_supported(x)

        def layer_backward(x, dout, conv, bn, out, extra=None, dout_bn=None, shortcut=None):
            """Gradient at the BatchNorm output of the layer's input (masked, with the BN-backward
            sums) and the layer's filter gradient. dout_bn = (y, _Bn): `dout` is the masked gradient
            at the output of the BatchNorm after this convolution and that BatchNorm's backward
            is applied on the fly (fused kernel only); shortcut = (res, dy_out) adds the gradient of
            an identity shortcut to it and writes the sum to dy_out."""
            s = conv.stride
            if fusable(x, conv):
                kw = {}
                if dout_bn is not None:
                    y_t, ybn = dout_bn
                    kw = dict(dy_bn=(y_t, ybn.fwd, ybn.gsums), dy_dgamma=a.grad(ybn.gamma_name), dy_dbeta=a.grad(ybn.beta_name))
                    if shortcut is not None:
                        kw.update(dy_res=shortcut[0], dy_out=shortcut[1], dy_res_option_a=shortcut[2])
                if self.dynamic_tiles:
                    nctr = nv.hip().BWD_TILE_COUNTERS
                    kw["tile_counters"] = self._tile_ctr[self._tile_ctr_used:self._tile_ctr_used + nctr]
                    self._tile_ctr_used += nctr
                return nv.conv_bwd_fused(dout, x, bn.fwd, self._wpk(conv.weight, True), dw_slot_view(conv.weight), bn.gsums,
                                         slot_stride=a.total, extra=extra, out=out, parts=self._parts,
                                         split=conv.weight in self._split_bwd, **kw)
            assert dout_bn is None and shortcut is None
            wgrad(x, dout, conv.weight, s, bn.fwd)
            _, pb, _ = same_pad(x.shape[1], 3, s)
            return nv.conv_fprop(dout, self._wpk(conv.weight, True), ks=3, cout=conv.cin, stride=1, upsample=(s == 2),
                                 pad=2 - pb, out_hw=(x.shape[1], x.shape[2]), epi_mode=nv.EPI_MASK, mask_src=x,
                                 mask_bn=bn.fwd, extra=extra, out_sums=bn.gsums, out=out)

        def end_bucket(lo, hi_):
            if side:
                p.sync_lanes(1, 0)
            if self._parts is not None:
                self._parts.reduce()
            nv.slot_reduce(a.grads, self._wslots, lo, hi_)
            self.bucket_ranges.append((lo, hi_))
            p.mark_segment()

        x_last = self.y[-1]
        dy = self._d_out[-1]
        nv.head_bwd(x_last, self.bnf.fwd, a.param("fc/fc_weights"), self.feat, self.dlogits, dy, self.bnf.gsums,
                    a.grad("fc/fc_weights"), a.grad("fc/fc_bias"))
        nv.bn_bwd_apply(dy, x_last, self.bnf.fwd, self.bnf.gsums, dgamma=a.grad("fc/gamma"), dbeta=a.grad("fc/beta"), out=dy)
        # A block's last step, d(block input) = BN1-backward(g1) + shortcut gradient of d(block output), is
        # folded into the convolution that consumes it (conv2 of the block before) for the channel
        # counts in DTF_BWD_SHORTCUT_FUSE: that kernel stages the sum from its three operands and
        # writes it out for the shortcut of the block before. Measured at batch 1024, per block:
        # 32 channels 110 -> 102 us, 16 channels 135 -> 112 us (over 4-row bands; the 8-row instance
        # spills), 64 channels 106 -> 103 us. Not across a bucket boundary: BN1's dgamma / dbeta
        # would be written after their bucket has left.
        fuse_c = {int(v) for v in os.environ.get("DTF_BWD_SHORTCUT_FUSE", "16,32,64").split(",") if v.strip()}

        def folds_into_previous(k):
            if k == 0 or k in marks or not self.fused_backward:
                return False
            blk, prev = blocks[k], blocks[k - 1]
            if blk.first_block or blk.conv1.cin not in fuse_c:
                return False
            if blk.increase_dim:   # option-A shortcut: conv1 is the stride-2 layer, its g1 comes from the separate kernels
                return fusable(self.c1[k - 1], prev.conv2)
            return fusable(self.y[k - 1], blk.conv1) and fusable(self.c1[k - 1], prev.conv2)

        # The first block's g1 feeds nothing but the stem's filter gradient, whose kernel applies
        # BN0's backward in its staging prologue (-35 us of elementwise pass at batch 1024).
        stem_fold = self.fused_backward and os.environ.get("DTF_BWD_STEM_FOLD", "1") != "0"
        pending = None   # (raw g1 of the block after, its input, its BN1, its output gradient, option-A shortcut?)
        for k in range(len(blocks) - 1, -1, -1):
            blk = blocks[k]
            x_in = self.c0 if k == 0 else self.y[k - 1]
            bn1 = self.bn0 if blk.first_block else self.bn1[k]
            bn2 = self.bn2[k]
            c1 = self.c1[k]
            dy = self._d_out[k]
            # conv2: filter gradient + masked data gradient, then BN backward
            if pending is not None:
                g_raw, nx_in, nbn1, ndy, opt_a = pending
                g2 = layer_backward(c1, g_raw, blk.conv2, bn2, self._d_c1[k], dout_bn=(nx_in, nbn1), shortcut=(ndy, dy, opt_a))
                pending = None
            else:
                g2 = layer_backward(c1, dy, blk.conv2, bn2, self._d_c1[k])
            fold = folds_into_previous(k)
            if fold:
                key = tuple(x_in.shape)
                if key not in self._g_raw:
                    self._g_raw[key] = torch.empty_like(x_in)
                g1_out = self._g_raw[key]
            else:
                g1_out = self._d_c0 if k == 0 else self._d_out[k - 1]
            if fusable(x_in, blk.conv1):
                # conv1's fused backward applies BN2's backward while it stages g2: dc1 is never written
                g1 = layer_backward(x_in, g2, blk.conv1, bn1, g1_out, extra=dy if blk.first_block else None,
                                    dout_bn=(c1, bn2))
            else:
                dc1 = nv.bn_bwd_apply(g2, c1, bn2.fwd, bn2.gsums, dgamma=a.grad(bn2.gamma_name), dbeta=a.grad(bn2.beta_name),
                                      out=g2)
                g1 = layer_backward(x_in, dc1, blk.conv1, bn1, g1_out, extra=dy if blk.first_block else None)
            if fold:
                pending = (g1, x_in, bn1, dy, blk.increase_dim)
            elif k == 0 and blk.first_block and stem_fold:
                pass   # BN0's backward is applied by the stem's filter-gradient kernel while it stages g1
            else:
                if blk.first_block:
                    res = dict(res_mode=nv.BWD_RES_NONE)
                elif blk.increase_dim:
                    res = dict(res_mode=nv.BWD_RES_SHORTCUT_A, res=dy)
                else:
                    res = dict(res_mode=nv.BWD_RES_IDENT, res=dy)
                nv.bn_bwd_apply(g1, x_in, bn1.fwd, bn1.gsums, dgamma=a.grad(bn1.gamma_name), dbeta=a.grad(bn1.beta_name),
                                out=g1, **res)
            if k in marks:
                end_bucket(marks[k], hi)
                hi = marks[k]
        # stem: filter gradient of conv0 from the raw images
        off0 = a.offsets["conv0/conv"]
        kw = {}
        if stem_fold and blocks[0].first_block:
            kw = dict(dy_bn=(self.c0, self.bn0.fwd, self.bn0.gsums), dy_dgamma=a.grad(self.bn0.gamma_name),
                      dy_dbeta=a.grad(self.bn0.beta_name))
        nv.conv_stem_wgrad(self.images, self._d_c0, self._wslots[0, off0:off0 + 432].view(3, 3, 3, 16), slot_stride=a.total,
                           parts=self._parts, **kw)
        end_bucket(0, hi)

    def _record(self, num_buckets: int) -> None:
        p = self.program
        p.begin_record()
        try:
            p.mark_phase("prologue")
            self._record_prologue([self.stats, self._tile_ctr] if self.training and self.dynamic_tiles else [self.stats])
            p.mark_phase("forward")
            self._record_forward()
            if self.training:
                p.mark_phase("backward")
                self._record_backward(num_buckets)
                p.mark_phase("optimizer")
                self._record_optimizer()
        finally:
            p.end_record()
        self.num_segments = p.num_segments
