"""Tensor-level wrappers around the gfx950 HIP extension (`_hip`).

The extension takes raw device addresses; this module checks shapes/dtypes/contiguity on the
host before every launch (a kernel that faults can take the whole GPU host down), allocates
outputs with torch, and passes the current torch stream so launches order with everything else
torch does. When a GPU is present the extension is mandatory: failing to import it raises
instead of silently falling back to PyTorch ops.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional

import numpy as np
import torch

from .reference import BN_EPSILON, same_pad

try:  # the .so is built in-tree by build_native.py
    from .. import _hip  # type: ignore
except ImportError as exc:  # pragma: no cover - exercised only on broken installs
    _hip = None
    _import_error = exc
else:
    _import_error = None

IN_RAW, IN_BNRELU = 0, 1
RES_NONE, RES_IDENT, RES_BNRELU, RES_SHORTCUT_A = 0, 1, 2, 3
EPI_PLAIN, EPI_BIAS_RELU, EPI_MASK, EPI_BIAS = 0, 1, 2, 3
BWD_RES_NONE, BWD_RES_IDENT, BWD_RES_SHORTCUT_A = 0, 1, 2


def available() -> bool:
    return _hip is not None and torch.cuda.is_available()


def require() -> None:
    """Raises unless the HIP extension is importable and a GPU is visible."""
    if _hip is None:
        raise RuntimeError(
            "distributed_tensorflow_amd._hip is not built (run `python build_native.py`): %r" % (_import_error,)
        )
    if not torch.cuda.is_available():
        raise RuntimeError("no GPU visible to torch; the HIP ops need an MI355X")


def hip():
    if _hip is None:
        raise RuntimeError("distributed_tensorflow_amd._hip is not built: %r" % (_import_error,))
    return _hip


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _chk(t: torch.Tensor, name: str, dtype=torch.float32) -> int:
    if not t.is_cuda:
        raise ValueError(f"{name}: expected a GPU tensor")
    if t.dtype != dtype:
        raise ValueError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: expected a contiguous tensor")
    if t.data_ptr() % 16 != 0 and t.numel() > 0:
        raise ValueError(f"{name}: storage must be 16-byte aligned")
    return t.data_ptr()


# A statistics block is STAT_SLOTS partial blocks of 2*C doubles; kernels add into the slot of
# their workgroup's XCD and consumers add the slots up (see csrc/kernels/common.h).
STAT_SLOTS = 8


def new_stats(c: int, device="cuda") -> torch.Tensor:
    """Zeroed statistics block for C channels."""
    return torch.zeros(STAT_SLOTS * 2 * c, device=device, dtype=torch.float64)


def stats_from_totals(totals: torch.Tensor) -> torch.Tensor:
    """Statistics block whose slots add up to `totals` ([2*C] float64)."""
    out = torch.zeros(STAT_SLOTS * totals.numel(), device=totals.device, dtype=torch.float64)
    out[: totals.numel()] = totals
    return out


def stats_total(sums: torch.Tensor) -> torch.Tensor:
    """[2*C] totals of a statistics block."""
    return sums.view(STAT_SLOTS, -1).sum(0)


@dataclass
class BnState:
    """Batch statistics (fp64 sum / sum of squares) plus affine parameters of one BN layer."""

    sums: torch.Tensor   # float64 [STAT_SLOTS * 2*C]
    gamma: torch.Tensor  # float32 [C]
    beta: torch.Tensor   # float32 [C]
    count: int           # B*H*W the statistics are taken over
    eps: float = BN_EPSILON

    def ref(self):
        c = self.gamma.numel()
        if self.sums.dtype != torch.float64 or self.sums.numel() != STAT_SLOTS * 2 * c:
            raise ValueError("BnState.sums must be float64 [STAT_SLOTS*2*C]")
        if self.beta.numel() != c:
            raise ValueError("BnState gamma/beta size mismatch")
        for t in (self.sums, self.gamma, self.beta):
            if not t.is_cuda or not t.is_contiguous():
                raise ValueError("BnState tensors must be contiguous GPU tensors")
        return (self.sums.data_ptr(), self.gamma.data_ptr(), self.beta.data_ptr(), 1.0 / float(self.count), float(self.eps))

    @property
    def channels(self) -> int:
        return self.gamma.numel()


def pad_cin(c: int) -> int:
    for p in (4, 16, 32, 64):
        if c <= p:
            return p
    raise ValueError("convolutions with more than 64 input channels are not supported")


def packed_floats(ks: int, cin: int, cout: int, flip: bool) -> int:
    return hip().packed_weight_floats(ks, cin, cout, int(flip))


def pack_weights(w: torch.Tensor, flip: bool = False, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """HWIO filter -> MFMA B-operand order; flip=True packs the data-gradient filter."""
    require()
    ks, ks2, cin, cout = w.shape
    if ks != ks2:
        raise ValueError("square filters only")
    n = packed_floats(ks, cin, cout, flip)
    if out is None:
        out = torch.empty(n, device=w.device, dtype=torch.float32)
    if out.numel() != n:
        raise ValueError("pack_weights: wrong output size")
    hip().pack_conv_weights(_chk(w, "w"), _chk(out, "out"), ks, cin, cout, int(flip), stream())
    return out


def split_supported(cin: int, cout: int, stride: int = 1) -> bool:
    """Shapes the bf16x6 kernel covers (see pack_split_weights)."""
    return stride == 1 and cin == cout and cin in (16, 32, 64)


def pack_split_weights(w: torch.Tensor, flip: bool = False, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """3x3 HWIO filter (C -> C channels, C = 16, 32 or 64) -> operand of the opt-in "bf16x6" kernel
    (csrc/kernels/conv_fprop_split.hip): every fp32 value cut exactly into three bf16 values, in
    MFMA fragment order. The tensor is float32-typed storage holding bf16 pairs. flip=True packs
    the data-gradient filter."""
    require()
    ks, ks2, cin, cout = w.shape
    if ks != 3 or ks2 != 3 or not split_supported(cin, cout):
        raise ValueError("pack_split_weights: 3x3 filters with C -> C channels, C in 16, 32, 64")
    n = int(hip().packed_split_floats(cin))
    if out is None:
        out = torch.empty(n, device=w.device, dtype=torch.float32)
    if out.numel() != n:
        raise ValueError("pack_split_weights: wrong output size")
    hip().pack_split_weights(_chk(w, "w"), _chk(out, "out"), cin, int(flip), stream())
    return out


def conv_fprop(
    x: torch.Tensor,
    wpk: torch.Tensor,
    *,
    ks: int,
    cout: int,
    stride: int = 1,
    upsample: bool = False,
    pad: Optional[int] = None,
    out_hw: Optional[tuple[int, int]] = None,
    in_bn: Optional[BnState] = None,
    res_mode: int = RES_NONE,
    res: Optional[torch.Tensor] = None,
    res_bn: Optional[BnState] = None,
    epi_mode: int = EPI_PLAIN,
    bias: Optional[torch.Tensor] = None,
    mask_src: Optional[torch.Tensor] = None,
    mask_bn: Optional[BnState] = None,
    extra: Optional[torch.Tensor] = None,
    out_sums: Optional[torch.Tensor] = None,
    out: Optional[torch.Tensor] = None,
    debug_skip: int = 0,
    dbg_stamps: Optional[torch.Tensor] = None,
    split: bool = False,
) -> torch.Tensor:
    """Forward-shaped convolution with fused prologue/epilogues (see csrc/kernels/conv.h).
    split=True: `wpk` comes from pack_split_weights and the product runs on the bf16 matrix pipe with
    three-way split operands (opt-in, stride-1 16- and 32-channel layers)."""
    require()
    b, h, w_, cin = x.shape
    if upsample:
        if stride != 1:
            raise ValueError("upsample implies an output stride of 1")
        if out_hw is None:
            out_hw = (2 * h, 2 * w_)
        if pad is None:
            raise ValueError("upsample needs an explicit pad")
    if out_hw is None:
        ho, pb, _ = same_pad(h, ks, stride)
        wo, pbw, _ = same_pad(w_, ks, stride)
        if pad is None:
            if pb != pbw:
                raise ValueError("non-square SAME padding is not supported")
            pad = pb
    else:
        ho, wo = out_hw
        if pad is None:
            raise ValueError("out_hw needs an explicit pad")
    # packed filter must match (ks, cin, cout)
    cp = pad_cin(cin)
    kg = ks * ks * (cp // 16) if cp >= 16 else (ks * ks + 3) // 4
    nt_total = (cout + 15) // 16
    if split:
        if ks != 3 or not split_supported(cin, cout, stride) or upsample or wpk.numel() != int(hip().packed_split_floats(cin)):
            raise ValueError("conv_fprop(split=True): 3x3 stride-1 C -> C layers (C = 16, 32, 64) and a pack_split_weights filter")
    elif wpk.numel() != kg * nt_total * 256:
        raise ValueError(f"packed filter has {wpk.numel()} floats, expected {kg * nt_total * 256}")
    if out is None:
        out = torch.empty((b, ho, wo, cout), device=x.device, dtype=torch.float32)
    elif tuple(out.shape) != (b, ho, wo, cout):
        raise ValueError("conv_fprop: out has the wrong shape")
    d = dict(
        B=b, H=h, W=w_, Cin=cin, Ho=ho, Wo=wo, Cout=cout, pad=pad, stride=stride, upsample=int(upsample),
        nt_total=nt_total, in_mode=IN_RAW, res_mode=res_mode, epi_mode=epi_mode, debug_skip=debug_skip,
    )
    d["in"] = _chk(x, "x")
    d["wpk"] = _chk(wpk, "wpk")
    d["out"] = _chk(out, "out")
    if in_bn is not None:
        if in_bn.channels != cin:
            raise ValueError("in_bn channel mismatch")
        d["in_mode"] = IN_BNRELU
        d["in_bn"] = in_bn.ref()
    if res_mode in (RES_IDENT, RES_BNRELU):
        if res is None or tuple(res.shape) != (b, ho, wo, cout):
            raise ValueError("residual must match the output shape")
        d["res"] = _chk(res, "res")
        if res_mode == RES_BNRELU:
            if res_bn is None or res_bn.channels != cout:
                raise ValueError("res_bn missing or channel mismatch")
            d["res_bn"] = res_bn.ref()
    elif res_mode == RES_SHORTCUT_A:
        if res is None or tuple(res.shape) != (b, 2 * ho, 2 * wo, cout // 2) or cout % 4:
            raise ValueError("option-A shortcut source must be [B,2Ho,2Wo,Cout/2]")
        d["res"] = _chk(res, "res")
    if epi_mode in (EPI_BIAS_RELU, EPI_BIAS):
        if bias is None or bias.numel() != cout:
            raise ValueError("bias missing or wrong size")
        d["bias"] = _chk(bias, "bias")
    elif epi_mode == EPI_MASK:
        if mask_src is None or tuple(mask_src.shape) != (b, ho, wo, cout) or mask_bn is None or mask_bn.channels != cout:
            raise ValueError("EPI_MASK needs mask_src of the output shape and mask_bn")
        d["mask_src"] = _chk(mask_src, "mask_src")
        d["mask_bn"] = mask_bn.ref()
        if extra is not None:
            if tuple(extra.shape) != (b, ho, wo, cout):
                raise ValueError("extra must match the output shape")
            d["extra"] = _chk(extra, "extra")
    if out_sums is not None:
        if out_sums.numel() != STAT_SLOTS * 2 * cout:
            raise ValueError("out_sums must be [STAT_SLOTS*2*Cout]")
        d["out_sums"] = _chk(out_sums, "out_sums", torch.float64)
    if dbg_stamps is not None:
        d["dbg_stamps"] = _chk(dbg_stamps, "dbg_stamps", torch.int64)
    if split:
        if not hip().conv_fprop_split(d, stream()):
            raise ValueError("conv_fprop(split=True): this combination of shape and fused operations is not covered")
        return out
    hip().conv_fprop(d, ks, stream())
    return out


class GradientParts:
    """Filter gradients without atomics. The filter-gradient kernels normally add each workgroup's
    sum to one of STAT_SLOTS gradient copies with float atomics, in whatever order the workgroups
    finish, so the last bits of a gradient differ from run to run. Handed one of these (`parts=`),
    a kernel instead stores every workgroup's sum into a copy of its own, and `reduce()` - one launch
    for all tensors registered since the last call - adds the copies up in a fixed order into the
    slots (`slot_reduce` then adds the slots, also in a fixed order). Same state in, same bits out.
    `slots` is the [STAT_SLOTS, n] buffer the gradients' `dw` views live in."""

    ENTRY = np.dtype([("part", "<u8"), ("dst_off", "<i8"), ("copies", "<i4"), ("numel4", "<i4"), ("first_item", "<i4"), ("pad", "<i4")])

    def __init__(self, slots: torch.Tensor):
        if slots.dim() != 2 or slots.shape[0] != STAT_SLOTS or slots.dtype != torch.float32:
            raise ValueError("GradientParts: slots must be float32 [STAT_SLOTS, n]")
        self.slots = slots
        self._pending = []
        self._keep = []          # buffers and tables the recorded launches point into
        self.bytes = 0

    def add(self, dw: torch.Tensor, copies: int) -> int:
        """Registers tensor `dw` (a view into slot 0) with `copies` workgroup copies; returns the
        device address of the copies' buffer."""
        numel = dw.numel()
        if copies < 1 or numel % 4 or dw.storage_offset() % 4 or dw.untyped_storage().data_ptr() != self.slots.untyped_storage().data_ptr():
            raise ValueError("GradientParts.add: dw must be a 16-byte aligned view into the slots")
        buf = torch.empty(copies * numel, device=dw.device, dtype=torch.float32)
        self._keep.append(buf)
        self.bytes += buf.numel() * 4
        self._pending.append((buf.data_ptr(), dw.storage_offset() - self.slots.storage_offset(), copies, numel // 4))
        return buf.data_ptr()

    def reduce(self) -> None:
        if not self._pending:
            return
        require()
        table = np.zeros(len(self._pending), dtype=self.ENTRY)
        items = 0
        for i, (ptr, off, copies, numel4) in enumerate(self._pending):
            table[i] = (ptr, off, copies, numel4, items, 0)
            items += (numel4 + 255) // 256 * STAT_SLOTS
        assert self.ENTRY.itemsize == hip().PART_ENTRY_BYTES
        t = torch.from_numpy(table.view(np.uint8).copy()).to(self.slots.device)
        self._keep.append(t)
        hip().part_reduce(t.data_ptr(), len(self._pending), items, self.slots.data_ptr(), self.slots.shape[1], stream())
        self._pending = []


def conv_wgrad(
    x: torch.Tensor,
    dout: torch.Tensor,
    dw: torch.Tensor,
    *,
    stride: int = 1,
    in_bn: Optional[BnState] = None,
    tiles_per_block: int = 0,
    slot_stride: int = 0,
    co_tiles: int = 0,
    debug_skip: int = 0,
    dbg_stamps: Optional[torch.Tensor] = None,
    parts: Optional[GradientParts] = None,
) -> torch.Tensor:
    """dw (HWIO, pre-zeroed or holding a running sum) += filter gradient.

    slot_stride > 0: `dw` is copy 0 of STAT_SLOTS copies spaced slot_stride floats apart in the
    same storage; each workgroup adds into the copy of its XCD and the caller sums the copies."""
    require()
    b, h, w_, cin = x.shape
    ks, _, cin2, cout = dw.shape
    b2, ho, wo, cout2 = dout.shape
    eho, pb, _ = same_pad(h, ks, stride)
    ewo, _, _ = same_pad(w_, ks, stride)
    if (b2, ho, wo, cout2, cin2) != (b, eho, ewo, cout, cin):
        raise ValueError("conv_wgrad: shape mismatch")
    d = dict(B=b, H=h, W=w_, Cin=cin, Ho=ho, Wo=wo, Cout=cout, pad=pb, stride=stride, in_mode=IN_RAW,
             tiles_per_block=tiles_per_block, dw_slot_stride=slot_stride, co_tiles=co_tiles, debug_skip=debug_skip)
    if slot_stride:
        last = (dw.storage_offset() + (STAT_SLOTS - 1) * slot_stride + dw.numel()) * 4
        if slot_stride < dw.numel() or last > dw.untyped_storage().nbytes():
            raise ValueError("conv_wgrad: the slot copies do not fit dw's storage")
    d["in"] = _chk(x, "x")
    d["dout"] = _chk(dout, "dout")
    d["dw"] = _chk(dw, "dw")
    if in_bn is not None:
        if in_bn.channels != cin:
            raise ValueError("in_bn channel mismatch")
        d["in_mode"] = IN_BNRELU
        d["in_bn"] = in_bn.ref()
    if dbg_stamps is not None:
        d["dbg_stamps"] = _chk(dbg_stamps, "dbg_stamps", torch.int64)
    if parts is not None:
        # only the band kernel has the atomic-free flush; other shapes keep the atomics
        copies = int(hip().conv_wgrad_band_copies(d, ks))
        if copies > 0:
            d["dw_part"] = parts.add(dw, copies)
    hip().conv_wgrad(d, ks, stream())
    return dw


def conv_stem_fwd(x: torch.Tensor, w: torch.Tensor, out: Optional[torch.Tensor] = None, out_sums: Optional[torch.Tensor] = None):
    """3x3 SAME convolution 3 -> 16 channels over [B,32,32,3] images with the plain [3,3,3,16]
    filter; out_sums (a statistics block for 16 channels) receives the output's per-channel sums."""
    require()
    b = int(x.shape[0])
    if tuple(x.shape[1:]) != (32, 32, 3) or tuple(w.shape) != (3, 3, 3, 16):
        raise ValueError("conv_stem_fwd: expected [B,32,32,3] images and a [3,3,3,16] filter")
    if out is None:
        out = torch.empty((b, 32, 32, 16), device=x.device, dtype=torch.float32)
    elif tuple(out.shape) != (b, 32, 32, 16):
        raise ValueError("conv_stem_fwd: out has the wrong shape")
    d = dict(x=_chk(x, "x"), w=_chk(w, "w"), y=_chk(out, "y"), B=b)
    if out_sums is not None:
        if out_sums.numel() != STAT_SLOTS * 32:
            raise ValueError("conv_stem_fwd: out_sums must be a statistics block of 16 channels")
        d["sums"] = _chk(out_sums, "sums", torch.float64)
    hip().conv_stem_fwd(d, stream())
    return out


def conv_stem_wgrad(x: torch.Tensor, dy: torch.Tensor, dw: torch.Tensor, slot_stride: int = 0,
                    dy_bn=None, dy_dgamma=None, dy_dbeta=None, parts: Optional[GradientParts] = None) -> None:
    """dw [3,3,3,16] += filter gradient of conv_stem_fwd (per-XCD copies with slot_stride).
    dy_bn = (y, BnState of y, backward sums): `dy` is the masked gradient at the output of the
    BatchNorm after the stem and that BatchNorm's backward is applied while dy is staged (as in
    conv_bwd_fused); dy_dgamma / dy_dbeta receive its parameter gradients."""
    require()
    b = int(x.shape[0])
    if tuple(x.shape[1:]) != (32, 32, 3) or tuple(dy.shape) != (b, 32, 32, 16) or dw.numel() != 27 * 16:
        raise ValueError("conv_stem_wgrad: expected x [B,32,32,3], dy [B,32,32,16], dw [3,3,3,16]")
    if slot_stride:
        last = (dw.storage_offset() + (STAT_SLOTS - 1) * slot_stride + dw.numel()) * 4
        if slot_stride < dw.numel() or last > dw.untyped_storage().nbytes():
            raise ValueError("conv_stem_wgrad: the slot copies do not fit dw's storage")
    d = dict(x=_chk(x, "x"), dy=_chk(dy, "dy"), dw=_chk(dw, "dw"), slot_stride=slot_stride, B=b)
    if dy_bn is not None:
        y, ybn, ygs = dy_bn
        if tuple(y.shape) != (b, 32, 32, 16) or ybn.channels != 16 or ygs.numel() != STAT_SLOTS * 32:
            raise ValueError("conv_stem_wgrad: dy_bn operands have the wrong shape")
        d.update(dy_x=_chk(y, "dy_x"), dy_bn=ybn.ref(), dy_gsums=_chk(ygs, "dy_gsums", torch.float64))
        for key, t in (("dy_dgamma", dy_dgamma), ("dy_dbeta", dy_dbeta)):
            if t is not None:
                if t.numel() != 16:
                    raise ValueError(f"{key} must have 16 elements")
                d[key] = _chk(t, key)
    if parts is not None:
        d["dw_part"] = parts.add(dw, int(hip().conv_stem_wgrad_copies(d)))
    hip().conv_stem_wgrad(d, stream())


def conv5x5_tile_supported(h: int, w_: int, cin: int, cout: int) -> bool:
    return bool(hip().conv5x5_tile_supported(h, w_, cin, cout))


def conv5x5_pack(w: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """HWIO 5x5 filter [5,5,Cin,Cout] -> [25][Cout][Cin], the forward operand of conv5x5_tile."""
    require()
    if w.dim() != 4 or w.shape[0] != 5 or w.shape[1] != 5:
        raise ValueError("conv5x5_pack: expected a [5,5,Cin,Cout] filter")
    cin, cout = int(w.shape[2]), int(w.shape[3])
    if out is None:
        out = torch.empty(25 * cin * cout, device=w.device, dtype=torch.float32)
    elif out.numel() != 25 * cin * cout:
        raise ValueError("conv5x5_pack: out has the wrong size")
    hip().conv5x5_pack(dict(w=_chk(w, "w"), out=_chk(out, "out"), Cin=cin, Cout=cout), stream())
    return out


def conv5x5_split_pack(w: torch.Tensor, flip: bool = False, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """HWIO 5x5 filter [5,5,32,64] -> operand of conv5x5_tile(..., split=True): every fp32 value cut into
    three bf16 values, in MFMA fragment order (csrc/kernels/conv5x5_split.hip). flip=True packs the
    data-gradient operand. float32-typed storage holding bf16 pairs."""
    require()
    if tuple(w.shape) != (5, 5, 32, 64):
        raise ValueError("conv5x5_split_pack: expected the [5,5,32,64] filter")
    n = int(hip().conv5x5_split_packed_floats())
    if out is None:
        out = torch.empty(n, device=w.device, dtype=torch.float32)
    elif out.numel() != n:
        raise ValueError("conv5x5_split_pack: out has the wrong size")
    hip().conv5x5_split_pack(dict(w=_chk(w, "w"), out=_chk(out, "out"), Cin=32, Cout=64, flip=int(flip)), stream())
    return out


def _pooled_operands(pooled, b: int, who: str) -> dict:
    dp, arg, pm = pooled
    shp = (b, 7, 7, 64)
    if tuple(dp.shape) != shp or tuple(arg.shape) != shp or tuple(pm.shape) != shp:
        raise ValueError(f"{who}: pooled = (dp, arg, p), each [B,7,7,64]")
    return dict(pool_dp=_chk(dp, "pool_dp"), pool_arg=_chk(arg, "pool_arg", torch.uint8), pool_p=_chk(pm, "pool_p"))


def conv5x5_tile(x: Optional[torch.Tensor], w: torch.Tensor, *, bias: Optional[torch.Tensor] = None, flip: bool = False,
                 out: Optional[torch.Tensor] = None, imgs: int = 0, pooled=None, pool_out=None, split: bool = False) -> torch.Tensor:
    """5x5 stride-1 SAME convolution over [B,14,14,C] maps with whole images per workgroup.
    flip=False: x has 32 channels, w = conv5x5_pack(filter), returns relu(conv + bias) with 64
    channels. flip=True: x is the 64-channel output gradient, w = the [5,5,32,64] filter itself,
    returns the 32-channel input gradient (see csrc/kernels/conv.h: Conv5x5Args). With
    pooled=(dp, arg, p) (flip only, x=None) the output gradient is the 2x2 max-pool + ReLU backward
    of dp [B,7,7,64], formed while staging instead of being read from memory. With
    pool_out=(p, arg) (forward form) the kernel writes the 2x2 max pool of the activation
    ([B,7,7,64] maxima, uint8 winner positions) instead of the activation and returns p.
    split=True: the product runs on the bf16 matrix pipe with operands split three ways and fp32
    accumulation; w is then conv5x5_split_pack(filter, flip=flip)."""
    require()
    if pooled is not None:
        if not flip or x is not None:
            raise ValueError("conv5x5_tile: pooled staging replaces x in the flip form")
        b, cin = int(pooled[0].shape[0]), 64
        dev = pooled[0].device
    else:
        if x.dim() != 4 or x.shape[1] != 14 or x.shape[2] != 14:
            raise ValueError("conv5x5_tile: expected [B,14,14,C] input")
        b, cin = int(x.shape[0]), int(x.shape[3])
        dev = x.device
    cout = 32 if flip else 64
    if cin != (64 if flip else 32) or w.numel() != (int(hip().conv5x5_split_packed_floats()) if split else 25 * 32 * 64):
        raise ValueError("conv5x5_tile: channel counts must be 32->64 (forward) or 64->32 (flip)")
    if imgs not in (0, 1, 2):
        raise ValueError("conv5x5_tile: imgs must be 0, 1 or 2")
    d = dict(w=_chk(w, "w"), B=b, Cin=cin, Cout=cout, flip=int(flip), imgs=imgs, split=bool(split))
    if pool_out is not None:
        pm, parg = pool_out
        if flip or tuple(pm.shape) != (b, 7, 7, 64) or tuple(parg.shape) != (b, 7, 7, 64):
            raise ValueError("conv5x5_tile: pool_out = (maxima, winner positions), each [B,7,7,64], forward form only")
        d["pool_out"] = _chk(pm, "pool_out")
        d["pool_arg_out"] = _chk(parg, "pool_arg_out", torch.uint8)
        out = pm
    else:
        if out is None:
            out = torch.empty((b, 14, 14, cout), device=dev, dtype=torch.float32)
        elif tuple(out.shape) != (b, 14, 14, cout):
            raise ValueError("conv5x5_tile: out has the wrong shape")
        d["y"] = _chk(out, "y")
    if pooled is not None:
        d.update(_pooled_operands(pooled, b, "conv5x5_tile"))
    else:
        d["x"] = _chk(x, "x")
    if not flip:
        if bias is None or bias.numel() != cout:
            raise ValueError("conv5x5_tile: the forward form needs a bias of Cout elements")
        d["bias"] = _chk(bias, "bias")
    hip().conv5x5_tile(d, stream())
    return out


def conv5x5_wgrad_parts(batch: int) -> int:
    """Number of per-workgroup partial results conv5x5_wgrad(partials=...) writes for this batch."""
    return int(hip().conv5x5_wgrad_parts(int(batch)))


def conv5x5_wgrad(x: torch.Tensor, dy: Optional[torch.Tensor], dw: torch.Tensor, slot_stride: int = 0,
                  partials: Optional[torch.Tensor] = None, pooled=None, split: bool = False) -> None:
    """dw [5,5,32,64] += filter gradient of the 5x5 SAME convolution over [B,14,14,32] inputs with
    output gradient dy [B,14,14,64]; with slot_stride into the per-XCD copies starting at dw.
    partials [conv5x5_wgrad_parts(B), 51200]: the workgroups store their partial results there
    without atomics and one partial_sum adds them to dw. pooled=(dp, arg, p) with dy=None: the
    output gradient is the max-pool + ReLU backward of dp [B,7,7,64], formed while staging.
    split=True: on the bf16 matrix pipe with both operands split three ways, fp32 accumulation."""
    require()
    b = int(x.shape[0])
    n = 25 * 32 * 64
    if tuple(x.shape) != (b, 14, 14, 32) or dw.numel() != n:
        raise ValueError("conv5x5_wgrad: expected x [B,14,14,32], dw [5,5,32,64]")
    d = dict(x=_chk(x, "x"), dw=_chk(dw, "dw"), slot_stride=slot_stride, B=b, split=bool(split))
    if pooled is not None:
        if dy is not None:
            raise ValueError("conv5x5_wgrad: pass either dy or pooled")
        d.update(_pooled_operands(pooled, b, "conv5x5_wgrad"))
    else:
        if tuple(dy.shape) != (b, 14, 14, 64):
            raise ValueError("conv5x5_wgrad: expected dy [B,14,14,64]")
        d["dy"] = _chk(dy, "dy")
    if partials is not None:
        parts = conv5x5_wgrad_parts(b)
        if partials.dim() != 2 or partials.shape[0] < parts or partials.shape[1] != n:
            raise ValueError(f"conv5x5_wgrad: partials must be [>= {parts}, {n}]")
        d["partials"] = _chk(partials, "partials")
        d["slot_stride"] = 0
        hip().conv5x5_wgrad(d, stream())
        hip().partial_sum(dict(y=_chk(dw, "dw"), ws=_chk(partials, "partials"), n=n, parts=parts, stride=n), stream())
        return
    if slot_stride:
        last = (dw.storage_offset() + (STAT_SLOTS - 1) * slot_stride + dw.numel()) * 4
        if slot_stride < dw.numel() or last > dw.untyped_storage().nbytes():
            raise ValueError("conv5x5_wgrad: the slot copies do not fit dw's storage")
    hip().conv5x5_wgrad(d, stream())


def conv_bwd_fused_supported(x: torch.Tensor) -> bool:
    b, h, w_, c = x.shape
    return bool(hip().conv_bwd_fused_supported(b, h, w_, c))


def conv_bwd_fused(dy, x, bn: BnState, wpk_t, dw, gsums, *, slot_stride: int = 0, extra=None, out=None,
                   dy_bn=None, dy_dgamma=None, dy_dbeta=None, dy_res=None, dy_out=None, dy_res_option_a=False,
                   tile_counters=None, parts: Optional[GradientParts] = None, split: bool = False, dbg_stamps=None):
    """Backward of y = conv3x3(relu(bn(x)), W), stride 1, in one kernel: returns the masked data
    gradient g, adds (sum g, sum g*xhat) to `gsums` and the filter gradient to `dw`
    (see csrc/kernels/conv.h: ConvBwdFusedArgs).

    dy_bn = (y, BnState of y, backward sums of that BatchNorm): `dy` is then the masked gradient
    at the output of the BatchNorm that follows the convolution, and the BatchNorm input gradient
    is formed on the fly; dy_dgamma / dy_dbeta receive that BatchNorm's parameter gradients.
    With dy_res / dy_out (and dy_bn) the gradient of an identity shortcut is added to that
    BatchNorm input gradient and the sum is written to dy_out (a buffer other than dy); with
    dy_res_option_a the shortcut is the resolution-halving one (2x2 average pool + channel padding)
    and dy_res is the [B, H/2, W/2, 2C] gradient at its output.

    split=True: both products run on the bf16 matrix pipe with operands split three ways and fp32
    accumulation (csrc/kernels/conv_bwd_split.hip); wpk_t is then pack_split_weights(w, flip=True)."""
    require()
    b, h, w_, c = x.shape
    if tuple(dy.shape) != (b, h, w_, c) or tuple(dw.shape) != (3, 3, c, c) or bn.channels != c:
        raise ValueError("conv_bwd_fused: shape mismatch")
    if split:
        if not hip().conv_bwd_split_supported(b, h, w_, c) or tile_counters is not None:
            raise ValueError("conv_bwd_fused: the split-operand kernel does not cover this call")
        want_w = int(hip().packed_split_floats(c))
    else:
        want_w = 9 * (c // 16) * (c // 16) * 256
    if wpk_t.numel() != want_w or gsums.numel() != STAT_SLOTS * 2 * c:
        raise ValueError("conv_bwd_fused: packed filter / statistics block have the wrong size")
    if slot_stride:
        last = (dw.storage_offset() + (STAT_SLOTS - 1) * slot_stride + dw.numel()) * 4
        if slot_stride < dw.numel() or last > dw.untyped_storage().nbytes():
            raise ValueError("conv_bwd_fused: the slot copies do not fit dw's storage")
    if out is None:
        out = torch.empty_like(x)
    elif tuple(out.shape) != (b, h, w_, c):
        raise ValueError("conv_bwd_fused: out has the wrong shape")
    d = dict(dy=_chk(dy, "dy"), x=_chk(x, "x"), wpk_t=_chk(wpk_t, "wpk_t"), g=_chk(out, "g"), dw=_chk(dw, "dw"),
             dw_slot_stride=slot_stride, bn=bn.ref(), gsums=_chk(gsums, "gsums", torch.float64), B=b, H=h, W=w_, C=c)
    if extra is not None:
        if tuple(extra.shape) != (b, h, w_, c):
            raise ValueError("extra must match the output shape")
        d["extra"] = _chk(extra, "extra")
    if dy_bn is not None:
        y, ybn, ygs = dy_bn
        if tuple(y.shape) != (b, h, w_, c) or ybn.channels != c or ygs.numel() != STAT_SLOTS * 2 * c:
            raise ValueError("conv_bwd_fused: dy_bn operands have the wrong shape")
        d["dy_x"] = _chk(y, "dy_x")
        d["dy_bn"] = ybn.ref()
        d["dy_gsums"] = _chk(ygs, "dy_gsums", torch.float64)
        for key, t in (("dy_dgamma", dy_dgamma), ("dy_dbeta", dy_dbeta)):
            if t is not None:
                if t.numel() != c:
                    raise ValueError(f"{key} must have C elements")
                d[key] = _chk(t, key)
    if (dy_res is None) != (dy_out is None) or (dy_res is not None and dy_bn is None):
        raise ValueError("conv_bwd_fused: dy_res, dy_out and dy_bn go together")
    if dy_res is not None:
        want_res = (b, h // 2, w_ // 2, 2 * c) if dy_res_option_a else (b, h, w_, c)
        if tuple(dy_res.shape) != want_res or tuple(dy_out.shape) != (b, h, w_, c):
            raise ValueError("conv_bwd_fused: dy_res / dy_out have the wrong shape")
        if dy_res_option_a and c == 64:
            raise ValueError("conv_bwd_fused: no option-A shortcut behind a 64-channel stage")
        if dy_out.data_ptr() in (dy.data_ptr(), dy_res.data_ptr(), x.data_ptr(), out.data_ptr()):
            raise ValueError("conv_bwd_fused: dy_out must not alias an operand (other bands still read their halo rows)")
        d["dy_res"] = _chk(dy_res, "dy_res")
        d["dy_out"] = _chk(dy_out, "dy_out")
        d["dy_res_mode"] = 1 if dy_res_option_a else 0
    if tile_counters is not None:
        # zeroed int32 block: bands are handed out dynamically (robust when CUs are shared with a collective)
        if tile_counters.dtype != torch.int32 or tile_counters.numel() < hip().BWD_TILE_COUNTERS:
            raise ValueError("tile_counters must be BWD_TILE_COUNTERS int32")
        d["tile_counters"] = _chk(tile_counters, "tile_counters", torch.int32)
    if split:
        d["split"] = True
    if dbg_stamps is not None:   # timing experiments only: int64 [blocks][4][8][6], see ConvBwdFusedArgs::dbg_stamps
        d["dbg_stamps"] = _chk(dbg_stamps, "dbg_stamps", torch.int64)
    if parts is not None:
        d["dw_part"] = parts.add(dw, int(hip().conv_bwd_fused_copies(d)))
    hip().conv_bwd_fused(d, stream())
    return out


# ----------------------------------------------------------------------------- batch norm
def bn_stats(x: torch.Tensor, sums: Optional[torch.Tensor] = None) -> torch.Tensor:
    require()
    c = x.shape[-1]
    if sums is None:
        sums = new_stats(c, x.device)
    hip().bn_stats(dict(x=_chk(x, "x"), sums=_chk(sums, "sums", torch.float64), n_rows=x.numel() // c, C=c), stream())
    return sums


def bn_apply(x: torch.Tensor, bn: BnState, relu: bool = True, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    require()
    c = x.shape[-1]
    if bn.channels != c:
        raise ValueError("bn channel mismatch")
    if out is None:
        out = torch.empty_like(x)
    hip().bn_apply(dict(x=_chk(x, "x"), y=_chk(out, "out"), bn=bn.ref(), n_rows=x.numel() // c, C=c, relu=int(relu)), stream())
    return out


def bn_relu_bwd_reduce(dy, x, bn: BnState, gsums, relu=True, out=None):
    require()
    c = x.shape[-1]
    if dy.shape != x.shape or bn.channels != c or gsums.numel() != STAT_SLOTS * 2 * c:
        raise ValueError("bn_relu_bwd_reduce: shape mismatch")
    if out is None:
        out = torch.empty_like(dy)
    hip().bn_relu_bwd_reduce(
        dict(dy=_chk(dy, "dy"), x=_chk(x, "x"), g=_chk(out, "g"), bn=bn.ref(), gsums=_chk(gsums, "gsums", torch.float64),
             n_rows=x.numel() // c, C=c, relu=int(relu)), stream())
    return out


def bn_bwd_apply(g, x, bn: BnState, gsums, *, dgamma=None, dbeta=None, res_mode=BWD_RES_NONE, res=None, out=None):
    require()
    c = x.shape[-1]
    if g.shape != x.shape or bn.channels != c or gsums.numel() != STAT_SLOTS * 2 * c:
        raise ValueError("bn_bwd_apply: shape mismatch")
    if out is None:
        out = torch.empty_like(g)
    d = dict(g=_chk(g, "g"), x=_chk(x, "x"), dx=_chk(out, "dx"), bn=bn.ref(), gsums=_chk(gsums, "gsums", torch.float64),
             n_rows=x.numel() // c, C=c, res_mode=res_mode)
    if dgamma is not None:
        if dgamma.numel() != c or dbeta is None or dbeta.numel() != c:
            raise ValueError("dgamma/dbeta size mismatch")
        d["dgamma"] = _chk(dgamma, "dgamma")
        d["dbeta"] = _chk(dbeta, "dbeta")
    if res_mode == BWD_RES_IDENT:
        if res is None or res.shape != x.shape:
            raise ValueError("identity residual gradient must match x")
        d["res"] = _chk(res, "res")
    elif res_mode == BWD_RES_SHORTCUT_A:
        b, h, w_, _ = x.shape
        if res is None or tuple(res.shape) != (b, h // 2, w_ // 2, 2 * c) or h % 2 or w_ % 2:
            raise ValueError("shortcut-A residual gradient must be [B,H/2,W/2,2C]")
        d["res"] = _chk(res, "res")
        d["H"], d["W"] = h, w_
    hip().bn_bwd_apply(d, stream())
    return out


# ----------------------------------------------------------------------------- head / loss
def head_fwd(x, bn: BnState, fc_w, fc_b, labels, feat, logits, dlogits, loss, correct=None):
    """`loss` (and `correct`) may have a power-of-two number of elements: image n accumulates into
    element n % len, the result is their sum (one shared element serialises the blocks' atomics)."""
    require()
    b, h, w_, c = x.shape
    k = fc_w.shape[1]
    if tuple(fc_w.shape) != (c, k) or fc_b.numel() != k or labels.numel() != b or bn.channels != c:
        raise ValueError("head_fwd: shape mismatch")
    if tuple(feat.shape) != (b, c) or tuple(logits.shape) != (b, k) or tuple(dlogits.shape) != (b, k):
        raise ValueError("head_fwd: output shape mismatch")
    d = dict(x=_chk(x, "x"), bn=bn.ref(), fc_w=_chk(fc_w, "fc_w"), fc_b=_chk(fc_b, "fc_b"),
             labels=_chk(labels, "labels", torch.int32), feat=_chk(feat, "feat"), logits=_chk(logits, "logits"),
             dlogits=_chk(dlogits, "dlogits"), loss=_chk(loss, "loss"), B=b, HW=h * w_, C=c, K=k, slots=loss.numel())
    if correct is not None:
        if correct.numel() != loss.numel():
            raise ValueError("head_fwd: loss and correct need the same number of slots")
        d["correct"] = _chk(correct, "correct", torch.int32)
    hip().head_fwd(d, stream())


def head_bwd(x, bn: BnState, fc_w, feat, dlogits, g, gsums, d_fc_w, d_fc_b):
    require()
    b, h, w_, c = x.shape
    k = fc_w.shape[1]
    if g.shape != x.shape or tuple(d_fc_w.shape) != (c, k) or d_fc_b.numel() != k or gsums.numel() != STAT_SLOTS * 2 * c:
        raise ValueError("head_bwd: shape mismatch")
    if tuple(feat.shape) != (b, c) or tuple(dlogits.shape) != (b, k):
        raise ValueError("head_bwd: saved tensor shape mismatch")
    hip().head_bwd(
        dict(x=_chk(x, "x"), bn=bn.ref(), fc_w=_chk(fc_w, "fc_w"), feat=_chk(feat, "feat"), dlogits=_chk(dlogits, "dlogits"),
             g=_chk(g, "g"), gsums=_chk(gsums, "gsums", torch.float64), d_fc_w=_chk(d_fc_w, "d_fc_w"),
             d_fc_b=_chk(d_fc_b, "d_fc_b"), B=b, HW=h * w_, C=c, K=k), stream())


def softmax_xent(logits, labels, loss, dlogits=None, correct=None):
    require()
    b, k = logits.shape
    if labels.numel() != b:
        raise ValueError("softmax_xent: label count mismatch")
    d = dict(logits=_chk(logits, "logits"), labels=_chk(labels, "labels", torch.int32), loss=_chk(loss, "loss"), B=b, K=k)
    if dlogits is not None:
        if dlogits.shape != logits.shape:
            raise ValueError("dlogits shape mismatch")
        d["dlogits"] = _chk(dlogits, "dlogits")
    if correct is not None:
        d["correct"] = _chk(correct, "correct", torch.int32)
    hip().softmax_xent(d, stream())


def top1_count(logits, labels, correct):
    require()
    b, k = logits.shape
    hip().top1_count(dict(logits=_chk(logits, "logits"), labels=_chk(labels, "labels", torch.int32),
                          correct=_chk(correct, "correct", torch.int32), B=b, K=k), stream())


# ----------------------------------------------------------------------------- optimizers / flat
HYPER_SLOTS = 8  # lr_t, beta1, beta2, eps, grad_scale, momentum, weight_decay, spare


def set_scalars(f_dst: Optional[torch.Tensor] = None, f_values=(), u_dst: Optional[torch.Tensor] = None, u_values=()) -> None:
    """Stream-ordered write of a few scalars that are passed by value with the launch (no host
    staging buffer): f_dst[:len(f_values)] = f_values (float32), u_dst[:len(u_values)] = u_values
    (64-bit integers)."""
    fp = up = 0
    if f_dst is not None:
        if f_dst.dtype != torch.float32 or not f_dst.is_cuda or f_dst.numel() < len(f_values):
            raise ValueError("set_scalars: f_dst must be a float32 GPU tensor holding all values")
        fp = f_dst.data_ptr()
    if u_dst is not None:
        if u_dst.dtype != torch.int64 or not u_dst.is_cuda or u_dst.numel() < len(u_values):
            raise ValueError("set_scalars: u_dst must be an int64 GPU tensor holding all values")
        up = u_dst.data_ptr()
    _hip.set_scalars(fp, [float(x) for x in f_values] if fp else [], up, [int(x) & 0xFFFFFFFFFFFFFFFF for x in u_values] if up else [],
                     stream())


def fused_adam(p, g, m, v, hyper):
    require()
    n = p.numel()
    if g.numel() != n or m.numel() != n or v.numel() != n or hyper.numel() < HYPER_SLOTS:
        raise ValueError("fused_adam: size mismatch")
    hip().fused_adam(dict(p=_chk(p, "p"), g=_chk(g, "g"), m=_chk(m, "m"), v=_chk(v, "v"), hyper=_chk(hyper, "hyper"), n=n), stream())


def fused_momentum_sgd(p, g, m, hyper):
    require()
    n = p.numel()
    if g.numel() != n or m.numel() != n or hyper.numel() < HYPER_SLOTS:
        raise ValueError("fused_momentum_sgd: size mismatch")
    hip().fused_momentum_sgd(dict(p=_chk(p, "p"), g=_chk(g, "g"), m=_chk(m, "m"), hyper=_chk(hyper, "hyper"), n=n), stream())


def flat_scale(x, scale: float = 1.0, scale_dev: Optional[torch.Tensor] = None):
    require()
    d = dict(x=_chk(x, "x"), scale=float(scale), n=x.numel())
    if scale_dev is not None:
        d["scale_dev"] = _chk(scale_dev, "scale_dev")
    hip().flat_scale(d, stream())
    return x


def flat_axpy(y, x, alpha: float):
    require()
    if x.numel() != y.numel():
        raise ValueError("flat_axpy: size mismatch")
    hip().flat_axpy(dict(y=_chk(y, "y"), x=_chk(x, "x"), alpha=float(alpha), n=y.numel()), stream())
    return y


def slot_reduce(y: torch.Tensor, slots: torch.Tensor, lo: int, hi: int):
    """y[lo:hi] += sum_r slots[r, lo:hi]; clears the copies (slots is [STAT_SLOTS, len(y)])."""
    require()
    if slots.dim() != 2 or slots.shape[0] != STAT_SLOTS or slots.shape[1] < hi or y.numel() < hi or lo % 4:
        raise ValueError("slot_reduce: bad shapes")
    if hi > lo:
        hip().slot_reduce(dict(y=_chk(y, "y") + 4 * lo, slots=_chk(slots, "slots") + 4 * lo, n=hi - lo,
                               slot_stride=slots.shape[1]), stream())
    return y


def memset_zero(t: torch.Tensor):
    require()
    hip().memset(t.data_ptr(), 0, t.numel() * t.element_size(), stream())


ZERO_MULTI_MAX = 8


def zero_multi(tensors) -> None:
    """Zeroes several buffers in one launch (at most 8; each 16-byte aligned, a multiple of 16 bytes)."""
    require()
    bufs = []
    for t in tensors:
        if not (t.is_cuda and t.is_contiguous()):
            raise ValueError("zero_multi: contiguous device tensors only")
        bufs.append((t.data_ptr(), t.numel() * t.element_size()))
    hip().zero_multi(bufs, stream())


# ----------------------------------------------------------------------------- elementwise & friends
def add_relu_fwd(a, b, relu_first: bool, out=None):
    require()
    if a.shape != b.shape:
        raise ValueError("add_relu: shape mismatch")
    if out is None:
        out = torch.empty_like(a)
    hip().add_relu_fwd(dict(a=_chk(a, "a"), b=_chk(b, "b"), out=_chk(out, "out"), n=a.numel(), relu_first=int(relu_first)), stream())
    return out


def add_relu_bwd(dout, a, b, relu_first: bool):
    require()
    da, db = torch.empty_like(a), torch.empty_like(b)
    hip().add_relu_bwd(dict(dout=_chk(dout, "dout"), a=_chk(a, "a"), b=_chk(b, "b"), da=_chk(da, "da"), db=_chk(db, "db"),
                            n=a.numel(), relu_first=int(relu_first)), stream())
    return da, db


def relu_bwd(dy, y, out=None):
    require()
    if out is None:
        out = torch.empty_like(dy)
    hip().relu_bwd(dict(dy=_chk(dy, "dy"), y=_chk(y, "y"), dx=_chk(out, "dx"), n=dy.numel()), stream())
    return out


def shortcut_a_fwd(x):
    require()
    b, h, w_, c = x.shape
    if h % 2 or w_ % 2 or c % 2:
        raise ValueError("shortcut_a: even H, W, C required")
    y = torch.empty((b, h // 2, w_ // 2, 2 * c), device=x.device, dtype=torch.float32)
    hip().shortcut_a_fwd(dict(x=_chk(x, "x"), y=_chk(y, "y"), B=b, H=h, W=w_, C=c), stream())
    return y


def shortcut_a_bwd(dy, x_shape, out=None, accumulate=False):
    require()
    b, h, w_, c = x_shape
    if tuple(dy.shape) != (b, h // 2, w_ // 2, 2 * c):
        raise ValueError("shortcut_a_bwd: shape mismatch")
    if out is None:
        out = torch.empty(x_shape, device=dy.device, dtype=torch.float32)
    hip().shortcut_a_bwd(dict(dy=_chk(dy, "dy"), dx=_chk(out, "dx"), B=b, H=h, W=w_, C=c, accumulate=int(accumulate)), stream())
    return out


def maxpool2x2_fwd(x):
    require()
    b, h, w_, c = x.shape
    y = torch.empty((b, (h + 1) // 2, (w_ + 1) // 2, c), device=x.device, dtype=torch.float32)
    arg = torch.empty(y.shape, device=x.device, dtype=torch.uint8)
    hip().maxpool2x2_fwd(dict(x=_chk(x, "x"), y=_chk(y, "y"), argmax=arg.data_ptr(), B=b, H=h, W=w_, C=c), stream())
    return y, arg


def conv1_pool_fwd(x, w, bias, out=None, arg=None):
    """relu(conv5x5 SAME(x, w) + bias) followed by the 2x2/2 max pool, for [B,28,28,1] images and a
    [5,5,1,32] filter, without materialising the full-resolution activation. Returns the pooled
    maximum [B,14,14,32] and the winner positions (uint8, as maxpool2x2_fwd)."""
    require()
    b = int(x.shape[0])
    if tuple(x.shape[1:]) != (28, 28, 1) or tuple(w.shape) != (5, 5, 1, 32) or bias.numel() != 32:
        raise ValueError("conv1_pool_fwd: expected [B,28,28,1] images, a [5,5,1,32] filter and 32 biases")
    if out is None:
        out = torch.empty((b, 14, 14, 32), device=x.device, dtype=torch.float32)
    if arg is None:
        arg = torch.empty((b, 14, 14, 32), device=x.device, dtype=torch.uint8)
    if tuple(out.shape) != (b, 14, 14, 32) or tuple(arg.shape) != (b, 14, 14, 32) or arg.dtype != torch.uint8:
        raise ValueError("conv1_pool_fwd: outputs must be [B,14,14,32] (float32 / uint8)")
    hip().conv1_pool_fwd(dict(x=_chk(x, "x"), w=_chk(w, "w"), bias=_chk(bias, "bias"), p=_chk(out, "p"),
                              arg=_chk(arg, "arg", torch.uint8), B=b), stream())
    return out, arg


def conv1_pool_bwd(x, dp, arg, p, dw, dbias, slot_stride: int = 0):
    """Adds the filter gradient (dw [5,5,1,32]) and bias gradient (dbias [32]) of conv1_pool_fwd
    given the gradient dp at the pooled output; with slot_stride the sums go to the per-XCD
    copies starting at dw / dbias (see slot_reduce)."""
    require()
    b = int(x.shape[0])
    shp = (b, 14, 14, 32)
    if tuple(x.shape[1:]) != (28, 28, 1) or tuple(dp.shape) != shp or tuple(arg.shape) != shp or tuple(p.shape) != shp:
        raise ValueError("conv1_pool_bwd: shape mismatch")
    if dw.numel() != 25 * 32 or dbias.numel() != 32:
        raise ValueError("conv1_pool_bwd: dw must hold 5*5*1*32 and dbias 32 elements")
    if slot_stride:
        for t in (dw, dbias):
            last = (t.storage_offset() + (STAT_SLOTS - 1) * slot_stride + t.numel()) * 4
            if last > t.untyped_storage().nbytes():
                raise ValueError("conv1_pool_bwd: the slot copies do not fit the gradient storage")
    hip().conv1_pool_bwd(dict(x=_chk(x, "x"), dp=_chk(dp, "dp"), arg=_chk(arg, "arg", torch.uint8), p=_chk(p, "p"),
                              dw=_chk(dw, "dw"), dbias=_chk(dbias, "dbias"), slot_stride=slot_stride, B=b), stream())


def maxpool2x2_bwd(dy, arg, y, x_shape, relu_mask: bool = False):
    require()
    b, h, w_, c = x_shape
    if dy.shape != y.shape or arg.shape != y.shape:
        raise ValueError("maxpool2x2_bwd: shape mismatch")
    dx = torch.empty(x_shape, device=dy.device, dtype=torch.float32)
    hip().maxpool2x2_bwd(dict(dy=_chk(dy, "dy"), argmax=arg.data_ptr(), y=_chk(y, "y"), dx=_chk(dx, "dx"),
                              B=b, H=h, W=w_, C=c, relu_mask=int(relu_mask)), stream())
    return dx


def dropout(x, keep_prob: float, seed: int, offset: int, out=None, offset_dev=None):
    """Philox dropout; the same (seed, offset[, *offset_dev]) regenerates the same mask, which is
    how the backward pass applies it to the gradient without storing it."""
    require()
    if out is None:
        out = torch.empty_like(x)
    d = dict(x=_chk(x, "x"), y=_chk(out, "y"), n=x.numel(), keep_prob=float(keep_prob), seed=int(seed), offset=int(offset))
    if offset_dev is not None:
        d["offset_dev"] = _chk(offset_dev, "offset_dev", torch.int64)
    hip().dropout(d, stream())
    return out


def fc_head(h_raw, b1, w2, b2, labels, logits, loss, correct=None, *, dh=None, dw2=None, db2=None, db1=None,
            keep_prob: float = 1.0, seed: int = 0, offset: int = 0, offset_dev=None, max_blocks: int = 0):
    """Bias + ReLU + dropout of the 512-unit hidden layer, the output layer, softmax cross-entropy
    and the backward pass down to the hidden pre-activation, in one launch (csrc/kernels/fc_head.hip).
    Gradient outputs (dh written, dw2 / db2 / db1 accumulated) are all given or all None."""
    require()
    b, h = h_raw.shape
    k = w2.shape[1]
    if tuple(w2.shape) != (h, k) or b1.numel() != h or b2.numel() != k or labels.numel() != b or tuple(logits.shape) != (b, k):
        raise ValueError("fc_head: shape mismatch")
    d = dict(h_raw=_chk(h_raw, "h_raw"), b1=_chk(b1, "b1"), w2=_chk(w2, "w2"), b2=_chk(b2, "b2"),
             labels=_chk(labels, "labels", torch.int32), logits=_chk(logits, "logits"), loss=_chk(loss, "loss"),
             B=b, H=h, K=k, keep_prob=float(keep_prob), seed=int(seed), offset=int(offset), max_blocks=int(max_blocks))
    if correct is not None:
        d["correct"] = _chk(correct, "correct", torch.int32)
    grads = (dh, dw2, db2, db1)
    if any(g is not None for g in grads):
        if any(g is None for g in grads):
            raise ValueError("fc_head: give all gradient outputs or none")
        if tuple(dh.shape) != (b, h) or dw2.numel() != h * k or db2.numel() != k or db1.numel() != h:
            raise ValueError("fc_head: gradient shape mismatch")
        d.update(dh=_chk(dh, "dh"), dw2=_chk(dw2, "dw2"), db2=_chk(db2, "db2"), db1=_chk(db1, "db1"))
    if offset_dev is not None:
        d["offset_dev"] = _chk(offset_dev, "offset_dev", torch.int64)
    hip().fc_head(d, stream())
    return logits


def gemm(a, b, *, trans_a=False, trans_b=False, bias=None, relu=False, out=None, accumulate=False, variant=0, splits=0):
    """C = op(A) @ op(B) (+bias)(relu) with row-major 2-D tensors."""
    require()
    if a.dim() != 2 or b.dim() != 2:
        raise ValueError("gemm: 2-D operands only")
    m, k = (a.shape[1], a.shape[0]) if trans_a else (a.shape[0], a.shape[1])
    kb, n = (b.shape[1], b.shape[0]) if trans_b else (b.shape[0], b.shape[1])
    if k != kb:
        raise ValueError("gemm: inner dimensions differ")
    if out is None:
        out = torch.empty((m, n), device=a.device, dtype=torch.float32)
    elif tuple(out.shape) != (m, n):
        raise ValueError("gemm: out has the wrong shape")
    d = dict(A=_chk(a, "A"), B=_chk(b, "B"), C=_chk(out, "C"), M=m, N=n, K=k, lda=a.shape[1], ldb=b.shape[1], ldc=n,
             trans_a=int(trans_a), trans_b=int(trans_b), relu=int(relu), accumulate=int(accumulate),
             variant=int(variant), splits=int(splits))
    if bias is not None:
        if bias.numel() != n:
            raise ValueError("gemm: bias size mismatch")
        d["bias"] = _chk(bias, "bias")
    hip().gemm_f32(d, stream())
    return out


def col_sum(x2d, out, relu_of=None, masked=None):
    """out[C] += column sums of x2d. relu_of (same shape): sums x2d * 1[relu_of > 0] instead - the
    ReLU backward of x2d - and stores that masked tensor to `masked` when given (x2d itself is fine)."""
    require()
    n_rows, c = x2d.shape
    if out.numel() != c:
        raise ValueError("col_sum: size mismatch")
    d = dict(x=_chk(x2d, "x"), out=_chk(out, "out"), n_rows=n_rows, C=c)
    if relu_of is not None:
        if relu_of.numel() != x2d.numel():
            raise ValueError("col_sum: relu_of size mismatch")
        d["relu_of"] = _chk(relu_of, "relu_of")
        if masked is not None:
            if masked.numel() != x2d.numel():
                raise ValueError("col_sum: masked size mismatch")
            d["masked"] = _chk(masked, "masked")
    elif masked is not None:
        raise ValueError("col_sum: masked needs relu_of")
    hip().col_sum(d, stream())
    return out


def zero_fraction(x) -> float:
    require()
    z = torch.zeros(1, device=x.device, dtype=torch.int64)
    hip().zero_count(dict(x=_chk(x, "x"), zeros=z.data_ptr(), n=x.numel()), stream())
    return float(z.item()) / max(1, x.numel())


_HIST_LIMITS: Dict = {}


def tf_histogram(x) -> dict:
    """tf.summary.histogram of a device tensor: dict(min, max, num, sum, sum_squares,
    bucket_limit, bucket) with TensorFlow's default buckets and its run-length compression of
    empty buckets (host statement: utils/summary.py::histogram_of)."""
    require()
    import numpy as np

    from ..utils import summary as sm

    limits_np = sm.default_bucket_limits()
    key = str(x.device)
    if key not in _HIST_LIMITS:
        _HIST_LIMITS[key] = torch.as_tensor(limits_np).to(x.device)
    limits = _HIST_LIMITS[key]
    nb = limits.numel()
    counts = torch.zeros(nb, device=x.device, dtype=torch.int64)
    moments = torch.zeros(2, device=x.device, dtype=torch.float64)
    minmax = torch.tensor([2**31 - 1, -2**31], device=x.device, dtype=torch.int32)
    flat = x.reshape(-1)
    hip().tf_histogram(dict(x=_chk(flat, "x"), n=flat.numel(), limits=_chk(limits, "limits", torch.float64), nb=nb,
                            counts=_chk(counts, "counts", torch.int64), moments=_chk(moments, "moments", torch.float64),
                            minmax=_chk(minmax, "minmax", torch.int32)), stream())
    keys = minmax.cpu().numpy().astype(np.int32)
    bits = np.where(keys >= 0, keys, keys ^ np.int32(0x7FFFFFFF)).astype(np.int32)
    lo, hi = bits.view(np.float32)
    mom = moments.cpu().numpy()
    bl, bc = sm.compress_histogram(limits_np, counts.cpu().numpy().astype(np.float64))
    return {"min": float(lo), "max": float(hi), "num": float(flat.numel()), "sum": float(mom[0]), "sum_squares": float(mom[1]),
            "bucket_limit": bl, "bucket": bc}


def crop_whiten(src, index=None, offs=None, out_hw=None, whiten=True, out=None):
    require()
    n, hp, wp, c = src.shape
    h, w_ = out_hw if out_hw is not None else (hp, wp)
    b = index.numel() if index is not None else n
    if out is None:
        out = torch.empty((b, h, w_, c), device=src.device, dtype=torch.float32)
    elif tuple(out.shape) != (b, h, w_, c):
        raise ValueError("crop_whiten: out has the wrong shape")
    d = dict(src=_chk(src, "src"), out=_chk(out, "out"), B=b, HP=hp, WP=wp, H=h, W=w_, C=c, whiten=int(whiten))
    if index is not None:
        d["index"] = _chk(index, "index", torch.int32)
    if offs is not None:
        if offs.numel() != 2 * b:
            raise ValueError("crop offsets must be [B,2]")
        d["offs"] = _chk(offs, "offs", torch.int32)
    hip().crop_whiten(d, stream())
    return out
