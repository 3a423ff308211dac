#!/usr/bin/env python3
"""Per-layer micro-benchmark of the convolution kernels at the ResNet shapes.

Times forward (BN prologue + residual + stats epilogue), data gradient (mask epilogue) and
filter gradient of each distinct conv shape with CUDA events, interleaving variants in one
process, and prints achieved TFLOP/s against the 155 TFLOP/s fp32-MFMA peak of an MI355X.
"""
from __future__ import annotations

import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from distributed_tensorflow_amd.ops import native as nv  # noqa: E402
from distributed_tensorflow_amd.ops.reference import same_pad  # noqa: E402

SHAPES = [  # (H, Cin, Cout, stride)
    (32, 3, 16, 1),
    (32, 16, 16, 1),
    (32, 16, 32, 2),
    (16, 32, 32, 1),
    (16, 32, 64, 2),
    (8, 64, 64, 1),
]


ITERS, WARM = 30, 5


def timeit(fn, iters=None, warm=None):
    iters = ITERS if iters is None else iters
    warm = WARM if warm is None else warm
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def bn_for(x):
    c = x.shape[-1]
    flat = x.reshape(-1, c).double()
    sums = torch.cat([flat.sum(0), (flat * flat).sum(0)])
    return nv.BnState(nv.stats_from_totals(sums), torch.ones(c, device="cuda"), torch.zeros(c, device="cuda"), flat.shape[0])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--tpb", type=int, default=0, help="wgrad tiles per block (0 = auto)")
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--warm", type=int, default=5)
    args = ap.parse_args()
    global ITERS, WARM
    ITERS, WARM = args.iters, args.warm
    B = args.batch
    print(f"batch {B}")
    print(f"{'shape':>22s} {'kind':>6s} {'us':>9s} {'TFLOP/s':>9s} {'GB/s(min)':>10s}")
    for (h, cin, cout, s) in SHAPES:
        ho = h // s
        x = torch.randn(B, h, h, cin, device="cuda")
        w = torch.randn(3, 3, cin, cout, device="cuda") * 0.1
        wpk = nv.pack_weights(w)
        y = torch.empty(B, ho, ho, cout, device="cuda")
        dy = torch.randn(B, ho, ho, cout, device="cuda")
        sums = nv.new_stats(cout)
        flops = 2.0 * B * ho * ho * cout * 9 * cin
        bn = bn_for(x) if cin >= 16 else None
        res = torch.randn_like(y) if s == 1 and cin == cout else None
        kw = dict(res_mode=nv.RES_IDENT, res=res) if res is not None else {}
        t = timeit(lambda: nv.conv_fprop(x, wpk, ks=3, cout=cout, stride=s, in_bn=bn, out=y, out_sums=sums, **kw))
        byt = (x.numel() + y.numel() * (2 if res is not None else 1)) * 4
        tag = f"{h}x{h} {cin}->{cout} s{s}"
        print(f"{tag:>22s} {'fwd':>6s} {t:9.1f} {flops / t / 1e6:9.1f} {byt / t / 1e3:10.0f}")
        if cin >= 16:
            wpk_t = nv.pack_weights(w, flip=True)
            _, pb, _ = same_pad(h, 3, s)
            dx = torch.empty_like(x)
            gs = nv.new_stats(cin)
            t = timeit(lambda: nv.conv_fprop(dy, wpk_t, ks=3, cout=cin, upsample=(s == 2), pad=2 - pb, out_hw=(h, h),
                                             epi_mode=nv.EPI_MASK, mask_src=x, mask_bn=bn, out_sums=gs, out=dx))
            byt = (dy.numel() + 2 * x.numel()) * 4
            print(f"{tag:>22s} {'dgrad':>6s} {t:9.1f} {flops / t / 1e6:9.1f} {byt / t / 1e3:10.0f}")
        dw = torch.zeros_like(w)
        t = timeit(lambda: nv.conv_wgrad(x, dy, dw, stride=s, in_bn=bn, tiles_per_block=args.tpb))
        byt = (x.numel() + dy.numel()) * 4
        print(f"{tag:>22s} {'wgrad':>6s} {t:9.1f} {flops / t / 1e6:9.1f} {byt / t / 1e3:10.0f}")
    # elementwise BN backward at the three activation sizes
    for (h, c) in [(32, 16), (16, 32), (8, 64)]:
        x = torch.randn(B, h, h, c, device="cuda")
        g = torch.randn_like(x)
        r = torch.randn_like(x)
        bn = bn_for(x)
        gs = nv.new_stats(c)
        out = torch.empty_like(x)
        t = timeit(lambda: nv.bn_bwd_apply(g, x, bn, gs, res_mode=nv.BWD_RES_IDENT, res=r, out=out))
        print(f"{f'{h}x{h}x{c}':>22s} {'bnbwd':>6s} {t:9.1f} {'':>9s} {4 * x.numel() * 4 / t / 1e3:10.0f}")


if __name__ == "__main__":
    main()
