#!/usr/bin/env python3
"""Sweep of the filter-gradient kernel's launch shape (output tiles per block x tiles walked per
block) over the ResNet layer shapes, with and without per-XCD gradient copies; graph-timed."""
from __future__ import annotations

import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from distributed_tensorflow_amd.ops import native as nv  # noqa: E402
from conv_ablation import time_program  # noqa: E402

SHAPES = [(32, 3, 16, 1), (32, 16, 16, 1), (32, 16, 32, 2), (16, 32, 32, 1), (16, 32, 64, 2), (8, 64, 64, 1)]


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    for (h, cin, cout, s) in SHAPES:
        ho = h // s
        x = torch.randn(B, h, h, cin, device="cuda")
        g = torch.randn(B, ho, ho, cout, device="cuda")
        n = 9 * cin * cout
        stride = (n + 63) // 64 * 64
        slots = torch.zeros(nv.STAT_SLOTS, stride, device="cuda")
        dw = slots[0, :n].view(3, 3, cin, cout)
        bn = None
        if cin >= 16:
            flat = x.reshape(-1, cin).double()
            bn = nv.BnState(nv.stats_from_totals(torch.cat([flat.sum(0), (flat * flat).sum(0)])), torch.ones(cin, device="cuda"),
                            torch.zeros(cin, device="cuda"), flat.shape[0])
        flops = 2.0 * B * ho * ho * cout * 9 * cin
        print(f"{h}x{h} {cin}->{cout} s{s}:")
        base = time_program(lambda: nv.conv_wgrad(x, g, dw, stride=s, in_bn=bn))
        print(f"   default one copy {base:7.1f}us; no-flush {time_program(lambda: nv.conv_wgrad(x, g, dw, stride=s, in_bn=bn, debug_skip=4)):6.1f}"
              f" no-mfma {time_program(lambda: nv.conv_wgrad(x, g, dw, stride=s, in_bn=bn, debug_skip=1)):6.1f}"
              f" no-loads {time_program(lambda: nv.conv_wgrad(x, g, dw, stride=s, in_bn=bn, debug_skip=2)):6.1f}"
              f" nothing {time_program(lambda: nv.conv_wgrad(x, g, dw, stride=s, in_bn=bn, debug_skip=7)):6.1f}")
        best = None
        for co_t in (1, 2):
            if (cout // 16) % co_t:
                continue
            row = []
            for tpb in (0, 1, 2, 4, 8, 16):
                try:
                    t = time_program(lambda: nv.conv_wgrad(x, g, dw, stride=s, in_bn=bn, slot_stride=stride, co_tiles=co_t, tiles_per_block=tpb))
                except Exception as e:  # register budget exceeded for this instance etc.
                    row.append("   n/a ")
                    continue
                row.append(f"{t:7.1f}")
                if best is None or t < best[0]:
                    best = (t, co_t, tpb)
            print(f"   slots co_t={co_t}: tpb auto,1,2,4,8,16 -> " + " ".join(row))
        print(f"   best {best[0]:.1f}us (co_t={best[1]}, tiles_per_block={best[2]}) = {flops / best[0] / 1e6:.1f} TFLOP/s")


if __name__ == "__main__":
    main()
