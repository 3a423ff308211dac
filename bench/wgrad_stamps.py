#!/usr/bin/env python3
"""Phase timeline of the filter-gradient kernel (thread 0 of every block stamps the wall clock)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import native as nv  # noqa: E402

NAMES = {0: "entry", 1: "first tile staged", 2: "first tile mfma done", 3: "all tiles done", 4: "tap0 flushed", 5: "tap1 flushed",
         8: "tap4 flushed", 12: "tap8 flushed", 15: "exit"}


def main():
    B = 256
    for (h, cin, cout, s, kw) in [(32, 16, 16, 1, {}), (32, 16, 16, 1, dict(tiles_per_block=4)), (8, 64, 64, 1, {}), (16, 32, 32, 1, {})]:
        ho = h // s
        x = torch.randn(B, h, h, cin, device="cuda")
        g = torch.randn(B, ho, ho, cout, device="cuda")
        n = 9 * cin * cout
        slots = torch.zeros(nv.STAT_SLOTS, n, device="cuda")
        dw = slots[0].view(3, 3, cin, cout)
        for skip in (0, 1):
            st = torch.zeros(8192 * 16, device="cuda", dtype=torch.int64)
            for _ in range(3):
                st.zero_()
                nv.conv_wgrad(x, g, dw, stride=s, slot_stride=n, debug_skip=skip, dbg_stamps=st, **kw)
            torch.cuda.synchronize()
            m = st.view(-1, 16).cpu()
            m = m[m[:, 0] > 0]
            t0 = int(m[:, 0].min())
            print(f"{h}x{h} {cin}->{cout} {kw} skip={skip}: {m.shape[0]} blocks")
            for k in sorted(NAMES):
                col = m[:, k]
                col = col[col > 0]
                if col.numel():
                    us = (col - t0).double() / 100.0
                    print(f"     {NAMES[k]:22s} min {us.min():6.2f}  med {us.median():6.2f}  max {us.max():6.2f}")


if __name__ == "__main__":
    main()
