#!/usr/bin/env python3
"""Where does a conv kernel's time go? Times the forward conv of each ResNet shape with parts
switched off (no MFMA / no input loads / no epilogue traffic), 20 launches per hipGraph replay so
host launch overhead is excluded.

The band kernels only carry the run-time switches in a build made with DTF_CONV_DEBUG=1
(`DTF_CONV_DEBUG=1 python build_native.py --only hip --force`); in the normal build `debug_skip` is
ignored by them and every column times the full kernel."""
from __future__ import annotations

import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from distributed_tensorflow_amd.ops import native as nv  # noqa: E402

SHAPES = [(32, 16, 16, 1), (16, 32, 32, 1), (8, 64, 64, 1), (32, 16, 32, 2)]
REPEAT = 20


def time_program(record) -> float:
    prog = nv.hip().Program()
    prog.begin_record()
    for _ in range(REPEAT):
        record()
    prog.end_record()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        prog.instantiate(s.cuda_stream)
        for _ in range(3):
            prog.replay(s.cuda_stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(5):
            prog.replay(s.cuda_stream)
        e1.record(s)
        s.synchronize()
    return e0.elapsed_time(e1) / (5 * REPEAT) * 1e3


NAMES = {0: "entry", 1: "staged", 2: "t0 lds ready", 3: "t0 mfma begin", 4: "t0 mfma end(grp0)", 5: "t0 done", 6: "t1 lds ready",
         7: "t1 mfma begin", 8: "t1 mfma end(grp0)", 9: "t1 done", 10: "t2 lds ready", 13: "t2 done", 14: "loop end", 15: "exit"}


def stamps(run):
    """Thread 0 of every block records the 100 MHz wall clock at phase boundaries; prints, per
    stamp, the min / median / max over blocks in microseconds since the first block entered."""
    for skip in (0, 7):
        st = torch.zeros(4096 * 16, device="cuda", dtype=torch.int64)
        for _ in range(3):
            st.zero_()
            run(st, skip)
        torch.cuda.synchronize()
        m = st.view(-1, 16).cpu()
        m = m[m[:, 0] > 0]
        t0 = int(m[:, 0].min())
        print(f"   stamps skip={skip}: {m.shape[0]} blocks")
        for k in sorted(NAMES):
            col = m[:, k]
            col = col[col > 0]
            if col.numel() == 0:
                continue
            us = (col - t0).double() / 100.0
            print(f"     {NAMES[k]:20s} min {us.min():6.2f}  med {us.median():6.2f}  max {us.max():6.2f}   (n={col.numel()})")


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    print(f"batch {B}; columns: full | no-MFMA | no-input-loads | no-epilogue-mem | only-MFMA(6) | nothing(7) | full-without-stats-flush(8) | nothing-at-all(15) | ping-pong kernel(64)")
    for (h, cin, cout, s) in SHAPES:
        ho = h // s
        x = torch.randn(B, h, h, cin, device="cuda")
        w = torch.randn(3, 3, cin, cout, device="cuda") * 0.1
        wpk = nv.pack_weights(w)
        y = torch.empty(B, ho, ho, cout, device="cuda")
        sums = nv.new_stats(cout)
        flat = x.reshape(-1, cin).double()
        bn = nv.BnState(nv.stats_from_totals(torch.cat([flat.sum(0), (flat * flat).sum(0)])), torch.ones(cin, device="cuda"),
                        torch.zeros(cin, device="cuda"), flat.shape[0])
        res = torch.randn_like(y) if s == 1 else None
        kw = dict(res_mode=nv.RES_IDENT, res=res) if res is not None else {}
        flops = 2.0 * B * ho * ho * cout * 9 * cin
        row = []
        for skip in (0, 1, 2, 4, 6, 7, 8, 15, 64):
            t = time_program(lambda: nv.conv_fprop(x, wpk, ks=3, cout=cout, stride=s, in_bn=bn, out=y, out_sums=sums,
                                                   debug_skip=skip, **kw))
            row.append(t)
        if os.environ.get("STAMPS"):
          stamps(lambda st, skip: nv.conv_fprop(x, wpk, ks=3, cout=cout, stride=s, in_bn=bn, out=y, out_sums=sums, debug_skip=skip,
                                               dbg_stamps=st, **kw))
        print(f"{h}x{h} {cin}->{cout} s{s}: " + " | ".join(f"{t:7.1f}us" for t in row) + f"   full = {flops / row[0] / 1e6:.1f} TFLOP/s")


if __name__ == "__main__":
    main()
