#!/usr/bin/env python3
"""Where does a conv kernel's time go? Times the forward conv of each ResNet shape with parts
switched off (no MFMA / no input loads / no epilogue traffic), 20 launches per hipGraph replay so
host launch overhead is excluded."""
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


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    print(f"batch {B}; columns: full | no-MFMA | no-input-loads | no-epilogue-mem | only-MFMA(6) | nothing(7)")
    for (h, cin, cout, s) in SHAPES:
        ho = h // s
        x = torch.randn(B, h, h, cin, device="cuda")
        w = torch.randn(3, 3, cin, cout, device="cuda") * 0.1
        wpk = nv.pack_weights(w)
        y = torch.empty(B, ho, ho, cout, device="cuda")
        sums = torch.zeros(2 * cout, device="cuda", dtype=torch.float64)
        flat = x.reshape(-1, cin).double()
        bn = nv.BnState(torch.cat([flat.sum(0), (flat * flat).sum(0)]), torch.ones(cin, device="cuda"),
                        torch.zeros(cin, device="cuda"), flat.shape[0])
        res = torch.randn_like(y) if s == 1 else None
        kw = dict(res_mode=nv.RES_IDENT, res=res) if res is not None else {}
        flops = 2.0 * B * ho * ho * cout * 9 * cin
        row = []
        for skip in (0, 1, 2, 4, 6, 7):
            t = time_program(lambda: nv.conv_fprop(x, wpk, ks=3, cout=cout, stride=s, in_bn=bn, out=y, out_sums=sums,
                                                   debug_skip=skip, **kw))
            row.append(t)
        print(f"{h}x{h} {cin}->{cout} s{s}: " + " | ".join(f"{t:7.1f}us" for t in row) + f"   full = {flops / row[0] / 1e6:.1f} TFLOP/s")


if __name__ == "__main__":
    main()
