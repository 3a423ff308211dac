#!/usr/bin/env python3
"""Cost of a kernel boundary: N dependent launches of a trivial kernel, as a hipGraph replay and as
plain stream launches issued from C++ (Program.run), for a tiny and for a chip-filling grid."""
from __future__ import annotations

import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from distributed_tensorflow_amd.ops import native as nv  # noqa: E402

N = 200


def main():
    for numel in (256, 1 << 20, 4 << 20):
        x = torch.ones(numel, device="cuda")
        prog = nv.hip().Program()
        prog.begin_record()
        for _ in range(N):
            nv.flat_scale(x, 1.0)
        prog.end_record()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            prog.instantiate(s.cuda_stream)
            out = {}
            for mode in ("graph", "stream"):
                fn = (lambda: prog.replay(s.cuda_stream)) if mode == "graph" else (lambda: prog.run(s.cuda_stream))
                for _ in range(2):
                    fn()
                s.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0 = time.perf_counter()
                e0.record(s)
                for _ in range(5):
                    fn()
                e1.record(s)
                host = time.perf_counter() - t0
                s.synchronize()
                out[mode] = (e0.elapsed_time(e1) / (5 * N) * 1e3, host / (5 * N) * 1e6)
        print(f"flat_scale n={numel}: graph {out['graph'][0]:.2f} us/kernel (host {out['graph'][1]:.2f}); "
              f"stream {out['stream'][0]:.2f} us/kernel (host issue {out['stream'][1]:.2f})")


if __name__ == "__main__":
    main()
