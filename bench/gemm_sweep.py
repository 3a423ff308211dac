"""Tile / split sweep of the fp32 GEMM on the fully connected products of the MNIST network.
Each configuration is captured 20 times into a graph and replayed, so the figure is kernel time
(plus the clear of a split product), not host launch time."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import native as nv  # noqa: E402

dev = "cuda"
REP = 20
NAMES = {1: "64x64x32g", 2: "128x128x32", 3: "64x64x128", 4: "64x128x64", 5: "128x128x64", 6: "64x64x64", 7: "64x64x64w8", 8: "64x64x128w8", 9: "128x128x32p"}


def timed(fn):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(REP):
                fn()
        g.replay()
        s.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(5):
            g.replay()
        e1.record(s)
        s.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * REP)


batches = [int(x) for x in sys.argv[1:]] or [128, 256, 1024]
for b in batches:
    shapes = {
        "fwd  [B,3136]x[3136,512]": (torch.randn(b, 3136, device=dev), torch.randn(3136, 512, device=dev), False, False, False),
        "dW1  [B,3136]^T x [B,512]": (torch.randn(b, 3136, device=dev), torch.randn(b, 512, device=dev), True, False, True),
        "dx   [B,512]x[3136,512]^T": (torch.randn(b, 512, device=dev), torch.randn(3136, 512, device=dev), False, True, False),
    }
    for name, (A, B, ta, tb, acc) in shapes.items():
        m = A.shape[1] if ta else A.shape[0]
        n = B.shape[0] if tb else B.shape[1]
        out = torch.zeros(m, n, device=dev)
        ref = (A.t() if ta else A) @ (B.t() if tb else B)
        res = []
        for variant in (0, 1, 2, 3, 4, 5, 6, 7, 8, 9):
            for splits in ((0,) if variant == 0 else (1, 2, 4, 8, 16, 32)):
                try:
                    out.zero_()
                    nv.gemm(A, B, trans_a=ta, trans_b=tb, out=out, accumulate=acc, variant=variant, splits=splits)
                    err = ((out - ref).abs().max() / ref.abs().max()).item()
                    t = timed(lambda: nv.gemm(A, B, trans_a=ta, trans_b=tb, out=out, accumulate=acc, variant=variant, splits=splits))
                except Exception as ex:   # a configuration the launcher refuses
                    print("  variant", variant, "splits", splits, "refused:", ex)
                    continue
                res.append((t, variant, splits, err))
        res.sort()
        auto = [r for r in res if r[1] == 0][0]
        print("batch %4d %s: auto %.1f us; best: " % (b, name, auto[0]) +
              ", ".join("%s/s%d %.1f us%s" % (NAMES[v], sp, t, "" if e < 1e-4 else " ERR %.1e" % e) for t, v, sp, e in res[:5] if v), flush=True)
        for vv in (6, 7, 9):
            print("      %s by splits: " % NAMES[vv] + ", ".join("s%d %.1f" % (sp, t) for t, v, sp, e in sorted(res, key=lambda r: r[2]) if v == vv), flush=True)
