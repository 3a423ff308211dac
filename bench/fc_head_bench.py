"""Time of the fused classifier kernel (fc_head) alone, per batch size and workgroup cap."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import native as nv  # noqa: E402

dev = "cuda"
for b in (128, 256, 1024, 4096):
    h = torch.randn(b, 512, device=dev)
    b1, w2, b2 = torch.randn(512, device=dev), torch.randn(512, 10, device=dev) * 0.1, torch.randn(10, device=dev)
    labels = torch.randint(0, 10, (b,), device=dev, dtype=torch.int32)
    logits, loss = torch.empty(b, 10, device=dev), torch.zeros(1, device=dev)
    dh, dw2, db2, db1 = torch.empty(b, 512, device=dev), torch.zeros(512, 10, device=dev), torch.zeros(10, device=dev), torch.zeros(512, device=dev)
    row = []
    for cap in (8, 16, 32, 64, 128, 256, 512):
        def run():
            nv.fc_head(h, b1, w2, b2, labels, logits, loss, None, dh=dh, dw2=dw2, db2=db2, db1=db1, keep_prob=0.5, seed=1, max_blocks=cap)
        for _ in range(5):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(50):
            run()
        e1.record()
        torch.cuda.synchronize()
        row.append("cap %3d: %5.1f us" % (cap, e0.elapsed_time(e1) * 1e3 / 50))
    print("batch %4d  " % b + "  ".join(row), flush=True)
