"""Where is the fixed cost of the whole-image 5x5 filter-gradient kernel? Small batches (one image per
workgroup, fewer workgroups) and the reduction kernel alone."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from distributed_tensorflow_amd.ops import native as nv

def t(fn, n=30):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

n = 25 * 32 * 64
for b in (8, 64, 128, 256, 257, 512):
    x = torch.randn(b, 14, 14, 32, device="cuda"); dy = torch.randn(b, 14, 14, 64, device="cuda")
    parts = nv.conv5x5_wgrad_parts(b)
    ws = torch.empty(parts, n, device="cuda"); dw = torch.zeros(5, 5, 32, 64, device="cuda")
    full = t(lambda: nv.conv5x5_wgrad(x, dy, dw, partials=ws))
    red = t(lambda: nv.hip().partial_sum(dict(y=dw.data_ptr(), ws=ws.data_ptr(), n=n, parts=parts, stride=n), nv.stream()))
    print(f"B={b}: parts {parts}, kernel + reduction {full:.1f} us, reduction alone {red:.1f} us")
