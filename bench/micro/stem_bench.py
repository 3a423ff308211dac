"""Stem convolution kernels (3 -> 16 channels) against the general kernels, standalone."""
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

w = torch.randn(3, 3, 3, 16, device="cuda") * 0.3
wpk = nv.pack_weights(w)
for b in (64, 256, 1024, 4096):
    x = torch.randn(b, 32, 32, 3, device="cuda"); dy = torch.randn(b, 32, 32, 16, device="cuda")
    y = torch.empty(b, 32, 32, 16, device="cuda"); sums = nv.new_stats(16)
    slots = torch.zeros(nv.STAT_SLOTS, 512, device="cuda"); dw = slots[0, :432].view(3, 3, 3, 16)
    f_new = t(lambda: nv.conv_stem_fwd(x, w, out=y, out_sums=sums))
    f_old = t(lambda: nv.conv_fprop(x, wpk, ks=3, cout=16, out=y, out_sums=sums))
    g_new = t(lambda: nv.conv_stem_wgrad(x, dy, dw, slot_stride=512))
    g_old = t(lambda: nv.conv_wgrad(x, dy, dw, stride=1, slot_stride=512))
    mb = b * 1024 * 16 * 4 / 1e6
    print(f"B={b}: forward {f_new:.1f} us (general {f_old:.1f}) | filter gradient {g_new:.1f} us (general {g_old:.1f}) | 16-channel tensor {mb:.0f} MB")
