"""Forward 5x5 tile kernel: one vs two images per workgroup tile, and the data-gradient form."""
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

w = torch.randn(5, 5, 32, 64, device="cuda") * 0.05
wp = nv.conv5x5_pack(w)
bias = torch.zeros(64, device="cuda")
for b in (128, 256, 512, 1024, 4096):
    x = torch.randn(b, 14, 14, 32, device="cuda")
    dy = torch.randn(b, 14, 14, 64, device="cuda")
    y = torch.empty(b, 14, 14, 64, device="cuda")
    dx = torch.empty(b, 14, 14, 32, device="cuda")
    flop = 2.0 * b * 196 * 64 * 800
    r = [t(lambda: nv.conv5x5_tile(x, wp, bias=bias, out=y, imgs=i)) for i in (1, 2)]
    d = t(lambda: nv.conv5x5_tile(dy, w, flip=True, out=dx))
    print(f"B={b}: fwd imgs=1 {r[0]:.1f} us ({flop/r[0]/1e6:.0f} TF) imgs=2 {r[1]:.1f} us ({flop/r[1]/1e6:.0f} TF) | dgrad {d:.1f} us ({flop/d/1e6:.0f} TF)")

print("filter gradient (per-XCD copies):")
slots = torch.zeros(nv.STAT_SLOTS, 25 * 32 * 64, device="cuda")
for b in (256, 512, 1024, 2048, 4096):
    x = torch.randn(b, 14, 14, 32, device="cuda")
    dy = torch.randn(b, 14, 14, 64, device="cuda")
    flop = 2.0 * b * 196 * 64 * 800
    d = t(lambda: nv.conv5x5_wgrad(x, dy, slots[0].view(5, 5, 32, 64), slot_stride=slots.shape[1]))
    ws = torch.empty(nv.conv5x5_wgrad_parts(b), 25 * 32 * 64, device="cuda")
    dwp = torch.zeros(5, 5, 32, 64, device="cuda")
    pw = t(lambda: nv.conv5x5_wgrad(x, dy, dwp, partials=ws))
    print(f"B={b}: with partial results + reduction {pw:.1f} us ({flop/pw/1e6:.0f} TF)")
    g = t(lambda: nv.conv_wgrad(x, dy, slots[0].view(5, 5, 32, 64), stride=1, slot_stride=slots.shape[1]))
    print(f"B={b}: whole-image wgrad {d:.1f} us ({flop/d/1e6:.0f} TF) | general kernel {g:.1f} us ({flop/g/1e6:.0f} TF)")
