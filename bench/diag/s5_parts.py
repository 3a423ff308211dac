"""Time the split 5x5 kernels (forward + pool, pooled data gradient) at batch 1024."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from distributed_tensorflow_amd.ops import native as N

def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1000

B = int(os.environ.get("B", "1024"))
dev = "cuda"
torch.manual_seed(0)
w = torch.randn(5, 5, 32, 64, device=dev) * 0.05
bias = torch.randn(64, device=dev) * 0.1
x = torch.randn(B, 14, 14, 32, device=dev)
wp = N.conv5x5_split_pack(w)
wpt = N.conv5x5_split_pack(w, flip=True)
pm = torch.empty(B, 7, 7, 64, device=dev); parg = torch.empty(B, 7, 7, 64, device=dev, dtype=torch.uint8)
dp = torch.randn(B, 7, 7, 64, device=dev)
dx = torch.empty(B, 14, 14, 32, device=dev)
N.conv5x5_tile(x, wp, bias=bias, pool_out=(pm, parg), split=True)
pcopy = pm.clone()
print("fwd+pool  %.1f us" % timeit(lambda: N.conv5x5_tile(x, wp, bias=bias, pool_out=(pm, parg), split=True)))
print("dgrad     %.1f us" % timeit(lambda: N.conv5x5_tile(None, wpt, flip=True, out=dx, pooled=(dp, parg, pcopy), split=True)))
