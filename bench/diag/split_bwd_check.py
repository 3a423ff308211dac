"""bf16x6 fused backward kernel against the exact-fp32 one: max error vs fp64 autograd on the same inputs
(data gradient, filter gradient, BatchNorm-backward sums), and time at batch 1024."""
import sys

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import test_headline_sizes_gpu as T  # noqa: E402
from distributed_tensorflow_amd.ops import native as nv  # noqa: E402
from distributed_tensorflow_amd.ops import reference as ref  # noqa: E402
from split_kernel_check import timeit  # noqa: E402


def run(x, dy, bn, w, split, gs, slots, n, c):
    wpk = nv.pack_split_weights(w, flip=True) if split else nv.pack_weights(w, flip=True)
    return nv.conv_bwd_fused(dy, x, bn, wpk, slots[0, :n].view(3, 3, c, c), gs, slot_stride=n, split=split)


shapes = [(16, 32), (32, 16)]
if nv.hip().conv_bwd_split_supported(1, 8, 8, 64):
    shapes.append((8, 64))
for hw, c in shapes:
    for b in (4, 96, 1024):
        x = T.rnd(b, hw, hw, c, scale=2.0, seed=3) + 0.3
        gamma, beta = T.rnd(c, seed=4).abs() + 0.5, T.rnd(c, scale=0.2, seed=5)
        wt = T.rnd(3, 3, c, c, scale=0.1, seed=6)
        dy = T.rnd(b, hw, hw, c, seed=7)
        bn = T.bn_state_for(x, gamma, beta)
        n = wt.numel()
        line = f"{hw}x{hw}x{c} B={b}:"
        if b <= 96:
            xd = x.double()
            u = ref.batch_norm(xd, gamma.double(), beta.double())
            a = torch.relu(u).requires_grad_(True)
            wd = wt.double().requires_grad_(True)
            y = T.conv2d_same_unfold(a, wd, 1)
            da, dw = torch.autograd.grad(y, (a, wd), dy.double())
            want_g = torch.where(u > 0, da, torch.zeros_like(da))
            for name, split in (("fp32", False), ("bf16x6", True)):
                slots = torch.zeros(nv.STAT_SLOTS, n, device="cuda")
                gs = nv.new_stats(c, "cuda")
                got = run(x, dy, bn, wt, split, gs, slots, n, c)
                torch.cuda.synchronize()
                line += f" | {name}: g rel {T.rel(got, want_g):.2e} abs {float((got.double() - want_g).abs().max()):.2e}" \
                        f" dw rel {T.rel(slots.sum(0).view(dw.shape), dw):.2e} abs {float((slots.double().sum(0).view(dw.shape) - dw).abs().max()):.2e}"
        else:
            slots = torch.zeros(nv.STAT_SLOTS, n, device="cuda")
            gs = nv.new_stats(c, "cuda")
            out = torch.empty_like(x)
            y = T.rnd(b, hw, hw, c, seed=8)
            ybn = T.bn_state_for(y, gamma, beta)
            ygs = nv.stats_from_totals(torch.cat([dy.reshape(-1, c).double().sum(0), (dy * y).reshape(-1, c).double().sum(0)]))
            res, dy_out = T.rnd(b, hw, hw, c, seed=9), torch.empty_like(x)
            res_a = T.rnd(b, hw // 2, hw // 2, 2 * c, seed=10) if c < 64 else None
            variants = [("plain", {}), ("bn", dict(dy_bn=(y, ybn, ygs))), ("bn+res", dict(dy_bn=(y, ybn, ygs), dy_res=res, dy_out=dy_out))]
            if res_a is not None:
                variants.append(("bn+resA", dict(dy_bn=(y, ybn, ygs), dy_res=res_a, dy_out=dy_out, dy_res_option_a=True)))
            for vname, kw in variants:
                line += f" | {vname}:"
                for name, split in (("fp32", False), ("bf16x6", True)):
                    wpk = nv.pack_split_weights(wt, flip=True) if split else nv.pack_weights(wt, flip=True)
                    t = timeit(lambda: nv.conv_bwd_fused(dy, x, bn, wpk, slots[0, :n].view(3, 3, c, c), gs, slot_stride=n, out=out, split=split, **kw))
                    line += f" {name} {t:.1f}"
            line += " us"
        print(line, flush=True)
