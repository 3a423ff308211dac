"""bf16x6 forward kernel against the exact-fp32 kernel: max error vs fp64 on the same inputs, and time."""
import sys

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import test_headline_sizes_gpu as T  # noqa: E402
from distributed_tensorflow_amd.ops import native as nv  # noqa: E402
from distributed_tensorflow_amd.ops import reference as ref  # noqa: E402


def timeit(fn, iters=50, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    for hw, c in [(16, 32), (32, 16), (8, 64)]:
        for b in (3, 300, 1024):
            x = T.rnd(b, hw, hw, c, scale=2.0, seed=3) + 0.3
            gamma, beta = T.rnd(c, seed=4).abs() + 0.5, T.rnd(c, scale=0.2, seed=5)
            wt = T.rnd(3, 3, c, c, scale=0.1, seed=6)
            res = T.rnd(b, hw, hw, c, seed=7)
            a = torch.relu(ref.batch_norm(x.double(), gamma.double(), beta.double()))
            want = T.conv2d_same_unfold(a, wt.double(), 1) + res.double()
            bn = T.bn_state_for(x, gamma, beta)
            wpk, wsp = nv.pack_weights(wt), nv.pack_split_weights(wt)
            outs = {}
            for name, w, kw in (("fp32", wpk, {}), ("bf16x6", wsp, dict(split=True))):
                sums = nv.new_stats(c, "cuda")
                got = nv.conv_fprop(x, w, ks=3, cout=c, in_bn=bn, out_sums=sums, res_mode=nv.RES_IDENT, res=res, **kw)
                torch.cuda.synchronize()
                tot = nv.stats_total(sums)
                flat = want.reshape(-1, c)
                outs[name] = (T.rel(got, want), float((got.double() - want).abs().max()), T.rel(tot[:c], flat.sum(0)), T.rel(tot[c:], (flat * flat).sum(0)))
            line = f"{hw}x{hw}x{c} B={b}: " + " | ".join(f"{k}: rel {v[0]:.2e} abs {v[1]:.2e} sums {v[2]:.1e}/{v[3]:.1e}" for k, v in outs.items())
            if b == 1024:
                out = torch.empty_like(x)
                sums = nv.new_stats(c, "cuda")
                t32 = timeit(lambda: nv.conv_fprop(x, wpk, ks=3, cout=c, in_bn=bn, out_sums=sums, res_mode=nv.RES_IDENT, res=res, out=out))
                tsp = timeit(lambda: nv.conv_fprop(x, wsp, ks=3, cout=c, in_bn=bn, out_sums=sums, res_mode=nv.RES_IDENT, res=res, out=out, split=True))
                line += f" | time fp32 {t32:.1f} us, bf16x6 {tsp:.1f} us"
            print(line, flush=True)



if __name__ == "__main__":
    main()
