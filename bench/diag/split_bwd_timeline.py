"""Where a band's time goes in the two-workgroup split backward kernel: wall-clock stamps (100 MHz) of
lane 0 of every wave around staging, the barriers, the prefetch issue and the MFMA phase."""
import sys

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import test_headline_sizes_gpu as T  # noqa: E402
from distributed_tensorflow_amd.ops import native as nv  # noqa: E402

for hw, c in [(32, 16), (16, 32)]:
    b = 1024
    x = T.rnd(b, hw, hw, c, scale=2.0, seed=3) + 0.3
    gamma, beta = T.rnd(c, seed=4).abs() + 0.5, T.rnd(c, scale=0.2, seed=5)
    wt = T.rnd(3, 3, c, c, scale=0.1, seed=6)
    dy = T.rnd(b, hw, hw, c, seed=7)
    y = T.rnd(b, hw, hw, c, seed=8)
    bn, ybn = T.bn_state_for(x, gamma, beta), T.bn_state_for(y, gamma, beta)
    ygs = nv.stats_from_totals(torch.cat([dy.reshape(-1, c).double().sum(0), (dy * y).reshape(-1, c).double().sum(0)]))
    n = wt.numel()
    slots = torch.zeros(nv.STAT_SLOTS, n, device="cuda")
    gs = nv.new_stats(c, "cuda")
    out = torch.empty_like(x)
    wpk = nv.pack_split_weights(wt, flip=True)
    for vname, kw in (("plain", {}), ("bn", dict(dy_bn=(y, ybn, ygs)))):
        stamps = torch.zeros(1024 * 4 * 48, device="cuda", dtype=torch.int64)
        for _ in range(3):
            stamps.zero_()
            nv.conv_bwd_fused(dy, x, bn, wpk, slots[0, :n].view(3, 3, c, c), gs, slot_stride=n, out=out, split=True, dbg_stamps=stamps, **kw)
        torch.cuda.synchronize()
        st = stamps.view(1024, 4, 48)[:, :, :42].reshape(1024, 4, 7, 6).cpu().double()
        used = st[:, :, :, 0] > 0
        nb = int(used[:, 0, 0].sum())
        st = st[:nb]
        t0 = st[:, :, 0, 0].min()
        us = (st - t0) / 100.0   # 100 MHz -> us
        print(f"{hw}x{hw}x{c} {vname}: {nb} workgroups; per band (mean over workgroups, us): stage | barrier wait | issue loads | MFMA phase | barrier wait; by role")
        for role, waves in (("data gradient waves", [0, 1]), ("filter gradient waves", [2, 3])):
            u = us[:, waves]
            d = [(u[..., k + 1] - u[..., k]).mean(dim=(0, 1)) for k in range(5)]
            for band in range(7):
                if float(st[:, 0, band, 0].max()) == 0:
                    break
                print(f"   {role} band {band}: " + " | ".join(f"{float(x[band]):6.2f}" for x in d) + f"   start at {float(u[..., band, 0].mean()):7.2f}, end {float(u[..., band, 5].mean()):7.2f}")
