"""Which workgroups of a 512-workgroup launch share a CU, and which SIMD each of their waves runs on
(HW_ID / XCC_ID registers written by the split backward kernel's timing stamps)."""
import collections
import sys

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import test_headline_sizes_gpu as T  # noqa: E402
from distributed_tensorflow_amd.ops import native as nv  # noqa: E402

hw, c, b = 32, 16, 1024
x = T.rnd(b, hw, hw, c, scale=2.0, seed=3) + 0.3
gamma, beta = T.rnd(c, seed=4).abs() + 0.5, T.rnd(c, scale=0.2, seed=5)
wt = T.rnd(3, 3, c, c, scale=0.1, seed=6)
dy = T.rnd(b, hw, hw, c, seed=7)
bn = T.bn_state_for(x, gamma, beta)
n = wt.numel()
slots = torch.zeros(nv.STAT_SLOTS, n, device="cuda")
gs = nv.new_stats(c, "cuda")
out = torch.empty_like(x)
wpk = nv.pack_split_weights(wt, flip=True)
stamps = torch.zeros(1024 * 4 * 48, device="cuda", dtype=torch.int64)
nv.conv_bwd_fused(dy, x, bn, wpk, slots[0, :n].view(3, 3, c, c), gs, slot_stride=n, out=out, split=True, dbg_stamps=stamps)
torch.cuda.synchronize()
st = stamps.view(1024, 4, 48)[:512].cpu()
hwid, xcc = st[:, :, 42], st[:, :, 43]
print("raw HW_ID of workgroup 0, waves 0-3:", [hex(int(v)) for v in hwid[0]], "XCC_ID:", [hex(int(v)) for v in xcc[0]])
# gfx9 HW_ID layout: wave_id [3:0], simd_id [5:4], pipe_id [7:6], cu_id [11:8], sh_id [12], se_id [15:13]
simd = (hwid >> 4) & 3
cu = (hwid >> 8) & 15
sh = (hwid >> 12) & 1
se = (hwid >> 13) & 7
xc = xcc & 15
print("SIMD of waves 0-3, first 8 workgroups:", simd[:8].tolist())
where = collections.defaultdict(list)
for wg in range(512):
    where[(int(xc[wg, 0]), int(se[wg, 0]), int(sh[wg, 0]), int(cu[wg, 0]))].append(wg)
print("distinct (xcc, se, sh, cu):", len(where), " workgroups per CU:", collections.Counter(len(v) for v in where.values()))
pairs = [v for v in where.values() if len(v) == 2]
diffs = collections.Counter(v[1] - v[0] for v in pairs)
print("difference of the two workgroup indices sharing a CU (most common):", diffs.most_common(6))
print("workgroup index mod 8 -> xcc:", sorted(set((wg % 8, int(xc[wg, 0])) for wg in range(512))))
same = sum(1 for v in pairs if simd[v[0]].tolist() == simd[v[1]].tolist())
print("pairs whose waves 0-3 sit on the same SIMDs in both workgroups:", same, "of", len(pairs))
print("examples:", [(v, simd[v[0]].tolist(), simd[v[1]].tolist()) for v in pairs[:4]])
