set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/lenet_prof
timeout -k 10 150 rocprofv3 --kernel-trace -d $R/gpurun_out/lenet_prof -o lenet -- python3 $R/bench/lenet_trace.py ${1:-1024} > $R/gpurun_out/lenet_prof.log 2>&1
cd $R
DB=$(find gpurun_out/lenet_prof -name "*.db" | head -1)
python3 - "$DB" > gpurun_out/lenet_stats.md <<'PY'
import collections, sqlite3, sys
sys.path.insert(0, "tools")
from trace_timeline import short
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end from kernels order by start"))
ad = [i for i, r in enumerate(rows) if "adam_kernel" in r[0]]
s, e = ad[-2] + 1, ad[-1] + 1
agg = collections.OrderedDict()
for name, t0, t1 in rows[s:e]:
    k = short(name); c, tot = agg.get(k, (0, 0.0)); agg[k] = (c + 1, tot + (t1 - t0) / 1e3)
total = sum(v[1] for v in agg.values())
print(f"step span {(rows[e-1][2]-rows[s][1])/1e3:.1f} us, sum of kernel durations {total:.1f} us, {e-s} kernels\n")
print("| kernel | launches | mean us | total us | share |\n|---|---:|---:|---:|---:|")
for k, (c, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"| `{k}` | {c} | {tot/c:.1f} | {tot:.1f} | {100*tot/total:.1f}% |")
PY
