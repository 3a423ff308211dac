#!/bin/bash
# Where does the asynchronous mode's extra time per step come from (one rank)?
mkdir -p gpurun_out
for sb in 16 4 1; do
  DTF_ASYNC_SERVICE_BLOCKS=$sb timeout -k 10 200 python bench.py --mode async --steps 100 --warmup 10 2>/dev/null > gpurun_out/probe.json; echo "service_blocks=$sb"; python tools/bench_brief.py gpurun_out/probe.json | head -1
done
DTF_ASYNC_BLOCKS=8 timeout -k 10 200 python bench.py --mode async --steps 100 --warmup 10 2>/dev/null > gpurun_out/probe.json; echo "blocks=8"; python tools/bench_brief.py gpurun_out/probe.json | head -1
timeout -k 10 200 python bench.py --mode sharded --steps 100 --warmup 10 --rehearse-collectives 2>/dev/null > gpurun_out/probe.json; echo "sharded (3 segments)"; python tools/bench_brief.py gpurun_out/probe.json | head -1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/async_trace
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/async_trace -o a -- python3 $R/bench.py --mode async --steps 6 --warmup 3 > $R/gpurun_out/async_trace.log 2>&1
cd $R
f=$(find gpurun_out/async_trace -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last full step: from the second to last async_pull to the last
idx = [i for i, r in enumerate(rows) if "async_pull" in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]["Start_Timestamp"])
prev_end = t0
for r in rows[a:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0][-50:]
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  gap {(s - prev_end) / 1e3:7.1f}  q{r.get('Queue_Id', '?')} {name}")
    prev_end = max(prev_end, e)
PY
