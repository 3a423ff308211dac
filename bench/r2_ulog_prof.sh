#!/bin/bash
# Kernel trace of rank 0 of a two-rank backup-workers run (both ranks on one GPU, gloo set-up).
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29631 WORLD_SIZE=2 LOCAL_WORLD_SIZE=2 LOCAL_RANK=0 HSA_ENABLE_IPC_MODE_LEGACY=0
ARGS="--gpus 2 --backend gloo --share-gpu --steps 60 --warmup 10 --mode backup --aggregate 2 --batch 256"
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/ulog_prof
RANK=1 timeout -k 10 200 python3 $R/bench.py $ARGS > $R/gpurun_out/ulog_prof_rank1.log 2>&1 &
PEER=$!
RANK=0 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ulog_prof -o u -- python3 $R/bench.py $ARGS > $R/gpurun_out/ulog_prof_rank0.log 2>&1
rc=$?
wait $PEER
echo "rank0 rc=$rc rank1 rc=$?"
cd $R
tail -2 gpurun_out/ulog_prof_rank0.log
f=$(find gpurun_out/ulog_prof -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("| kernel | calls | mean us | total ms | share |")
print("|---|---:|---:|---:|---:|")
for r in rows[:14]:
    name = r["Name"].replace("void dtf::(anonymous namespace)::", "").replace("dtf::(anonymous namespace)::", "").split("(")[0][:60]
    print(f"| `{name}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} | {float(r['TotalDurationNs'])/1e6:.2f} | {float(r['Percentage']):.1f}% |")
PY
