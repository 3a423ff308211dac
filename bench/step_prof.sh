#!/bin/bash
# Per-kernel totals of one ResNet-20 training step at batch $1 (rocprofv3 --kernel-trace + tools/trace_stats.py).
set -e
B=${1:-1024}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/step_prof_$B
timeout -k 10 200 rocprofv3 --kernel-trace -d $R/gpurun_out/step_prof_$B -o s -- python3 $R/bench.py --batch $B --steps 5 --warmup 3 > $R/gpurun_out/step_prof_$B.log 2>&1
cd $R
python3 tools/trace_stats.py $(find gpurun_out/step_prof_$B -name "*.db" | head -1) > gpurun_out/step_stats_$B.md
