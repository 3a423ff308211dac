#!/bin/bash
# Per-kernel totals of one ResNet-20 training step at batch $1 (rocprofv3 --kernel-trace + tools/trace_stats.py).
# $2 names the output (default: the batch), further arguments go to bench.py.
set -e
B=${1:-1024}
TAG=${2:-$B}
shift; shift || true
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/step_prof_$TAG
timeout -k 10 200 rocprofv3 --kernel-trace -d $R/gpurun_out/step_prof_$TAG -o s -- python3 $R/bench.py --batch $B --steps 5 --warmup 3 "$@" > $R/gpurun_out/step_prof_$TAG.log 2>&1
cd $R
python3 tools/trace_stats.py $(find gpurun_out/step_prof_$TAG -name "*.db" | head -1) > gpurun_out/step_stats_$TAG.md
