#!/bin/bash
# Raw rocprofv3 --stats summary of a bench.py run (all launches of 25 steps).
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/stats_prof
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/stats_prof -o bench -- python3 $R/bench.py --steps 20 --warmup 5 > $R/gpurun_out/stats_prof.log 2>&1
ls -la $R/gpurun_out/stats_prof
