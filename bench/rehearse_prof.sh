#!/bin/bash
# Kernel timeline of the bucketed step with a one-rank RCCL group: where do segment boundaries cost time?
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/rehearse_prof
MASTER_ADDR=127.0.0.1 MASTER_PORT=29755 timeout -k 10 200 rocprofv3 --kernel-trace -d $R/gpurun_out/rehearse_prof -o r -- python3 $R/bench.py --rehearse-collectives --steps 6 --warmup 3 > $R/gpurun_out/rehearse_prof.log 2>&1
cd $R
DB=$(find gpurun_out/rehearse_prof -name "*.db" | head -1)
python3 tools/trace_timeline.py $DB > gpurun_out/rehearse_timeline.txt
