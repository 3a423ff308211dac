#!/bin/bash
# Memory-side counters per kernel of a training step (one pass): L2 requests / hits / misses, vector-L1 stalls.
#   bench/pmc_mem.sh [bench.py arguments]
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc_mem
timeout -s KILL 150 rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TA_BUSY_avr SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_mem -o pmc -- python3 $R/bench.py --steps 2 --warmup 1 "$@" > $R/gpurun_out/pmc_mem.log 2>&1
cd $R
python3 tools/pmc_summary.py gpurun_out/pmc_mem > gpurun_out/pmc_mem_summary.txt
