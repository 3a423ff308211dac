#!/bin/bash
# SQ counters per kernel of a LeNet training step (one counter pass, kernels serialised by the profiler).
#   bench/pmc_lenet.sh [batch]
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B=${1:-256}
rm -rf $R/gpurun_out/pmc_lenet
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_lenet -o pmc -- python3 $R/bench/lenet_trace.py $B > $R/gpurun_out/pmc_lenet.log 2>&1
cd $R
python3 tools/pmc_table.py gpurun_out/pmc_lenet > gpurun_out/pmc_lenet_b$B.md
