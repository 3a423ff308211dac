#!/bin/bash
# SQ counters per kernel of a ResNet-20 training step (one counter pass, kernels serialised by the profiler).
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc_step
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_step -o pmc -- python3 $R/bench.py --steps 2 --warmup 1 "$@" > $R/gpurun_out/pmc_step.log 2>&1
cd $R
python3 tools/pmc_summary.py gpurun_out/pmc_step > gpurun_out/pmc_step_summary.txt
