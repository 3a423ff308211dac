#!/bin/bash
# Bytes fetched from / written to memory per kernel of a bf16x6 ResNet-20 step (one counter pass each).
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_$c
  timeout -s KILL 150 rocprofv3 --pmc $c GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_$c -o pmc -- python3 $R/bench.py --steps 2 --warmup 1 --matmul bf16x6 --no-second-line > $R/gpurun_out/pmc_$c.log 2>&1
  (cd $R && python3 tools/pmc_summary.py gpurun_out/pmc_$c > gpurun_out/pmc_${c}_summary.txt)
done
