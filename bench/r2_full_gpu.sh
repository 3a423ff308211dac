#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread -p no:cacheprovider > gpurun_out/r2_gputests.log 2>&1
rc=$?
echo "pytest rc=$rc"; tail -8 gpurun_out/r2_gputests.log
exit $rc
