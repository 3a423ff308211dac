#!/bin/bash
# Device update log: the new GPU tests, then the suites it touches.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_update_log_gpu.py -x -v --timeout 200 --timeout-method thread -p no:cacheprovider > gpurun_out/r2_ulog_tests.log 2>&1
rc=$?
echo "ulog pytest rc=$rc"
tail -25 gpurun_out/r2_ulog_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 900 python -u -m pytest tests/test_lenet_and_driver_gpu.py tests/test_resnet_gpu.py -x -q --timeout 200 --timeout-method thread -p no:cacheprovider > gpurun_out/r2_ulog_tests2.log 2>&1
rc=$?
echo "driver pytest rc=$rc"
tail -15 gpurun_out/r2_ulog_tests2.log
exit $rc
