#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_update_log_gpu.py tests/test_lenet_and_driver_gpu.py -x -q -k "killed" --timeout 300 --timeout-method thread -p no:cacheprovider > gpurun_out/r2_kill_tests.log 2>&1
rc=$?; echo "rc=$rc"; tail -50 gpurun_out/r2_kill_tests.log; exit $rc
