#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests/test_lenet_and_driver_gpu.py -x -q -k "straggler_policies" --timeout 300 --timeout-method thread -p no:cacheprovider > gpurun_out/r2_driver_tests.log 2>&1
rc=$?; echo "rc=$rc"; tail -60 gpurun_out/r2_driver_tests.log; exit $rc
