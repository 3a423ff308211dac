#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_update_log_gpu.py -x -q -k "short_circuit" --timeout 200 --timeout-method thread -p no:cacheprovider 2>&1 | tail -25
