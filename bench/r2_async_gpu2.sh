#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_update_log_gpu.py "tests/test_lenet_and_driver_gpu.py::test_async_parameter_shards_in_hbm_over_ipc" -x -q --timeout 200 --timeout-method thread -p no:cacheprovider > gpurun_out/r2_async_tests.log 2>&1
rc=$?
echo "pytest rc=$rc"; tail -5 gpurun_out/r2_async_tests.log
[ $rc -eq 0 ] || { tail -40 gpurun_out/r2_async_tests.log; exit $rc; }
timeout -k 10 300 python bench.py --mode async --steps 100 --warmup 10 > gpurun_out/r2_one_async.json 2> gpurun_out/r2_one_async.err; echo "one async rc=$?"; python tools/bench_brief.py gpurun_out/r2_one_async.json
timeout -k 10 300 python bench.py --gpus 2 --backend gloo --share-gpu --steps 100 --warmup 10 --mode async > gpurun_out/r2_share2_async.json 2> gpurun_out/r2_share2_async.err; echo "share2 async rc=$?"; python tools/bench_brief.py gpurun_out/r2_share2_async.json
timeout -k 10 300 python bench.py --gpus 2 --backend gloo --share-gpu --steps 100 --warmup 10 > gpurun_out/r2_share2_sync.json 2> gpurun_out/r2_share2_sync.err; echo "share2 sync rc=$?"; python tools/bench_brief.py gpurun_out/r2_share2_sync.json
