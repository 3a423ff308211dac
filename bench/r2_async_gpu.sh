#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_update_log_gpu.py "tests/test_lenet_and_driver_gpu.py::test_async_parameter_shards_in_hbm_over_ipc" "tests/test_lenet_and_driver_gpu.py::test_two_gpu_replicas_stay_identical" -x -q --timeout 200 --timeout-method thread -p no:cacheprovider > gpurun_out/r2_async_tests.log 2>&1
rc=$?
echo "pytest rc=$rc"; tail -30 gpurun_out/r2_async_tests.log
[ $rc -eq 0 ] || exit $rc
for m in "sync" "async"; do
  extra=""; [ $m = async ] && extra="--mode async"
  timeout -k 10 300 python bench.py --gpus 2 --backend gloo --share-gpu --steps 100 --warmup 10 $extra > gpurun_out/r2_share2_$m.json 2> gpurun_out/r2_share2_$m.err
  rc=$?; echo "$m rc=$rc"; python tools/bench_brief.py gpurun_out/r2_share2_$m.json; [ $rc -eq 0 ] || { tail -20 gpurun_out/r2_share2_$m.err; exit $rc; }
done
timeout -k 10 300 python bench.py --mode async --steps 100 --warmup 10 > gpurun_out/r2_one_async.json 2> gpurun_out/r2_one_async.err; echo "one async rc=$?"; python tools/bench_brief.py gpurun_out/r2_one_async.json
