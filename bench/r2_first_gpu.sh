#!/bin/bash
# First GPU pass of a round: kernel tests, headline bench, launcher behaviour on a 1-GPU lease.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -q --timeout 180 --timeout-method thread -p no:cacheprovider > gpurun_out/r2_gputests.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/r2_gputests.log
tail -5 gpurun_out/r2_gputests.log
timeout -k 10 200 python bench.py > gpurun_out/r2_bench_default.json 2> gpurun_out/r2_bench_default.err || { echo bench failed; tail -5 gpurun_out/r2_bench_default.err; exit 1; }
cat gpurun_out/r2_bench_default.json
timeout -k 10 200 python bench.py --batch 256 > gpurun_out/r2_bench_b256.json 2> gpurun_out/r2_bench_b256.err && cat gpurun_out/r2_bench_b256.json
timeout -k 10 300 python bench.py --gpus 2 --backend gloo --share-gpu --steps 20 --warmup 3 > gpurun_out/r2_bench_share2.json 2> gpurun_out/r2_bench_share2.err; echo "share2 rc=$?"; cat gpurun_out/r2_bench_share2.json; tail -3 gpurun_out/r2_bench_share2.err
timeout -k 10 120 python bench.py --gpus 2 --steps 5 --warmup 1 > gpurun_out/r2_bench_refuse.json 2> gpurun_out/r2_bench_refuse.err; echo "refuse rc=$? (must be non-zero)"; tail -2 gpurun_out/r2_bench_refuse.err
