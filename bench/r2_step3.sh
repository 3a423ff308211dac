#!/bin/bash
# GPU suite after the timeline / watchdog / evaluator changes + kernel profiles of the current step.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread -p no:cacheprovider > gpurun_out/r2_gputests.log 2>&1
rc=$?
echo "pytest rc=$rc"; tail -8 gpurun_out/r2_gputests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python bench.py > gpurun_out/r2_bench_default.json 2> gpurun_out/r2_bench_default.err || { tail -5 gpurun_out/r2_bench_default.err; exit 1; }
python tools/bench_brief.py gpurun_out/r2_bench_default.json
bash bench/step_prof.sh 1024 && cat gpurun_out/step_stats_1024.md
bash bench/step_prof.sh 256 && cat gpurun_out/step_stats_256.md
