#!/bin/bash
# Full GPU suite, then the aggregation policies with two ranks sharing the GPU (gloo set-up, IPC data plane).
set -o pipefail
mkdir -p gpurun_out
if [ -z "$SKIP_TESTS" ]; then
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread -p no:cacheprovider > gpurun_out/r2_gputests.log 2>&1
rc=$?
echo "pytest rc=$rc"; tail -6 gpurun_out/r2_gputests.log
[ $rc -eq 0 ] || exit $rc
fi
run() { # name, args...
  local name=$1; shift
  echo "== $name"
  timeout -k 10 300 python bench.py "$@" > gpurun_out/r2_$name.json 2> gpurun_out/r2_$name.err
  local rc=$?
  echo "$name rc=$rc"; python tools/bench_brief.py gpurun_out/r2_$name.json; [ $rc -eq 0 ] || { tail -20 gpurun_out/r2_$name.err; exit $rc; }
}
run share2_sync   --gpus 2 --backend gloo --share-gpu --steps 100 --warmup 10
run share2_backup --gpus 2 --backend gloo --share-gpu --steps 100 --warmup 10 --mode backup
run share2_backup_straggler --gpus 2 --backend gloo --share-gpu --steps 100 --warmup 10 --mode backup --inject-straggler-ms 20
run share3_backup_straggler --gpus 3 --backend gloo --share-gpu --steps 100 --warmup 10 --mode backup --inject-straggler-ms 20 --batch 256
run share2_interval --gpus 2 --backend gloo --share-gpu --steps 100 --warmup 10 --mode interval --interval-ms 5
run share2_softkill --gpus 2 --backend gloo --share-gpu --steps 100 --warmup 10 --mode soft_kill --inject-straggler-ms 20
run one_backup --steps 100 --warmup 10 --mode backup
