#!/bin/bash
# ResNet-32 under backup workers and asynchronous shards (ranks sharing one MI355X; see README).
mkdir -p gpurun_out
run() { local name=$1; shift; echo "== $name"; timeout -k 10 300 python bench.py "$@" > gpurun_out/r2_$name.json 2> gpurun_out/r2_$name.err; local rc=$?; python tools/bench_brief.py gpurun_out/r2_$name.json; [ $rc -eq 0 ] || { tail -20 gpurun_out/r2_$name.err; exit $rc; }; }
run r32_one_sync --model resnet32 --steps 100 --warmup 10
run r32_share3_sync --model resnet32 --gpus 3 --backend gloo --share-gpu --steps 60 --warmup 10 --batch 256
run r32_share3_backup --model resnet32 --gpus 3 --backend gloo --share-gpu --steps 60 --warmup 10 --batch 256 --mode backup --aggregate 2
run r32_share3_backup_straggler --model resnet32 --gpus 3 --backend gloo --share-gpu --steps 60 --warmup 10 --batch 256 --mode backup --aggregate 2 --inject-straggler-ms 20
run r32_share2_async --model resnet32 --gpus 2 --backend gloo --share-gpu --steps 60 --warmup 10 --mode async
run r32_one_async --model resnet32 --steps 100 --warmup 10 --mode async
