#!/bin/bash
# The BASELINE configurations with eight ranks, all sharing ONE MI355X (rehearsal of the 8-GPU protocols).
mkdir -p gpurun_out
run() { local name=$1; shift; echo "== $name"; timeout -k 10 400 python bench.py "$@" > gpurun_out/r2_$name.json 2> gpurun_out/r2_$name.err; local rc=$?; python tools/bench_brief.py gpurun_out/r2_$name.json; [ $rc -eq 0 ] || { grep -v "amdgpu.ids\|socket.cpp\|Gloo" gpurun_out/r2_$name.err | tail -15; exit $rc; }; }
C="--gpus 8 --backend gloo --share-gpu --batch 128 --steps 60 --warmup 10"
run e8_sync --model resnet32 $C
run e8_backup7 --model resnet32 $C --mode backup --aggregate 7
run e8_backup7_straggler --model resnet32 $C --mode backup --aggregate 7 --inject-straggler-ms 30
run e8_async --model resnet32 $C --mode async
run e8_interval --model resnet32 $C --mode interval --interval-ms 5
