#!/bin/bash
# bench.py lines for the numbers the documents quote: ResNet-20 and LeNet, replayed batch and
# fresh batches through the GPU input path, at the benchmark batch and at the reference's.
out=gpurun_out/r3_bench_matrix.jsonl
: > $out
run() { timeout -k 10 200 python bench.py "$@" --steps 200 --warmup 20 2>> gpurun_out/r3_bench_matrix.err | tail -1 >> $out || exit 1; }
run --model resnet20 --batch 1024
run --model resnet20 --batch 256
run --model resnet20 --batch 1024 --input gpu_batcher
run --model resnet20 --batch 256 --input gpu_batcher
run --model lenet --batch 1024
run --model lenet --batch 256
run --model lenet --batch 128
run --model lenet --batch 1024 --input gpu_batcher
run --model lenet --batch 128 --input gpu_batcher
python tools/bench_brief.py $out
