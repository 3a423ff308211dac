#!/bin/bash
# Per-GPU batch sweep of the headline benchmark (one GPU).
set -e
for b in 128 256 512 1024 2048; do
  timeout -k 10 150 python bench.py --steps 40 --warmup 10 --batch $b | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['config']['per_gpu_batch'], d['ms_per_step'], d['value'])"
done
