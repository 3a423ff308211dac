#!/bin/bash
# Step time with the deterministic gradient flush (per-workgroup copies + fixed-order sum) against
# the float-atomic flush, ResNet-20, one GPU. Output: gpurun_out/r3_det_cost.txt
out=gpurun_out/r3_det_cost.txt
: > $out
for b in 128 256 512 1024; do
  for det in 1 0; do
    v=$(DTF_DETERMINISTIC=$det timeout -k 10 120 python bench.py --batch $b --steps 200 --warmup 20 | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'])") || exit 1
    echo "batch $b deterministic=$det ms/step images/s: $v" | tee -a $out
  done
done
