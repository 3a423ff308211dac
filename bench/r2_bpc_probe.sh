#!/bin/bash
for b in 128 256 512; do
  for bpc in 2 1; do
    DTF_CONV_BPC=$bpc timeout -k 10 150 python bench.py --steps 100 --warmup 10 --batch $b 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('batch', d['config']['per_gpu_batch'], 'fwd blocks/CU', $bpc, d['ms_per_step'], 'ms', d['value'], 'img/s')"
  done
done
