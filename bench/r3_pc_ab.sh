#!/bin/bash
# Same-box A/B of the producer / consumer organisation for the 16- and 32-channel backward kernels.
for rep in 1 2; do
  for pc in "" "16" "32" "16,32"; do
    for b in 1024 256; do
      DTF_BWD_SPLIT_PC=$pc timeout -k 10 120 python bench.py --matmul bf16x6 --batch $b --steps 200 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('pc=%-6s batch %4d  %.4f ms' % ('$pc', d['config']['per_gpu_batch'], d['ms_per_step']))" || exit 1
    done
  done
done
