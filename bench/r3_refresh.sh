#!/bin/bash
# Refreshes the numbers quoted in the documents for both matrix paths: 200-step bench lines, batch sweeps,
# per-kernel totals at batch 1024 / 256 and one counter pass each.
set -e
mkdir -p gpurun_out
out=gpurun_out/r3_final_bench.jsonl
: > $out
for mm in fp32 bf16x6; do
  for b in 1024 256; do
    timeout -k 10 200 python bench.py --batch $b --matmul $mm --no-second-line --steps 200 --warmup 20 2>> gpurun_out/r3_final_bench.err | tail -1 >> $out
  done
done
python tools/bench_brief.py $out
for mm in fp32 bf16x6; do
  echo "ResNet-20, --matmul $mm, one MI355X, bench.py --batch B --steps 40 (per-GPU batch, ms/step, images/s)"
  for b in 128 256 512 1024 2048; do
    timeout -k 10 150 python bench.py --steps 40 --warmup 10 --batch $b --matmul $mm --no-second-line 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['config']['per_gpu_batch'], d['ms_per_step'], d['value'])"
  done
done > gpurun_out/r3_batch_sweep.txt
cat gpurun_out/r3_batch_sweep.txt
for mm in fp32 bf16x6; do
  bash bench/step_prof.sh 1024 ${mm}_b1024 --matmul $mm --no-second-line
  bash bench/step_prof.sh 256 ${mm}_b256 --matmul $mm --no-second-line
done
echo kernel stats done
for mm in fp32 bf16x6; do
  bash bench/pmc_step.sh --matmul $mm --no-second-line
  cp gpurun_out/pmc_step_summary.txt gpurun_out/r3_pmc_step_$mm.txt
done
echo counters done
# the driver's own invocation: headline + second line in one JSON object
timeout -k 10 200 python bench.py --steps 200 --warmup 20 2>/dev/null | tail -1 > gpurun_out/r3_final_bench_default.json
cat gpurun_out/r3_final_bench_default.json
