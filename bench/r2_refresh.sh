#!/bin/bash
# Refreshes the numbers quoted in the documents: batch sweep, ResNet-110, LeNet, counter pass.
mkdir -p gpurun_out
{ echo "ResNet-20, fp32, one MI355X, bench.py --batch B --steps 40 (per-GPU batch, ms/step, images/s)"; bash bench/batch_sweep.sh; } > gpurun_out/r2_batch_sweep.txt 2>gpurun_out/r2_batch_sweep.err
cat gpurun_out/r2_batch_sweep.txt
bash bench/large_batch.sh > gpurun_out/r2_resnet110.txt 2>&1; python tools/bench_brief.py gpurun_out/r2_resnet110.txt
timeout -k 10 200 python bench/lenet_bench.py > gpurun_out/r2_lenet.txt 2>&1; cat gpurun_out/r2_lenet.txt | tail -5
bash bench/pmc_step.sh; cat gpurun_out/pmc_step_summary.txt | head -60
