#!/bin/bash
# Re-measures every number the documents quote on one GPU; results under gpurun_out/refresh/.
set -e
cd "$(dirname "$0")/.."
O=gpurun_out/refresh
mkdir -p $O
bash bench/batch_sweep.sh > $O/batch_sweep.txt 2>$O/batch_sweep.err
for m in resnet32 resnet110; do
  timeout -k 10 250 python bench.py --model $m --steps 20 --warmup 5 > $O/$m.json 2>$O/$m.err
done
for m in sharded backup async; do
  MASTER_PORT=29611 timeout -k 10 200 python bench.py --steps 30 --warmup 5 --mode $m > $O/mode_$m.json 2>$O/mode_$m.err
done
timeout -k 10 200 python bench.py --steps 50 --warmup 10 --rehearse-collectives > $O/rehearse.json 2>$O/rehearse.err
timeout -k 10 200 python bench.py --steps 50 --warmup 10 --rehearse-collectives --buckets 3 > $O/rehearse_3buckets.json 2>$O/rehearse_3buckets.err
bash bench/driver_throughput.sh resnet > $O/driver_resnet.txt 2>&1
bash bench/driver_throughput.sh lenet > $O/driver_lenet.txt 2>&1
timeout -k 10 200 python bench/lenet_bench.py > $O/lenet_bench.txt 2>&1
