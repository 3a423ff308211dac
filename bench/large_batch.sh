#!/bin/bash
# ResNet-110 at growing per-GPU batch: throughput and peak HBM use (288 GB per MI355X).
set -e
cd "$(dirname "$0")/.."
for b in 1024 4096 8192; do
  timeout -k 10 280 python bench.py --model resnet110 --batch $b --steps 10 --warmup 3 2>/dev/null | tail -1
done
