#!/bin/bash
# Throughput of the real training entry points (data pipeline, logging, supervisor included).
#   bench/driver_throughput.sh resnet|lenet [batch] [steps]
set -e
cd "$(dirname "$0")/.."
D=$(mktemp -d)
M=${1:-resnet}
if [ "$M" = resnet ]; then
  E="DistributedResNet/src/resnet_distributed_train.py --num_residual_blocks 3"
else
  E="DistributedLeNet/src/mnist_distributed_train.py"
fi
timeout -k 10 280 python $E --train_dir $D/train --batch_size ${2:-1024} --max_steps ${3:-300} --log_every 50 \
  --save_interval_secs 600 2>&1 | grep -E "step [0-9]+, loss|Elapsed" | tail -6
rm -rf $D
