set -e
cd /root/repo
timeout -k 10 200 python bench.py --steps 100 --warmup 10 > gpurun_out/bench_p50.log 2>&1
for m in sharded backup async; do
  MASTER_PORT=29611 timeout -k 10 200 python bench.py --steps 30 --warmup 5 --mode $m > gpurun_out/bench_mode_$m.log 2>&1
done
for m in sharded backup async; do
  timeout -k 10 250 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29633 bench.py --gpus 2 --steps 20 --warmup 3 --backend gloo --mode $m > gpurun_out/bench_mode2_$m.log 2>&1
done
