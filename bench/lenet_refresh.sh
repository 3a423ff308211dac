set -e
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 200 python -u bench/lenet_bench.py > gpurun_out/lenet_bench.log 2>&1
bash bench/lenet_prof.sh 256
cp gpurun_out/lenet_stats.md gpurun_out/lenet_b256_stats.md
bash bench/lenet_prof.sh 1024
cp gpurun_out/lenet_stats.md gpurun_out/lenet_b1024_stats.md
