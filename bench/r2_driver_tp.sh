set -e
cd $GRAFT_REPO_ROOT
{
echo "# bench/driver_throughput.sh resnet (batch 1024)"; bash bench/driver_throughput.sh resnet 1024 300
echo "# bench/driver_throughput.sh lenet (batch 1024)"; bash bench/driver_throughput.sh lenet 1024 300
echo "# bench/driver_throughput.sh lenet (batch 256)"; bash bench/driver_throughput.sh lenet 256 600
echo "# bench/driver_throughput.sh resnet (batch 256)"; bash bench/driver_throughput.sh resnet 256 600
} > gpurun_out/r2_entrypoint_throughput.txt 2>&1
