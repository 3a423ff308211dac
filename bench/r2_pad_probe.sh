#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_kernels_gpu.py tests/test_resnet_gpu.py -x -q --timeout 200 --timeout-method thread -p no:cacheprovider 2>&1 | tail -4
timeout -k 10 200 python bench.py --steps 100 --warmup 10 2>/dev/null > gpurun_out/probe.json; python tools/bench_brief.py gpurun_out/probe.json
timeout -k 10 200 python bench.py --steps 100 --warmup 10 --batch 256 2>/dev/null > gpurun_out/probe.json; python tools/bench_brief.py gpurun_out/probe.json
bash bench/step_prof.sh 1024 && head -14 gpurun_out/step_stats_1024.md
