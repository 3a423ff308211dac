#!/bin/bash
# Shader clock and power while bench.py runs (sampled twice a second), against the idle reading.
out=gpurun_out/r3_clock_under_load.txt
: > $out
echo "== idle" >> $out
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" >> $out
timeout -k 10 120 python bench.py --steps 2500 --warmup 50 > gpurun_out/r3_clock_bench.json 2>/dev/null &
pid=$!
sleep 2.5
for i in 1 2 3 4 5 6; do
  echo "== under load, sample $i" >> $out
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" >> $out
  sleep 0.5
done
wait $pid
tail -1 gpurun_out/r3_clock_bench.json >> $out
cat $out
