#!/bin/bash
mkdir -p gpurun_out
for i in 1 2 3 4; do
  timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread -p no:cacheprovider > gpurun_out/r2_flaky_$i.log 2>&1
  rc=$?; echo "run $i rc=$rc: $(tail -1 gpurun_out/r2_flaky_$i.log)"
  [ $rc -eq 0 ] || { grep -n "^E \|FAILED" gpurun_out/r2_flaky_$i.log | head -20; exit $rc; }
done
