#!/bin/bash
# round 5, GPU call ak: the decoder's weight shadows refreshed by the gather ride's passengers (out of the prologue launch) -- tests, headline A/B
OUT=gpurun_out/r5ak; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_hip_staging.py tests/test_hip_graphs.py -q -m gpu -x > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -3 $OUT/tests.log
for rep in 1 2 3; do
  for v in "--no-ride-shadows" ""; do
    echo "headline [$v]: $(timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --no-roofline $v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")"
  done
done
