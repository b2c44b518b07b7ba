#!/bin/bash
# round 5, GPU call ar: the resident-weights kernel for the encoder's input projection -- bit-identity test, probe, headline A/B
OUT=gpurun_out/r5ar; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_hip_ops.py -q -m gpu -x -k "resident or tall" > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -2 $OUT/tests.log | cut -c1-200
timeout -k 10 120 python scripts/wres_probe.py 2>&1 | tail -5 &&
for rep in 1 2; do
  for v in "--tunable 13=1" ""; do
    echo "headline [$v]: $(timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --no-roofline $v 2>$OUT/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")"
  done
done
