#!/bin/bash
# round 5, GPU call aq: recurrence workgroups on XCDs 0-3, passengers on XCDs 4-7 -- recurrence / ride tests, headline A/B, BPTT passenger sweep
OUT=gpurun_out/r5aq; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_hip_staging.py tests/test_hip_modules.py tests/test_hip_graphs.py -q -m gpu -x -k "encoder or lstm or recurrence or iteration_graph or ride or riding or hand_off or persistent or pulled" > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -2 $OUT/tests.log | cut -c1-200
for rep in 1 2; do
  for v in "--tunable 15=1" "" "--tunable 11=96" "--tunable 11=128"; do
    echo "headline [$v]: $(timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --no-roofline $v 2>$OUT/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")"
  done
done
