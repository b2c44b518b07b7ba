#!/bin/bash
# round 5, GPU call au: the forward recurrence's granules stay in the XCD's L2 when the group verified it runs on one XCD -- tests, headline A/B (tunable 14)
OUT=gpurun_out/r5au; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_hip_modules.py tests/test_hip_staging.py tests/test_hip_graphs.py -q -m gpu -x -k "encoder or lstm or recurrence or iteration_graph or ride or riding or hand_off or persistent or pulled or speaker or sync_workspace" > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -2 $OUT/tests.log | cut -c1-200
for rep in 1 2 3; do
  for v in "--tunable 14=1" ""; do
    echo "headline [$v]: $(timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --no-roofline $v 2>$OUT/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")"
  done
done
