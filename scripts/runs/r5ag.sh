#!/bin/bash
# round 5, GPU call ag: the backward recurrence's hand-off stays in the XCD's L2 when the group verified it runs on one XCD -- probe, tests, headline A/B
OUT=gpurun_out/r5ag; mkdir -p $OUT
timeout -k 10 120 ./scripts/lstm_probe_local 2>&1 | grep -E "protocol|bwd launch|dgates" | sed -n 4,6p
timeout -k 10 900 python -m pytest tests/test_hip_modules.py tests/test_hip_graphs.py tests/test_hip_full_size_agents.py -q -m gpu -x -k "encoder or lstm or recurrence or iteration_graph or ride or speaker or monitor_cfg2 or persistent" > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -2 $OUT/tests.log
for rep in 1 2 3; do
  for v in "" "--tunable 14=1"; do
    echo "headline $v: $(timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --no-roofline $v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")"
  done
done
