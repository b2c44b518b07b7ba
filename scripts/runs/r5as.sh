#!/bin/bash
# round 5, GPU call as: the full GPU suite on the final sources + the other workloads' iteration times (no profiler)
OUT=gpurun_out/r5as; mkdir -p $OUT
timeout -k 10 1100 python -m pytest tests -q -m gpu -x > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -2 $OUT/tests.log | cut -c1-200
for w in "monitor bf16" "monitor fp32" "follower bf16" "follower fp32" "speaker bf16"; do
  set -- $w
  timeout -k 10 300 python3 scripts/bench_agents.py $1 --dtype $2 --steps 60 --fused-only > $OUT/$1_$2.log 2>&1
  echo "$w: $(grep ms_per_iteration $OUT/$1_$2.log | tail -1 | cut -c1-200)"
done
