#!/bin/bash
# round 5, GPU call w: Self-Monitor step with slab-summing consumers + the two-batch BN-MLP reading its inputs in place -- tests, then A/B
OUT=gpurun_out/r5w; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_hip_full_size_agents.py tests/test_hip_agents.py tests/test_hip_headline_vs_oracle.py tests/test_hip_graphs.py tests/test_rollout_tapes.py tests/test_hip_ops.py -q -m gpu -x -k "monitor or other_agents or bn or mlp or attn_dot or add_n" > $OUT/tests.log 2>&1
echo "tests rc=$?" ; tail -3 $OUT/tests.log
for rep in 1 2; do
  for v in "" "--tunable 9=1"; do
    echo "monitor bf16 $v: $(timeout -k 10 200 python scripts/bench_agents.py monitor --fused-only --steps 20 $v 2>/dev/null | tail -1 | cut -c1-130)"
  done
done
echo "monitor fp32: $(timeout -k 10 200 python scripts/bench_agents.py monitor --dtype fp32 --fused-only --steps 20 2>/dev/null | tail -1 | cut -c1-130)"
echo "monitor fp32 9=1: $(timeout -k 10 200 python scripts/bench_agents.py monitor --dtype fp32 --fused-only --steps 20 --tunable 9=1 2>/dev/null | tail -1 | cut -c1-130)"
