#!/bin/bash
# round 5, GPU call y: rollout-level context gradient of the Self-Monitor / Follower steps (vln_dctx_term) + wide-shallow products split for consumers
OUT=gpurun_out/r5y; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_hip_full_size_agents.py tests/test_hip_agents.py tests/test_hip_headline_vs_oracle.py tests/test_hip_graphs.py tests/test_rollout_tapes.py tests/test_hip_ops.py -q -m gpu -x -k "monitor or follower or other_agents or bn or mlp or attn_dot or add_n" > $OUT/tests.log 2>&1
echo "tests rc=$?" ; tail -3 $OUT/tests.log
for rep in 1 2; do
  echo "monitor bf16: $(timeout -k 10 200 python scripts/bench_agents.py monitor --fused-only --steps 20 2>/dev/null | tail -1 | cut -c1-130)"
  echo "follower bf16: $(timeout -k 10 200 python scripts/bench_agents.py follower --fused-only --steps 20 2>/dev/null | tail -1 | cut -c1-130)"
done
echo "monitor fp32: $(timeout -k 10 200 python scripts/bench_agents.py monitor --dtype fp32 --fused-only --steps 20 2>/dev/null | tail -1 | cut -c1-130)"
