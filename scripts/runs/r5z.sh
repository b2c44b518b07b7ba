#!/bin/bash
# round 5, GPU call z: Follower step launch folding + in-launch mean of the CE loss
OUT=gpurun_out/r5z; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_hip_full_size_agents.py tests/test_hip_agents.py tests/test_hip_graphs.py tests/test_rollout_tapes.py tests/test_hip_staging.py tests/test_hip_ops.py -q -m gpu -x -k "follower or other_agents or ce or cross or attn_dot or loss" > $OUT/tests.log 2>&1
echo "tests rc=$?" ; tail -3 $OUT/tests.log
for rep in 1 2; do
  echo "follower bf16: $(timeout -k 10 200 python scripts/bench_agents.py follower --fused-only --steps 20 2>/dev/null | tail -1 | cut -c1-130)"
done
echo "follower fp32: $(timeout -k 10 200 python scripts/bench_agents.py follower --dtype fp32 --fused-only --steps 20 2>/dev/null | tail -1 | cut -c1-130)"
