#!/bin/bash
# round 5, GPU call ae: the input BatchNorm's gradients from the first layer's weight gradient (vln_bn0_grads_from_wgrad)
OUT=gpurun_out/r5ae; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_hip_agents.py tests/test_hip_full_size_agents.py tests/test_hip_graphs.py tests/test_hip_headline_vs_oracle.py tests/test_rollout_tapes.py -q -m gpu -x -k "monitor or other_agents or bn or mlp or rollout_level" > $OUT/tests.log 2>&1
echo "tests rc=$?" ; tail -3 $OUT/tests.log
for rep in 1 2; do
  echo "monitor bf16: $(timeout -k 10 200 python scripts/bench_agents.py monitor --fused-only --steps 20 2>/dev/null | tail -1 | cut -c1-130)"
done
echo "monitor fp32: $(timeout -k 10 200 python scripts/bench_agents.py monitor --dtype fp32 --fused-only --steps 20 2>/dev/null | tail -1 | cut -c1-130)"
