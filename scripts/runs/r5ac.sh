#!/bin/bash
# round 5, GPU call ac: tall products on the row-block tiling (gemm_rows.h) -- agents' tests and timings, headline A/B
OUT=gpurun_out/r5ac; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_hip_full_size_agents.py tests/test_hip_agents.py tests/test_hip_ops.py -q -m gpu -x > $OUT/tests.log 2>&1
echo "tests rc=$?" ; tail -2 $OUT/tests.log
echo "monitor bf16: $(timeout -k 10 200 python scripts/bench_agents.py monitor --fused-only --steps 20 2>/dev/null | tail -1 | cut -c1-130)"
echo "monitor bf16 12=1: $(timeout -k 10 200 python scripts/bench_agents.py monitor --fused-only --steps 20 --tunable 12=1 2>/dev/null | tail -1 | cut -c1-130)"
echo "monitor fp32: $(timeout -k 10 200 python scripts/bench_agents.py monitor --dtype fp32 --fused-only --steps 20 2>/dev/null | tail -1 | cut -c1-130)"
echo "monitor fp32 12=1: $(timeout -k 10 200 python scripts/bench_agents.py monitor --dtype fp32 --fused-only --steps 20 --tunable 12=1 2>/dev/null | tail -1 | cut -c1-130)"
echo "follower bf16: $(timeout -k 10 200 python scripts/bench_agents.py follower --fused-only --steps 20 2>/dev/null | tail -1 | cut -c1-130)"
for rep in 1 2; do
  timeout -k 10 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-secondary --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline ms', d['ms_per_step'])"
done
