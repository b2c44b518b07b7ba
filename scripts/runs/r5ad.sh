#!/bin/bash
# round 5, GPU call ad: fixed-association dot in the candidate-logit kernels; gemm_rows2 (operand preparation under the MFMAs) A/B
OUT=gpurun_out/r5ad; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_hip_modules.py tests/test_hip_ops.py tests/test_hip_agents.py -q -m gpu -x > $OUT/tests.log 2>&1
echo "tests rc=$?" ; tail -2 $OUT/tests.log
timeout -k 10 300 python scripts/rows_probe.py
