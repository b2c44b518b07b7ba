#!/bin/bash
# round 5, GPU call ao: the full GPU suite on the non-temporal / re-pipelined gather ride, ABI 17 (ride-carried shadows, pushed batches)
OUT=gpurun_out/r5ao; mkdir -p $OUT
timeout -k 10 1100 python -m pytest tests -q -m gpu -x > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -3 $OUT/tests.log | cut -c1-200
