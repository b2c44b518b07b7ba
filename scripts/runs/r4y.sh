#!/bin/bash
OUT=gpurun_out/r4y; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_hip_ops.py -m gpu -q -p no:cacheprovider -k "slabs or split_fp32" > $OUT/test.log 2>&1; echo "pytest rc=$?" > $OUT/rc.txt
tail -4 $OUT/test.log | cut -c1-200; cat $OUT/rc.txt; grep -E "^E  " $OUT/test.log | head -5
