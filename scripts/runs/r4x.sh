#!/bin/bash
OUT=gpurun_out/r4x; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_hip_graphs.py -m gpu -q -p no:cacheprovider -x -k "a2c" > $OUT/test.log 2>&1; echo "pytest rc=$?" > $OUT/rc.txt
tail -4 $OUT/test.log | cut -c1-220; cat $OUT/rc.txt
