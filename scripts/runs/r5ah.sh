#!/bin/bash
# round 5, GPU call ah: gather passengers -- index loads one iteration ahead (lane-parallel), no vmcnt(0) in the loop; 1 / 2 groups of feature loads in flight.  ride tests, headline A/B against the previous library
OUT=gpurun_out/r5ah; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_hip_staging.py -q -m gpu -x > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -2 $OUT/tests.log
for rep in 1 2 3; do
  for d in head n1 n2; do
    cp scripts/ab/lib_$d.so curriculum-learning-for-vln_amd/libvln_hip.so
    echo "headline $d: $(timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")"
  done
done
