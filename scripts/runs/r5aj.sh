#!/bin/bash
# round 5, GPU call aj: non-temporal gather ride (final form) -- staging tests; + non-temporal operand reads of the gradient ride's pack phase, BPTT passengers 64 / 96 / 128
OUT=gpurun_out/r5aj; mkdir -p $OUT
cp scripts/ab/lib_f.so curriculum-learning-for-vln_amd/libvln_hip.so
timeout -k 10 600 python -m pytest tests/test_hip_staging.py -q -m gpu -x > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -2 $OUT/tests.log
for rep in 1 2; do
  for d in head f fw; do
    cp scripts/ab/lib_$d.so curriculum-learning-for-vln_amd/libvln_hip.so
    for t in 0 96 128; do
      if [ $d = head ] && [ $t != 0 ]; then continue; fi
      echo "headline $d bptt-passengers $t: $(timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --no-roofline --tunable 11=$t 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")"
    done
  done
done
