#!/bin/bash
# round 5, GPU call av: the recurrences' own streamed operands (xproj, saved activations, outputs) non-temporal
OUT=gpurun_out/r5av; mkdir -p $OUT
for rep in 1 2 3; do
  for d in base recnt; do
    cp scripts/ab/lib_$d.so curriculum-learning-for-vln_amd/libvln_hip.so
    echo "headline $d: $(timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")"
  done
done
