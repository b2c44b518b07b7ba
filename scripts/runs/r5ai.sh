#!/bin/bash
# round 5, GPU call ai: the faster gather ride -- fewer passengers / non-temporal accesses, ride probe + headline
OUT=gpurun_out/r5ai; mkdir -p $OUT
for d in n1 n1nt; do
  for t in 0 96 64; do
    echo "== $d passengers<=$t"
    timeout -k 10 300 python scripts/ride_probe.py --lib scripts/ab/lib_$d.so --tunable 15=$t 2>&1 | grep -E "with the ride|ride, L"
  done
done
for rep in 1 2; do
  for d in head n1 n1nt; do
    cp scripts/ab/lib_$d.so curriculum-learning-for-vln_amd/libvln_hip.so
    for t in 0 96 64; do
      if [ $d = head ] && [ $t != 0 ]; then continue; fi
      echo "headline $d <=$t: $(timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --no-roofline --tunable 15=$t 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")"
    done
  done
done
