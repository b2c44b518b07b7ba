#!/bin/bash
# round 5, GPU call at: gather passengers 128 / 112 / 96 / 80 on their own XCDs (tunable 13); the fp32 headline
OUT=gpurun_out/r5at; mkdir -p $OUT
for rep in 1 2 3; do
  for t in 0 112 96 80; do
    echo "headline gather passengers<=$t: $(timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --no-roofline --tunable 13=$t 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")"
  done
done
echo "fp32: $(timeout -k 10 300 python bench.py --dtype fp32 --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")"
