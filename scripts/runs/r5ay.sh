#!/bin/bash
# round 5, GPU call ay: the encoder's posted launches (d x + column sums in the pack launch, layout changes in the bridge products') -- tests, headline A/B
OUT=gpurun_out/r5ay; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_hip_modules.py tests/test_hip_graphs.py tests/test_hip_headline_vs_oracle.py -q -m gpu -x -k "encoder or iteration_graph or pulled or segmented or speaker or posted or headline" > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -2 $OUT/tests.log | cut -c1-200
for rep in 1; do
  for v in "--no-dx-post" ""; do
    echo "headline [$v]: $(timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --no-roofline $v 2>$OUT/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")"
  done
done
