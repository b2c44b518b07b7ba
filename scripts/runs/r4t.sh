#!/bin/bash
# round 4, GPU call t: LSTM bias gradients accumulated inside the BPTT launch: tests + timing
OUT=gpurun_out/r4t; mkdir -p $OUT
timeout -k 10 1100 python -m pytest tests/test_hip_modules.py tests/test_hip_graphs.py tests/test_hip_agents.py tests/test_hip_full_size_agents.py -m gpu -q -p no:cacheprovider > $OUT/test.log 2>&1; echo "pytest rc=$?" > $OUT/rc.txt
tail -3 $OUT/test.log | cut -c1-200; cat $OUT/rc.txt; grep -E "^FAILED|^E  " $OUT/test.log | head -10
B="--steps 100 --warmup 8 --no-cpu-baseline --no-secondary --no-roofline"
for i in 1 2 3; do
timeout -k 10 200 python bench.py $B > $OUT/bench_$i.json 2> $OUT/bench_$i.err
done
for f in $OUT/bench_*.json; do python -c "import json,sys; j=json.load(open(sys.argv[1])); print(sys.argv[1], j['ms_per_step'])" $f; done
