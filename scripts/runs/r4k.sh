#!/bin/bash
# round 4, GPU call k: how many passengers should carry the decoder's gradient ride (tunable 11)
OUT=gpurun_out/r4k2; mkdir -p $OUT
B="--steps 100 --warmup 8 --no-cpu-baseline --no-secondary --no-roofline"
for i in 1 2 3; do
for np in 48 64 80 96; do
timeout -k 10 200 python bench.py $B --tunable 11=$np > $OUT/bench_np${np}_$i.json 2> $OUT/bench_np${np}_$i.err
done
timeout -k 10 200 python bench.py $B --no-ride-wgrads > $OUT/bench_own_$i.json 2> $OUT/bench_own_$i.err
done
for f in $OUT/bench_*.json; do python -c "import json,sys; j=json.load(open(sys.argv[1])); print(sys.argv[1], j['ms_per_step'])" $f; done
