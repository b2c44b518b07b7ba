#!/bin/bash
# round 4, GPU call s: the decoder-only part of the pulled batch as a passenger of the forward recurrence launch: tests + A/B x3
OUT=gpurun_out/r4s; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_hip_graphs.py tests/test_hip_staging.py -m gpu -q -p no:cacheprovider -x > $OUT/test.log 2>&1; echo "pytest rc=$?" > $OUT/rc.txt
tail -3 $OUT/test.log | cut -c1-200; cat $OUT/rc.txt
if grep -q "rc=0" $OUT/rc.txt; then
B="--steps 100 --warmup 8 --no-cpu-baseline --no-secondary --no-roofline"
for i in 1 2 3; do
timeout -k 10 200 python bench.py $B > $OUT/bench_split$i.json 2> $OUT/bench_split$i.err
timeout -k 10 200 python bench.py $B --no-split-pull > $OUT/bench_whole$i.json 2> $OUT/bench_whole$i.err
done
for f in $OUT/bench_*.json; do python -c "import json,sys; j=json.load(open(sys.argv[1])); print(sys.argv[1], j['ms_per_step'], j['config'].get('batch_tail_under_recurrence'))" $f; done
fi
