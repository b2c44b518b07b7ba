#!/bin/bash
# round 4, GPU call j: decoder gradient ride in the BPTT launch: tests, A/B x3, per-kernel stats
OUT=gpurun_out/r4j; mkdir -p $OUT
timeout -k 10 500 python -m pytest tests/test_hip_graphs.py -m gpu -q -p no:cacheprovider -x -k "ride or equals_eager" > $OUT/test.log 2>&1; echo "pytest rc=$?" > $OUT/rc.txt
if grep -q "rc=0" $OUT/rc.txt; then
B="--steps 100 --warmup 8 --no-cpu-baseline --no-secondary --no-roofline"
for i in 1 2 3; do
timeout -k 10 200 python bench.py $B > $OUT/bench_ride$i.json 2> $OUT/bench_ride$i.err
timeout -k 10 200 python bench.py $B --no-ride-wgrads > $OUT/bench_own$i.json 2> $OUT/bench_own$i.err
done
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python3 bench.py $B > $OUT/bench_trace.json 2> $OUT/bench_trace.err
python3 scripts/rocpd_stats.py $(ls $OUT/trace/*results.db | head -1) --iters 108 > $OUT/stats.txt
rm -rf $OUT/trace
for f in $OUT/bench_ride?.json $OUT/bench_own?.json; do python -c "import json,sys; j=json.load(open(sys.argv[1])); print(sys.argv[1], j['ms_per_step'], j['config'].get('decoder_wgrad_ride'))" $f; done
grep -n "persist_bwd\|wgrad\|colsum" $OUT/stats.txt | head
fi
tail -5 $OUT/test.log; cat $OUT/rc.txt
