#!/bin/bash
# round 4, GPU call f: graphs tests, chained steps A/B x3, the full bench line, PMC passes
OUT=gpurun_out/r4f; mkdir -p $OUT
python -m pytest tests/test_hip_graphs.py tests/test_hip_modules.py -m gpu -q -p no:cacheprovider > $OUT/test.log 2>&1; echo "pytest rc=$?" > $OUT/rc.txt
B="--steps 100 --warmup 8 --no-cpu-baseline --no-secondary --no-roofline"
for i in 1 2 3; do
python bench.py $B > $OUT/bench_chain$i.json 2> $OUT/bench_chain$i.err
python bench.py $B --no-chain > $OUT/bench_nochain$i.json 2> $OUT/bench_nochain$i.err
done
python bench.py --steps 20 --warmup 5 > $OUT/bench_full.json 2> $OUT/bench_full.err; echo "bench rc=$?" >> $OUT/rc.txt
bash scripts/run_pmc.sh $OUT/pmc > $OUT/pmc.log 2>&1; echo "pmc rc=$?" >> $OUT/rc.txt
rm -rf $OUT/pmc/trace
tail -3 $OUT/test.log; cat $OUT/bench_*chain*.json | cut -c1-200; cat $OUT/rc.txt; cat $OUT/pmc/gemm_nt_by_shape.txt
