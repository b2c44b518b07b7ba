#!/bin/bash
# round 4, GPU call i: fused d-embedding gradient (vln_embed_bwd_proj, split-K chunks of tunable 9): tests + A/B x3 + stats
OUT=gpurun_out/r4i; mkdir -p $OUT
python -m pytest tests/test_hip_modules.py tests/test_hip_graphs.py -m gpu -q -p no:cacheprovider > $OUT/test.log 2>&1; echo "pytest rc=$?" > $OUT/rc.txt
B="--steps 100 --warmup 8 --no-cpu-baseline --no-secondary --no-roofline"
for i in 1 2 3; do
python bench.py $B > $OUT/bench_split$i.json 2> $OUT/bench_split$i.err
python bench.py $B --tunable 9=0 > $OUT/bench_one$i.json 2> $OUT/bench_one$i.err
python bench.py $B --tunable 9=4 > $OUT/bench_four$i.json 2> $OUT/bench_four$i.err
done
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python3 bench.py $B > $OUT/bench_trace.json 2> $OUT/bench_trace.err
python3 scripts/rocpd_stats.py $(ls $OUT/trace/*results.db | head -1) --iters 108 --shapes gemm_nt > $OUT/stats.txt
rm -rf $OUT/trace
tail -3 $OUT/test.log; for f in $OUT/bench_split?.json $OUT/bench_one?.json $OUT/bench_four?.json; do python -c "import json,sys; print(sys.argv[1], json.load(open(sys.argv[1]))['ms_per_step'])" $f; done; cat $OUT/rc.txt
grep -n "80)\|embed_bwd" $OUT/stats.txt
