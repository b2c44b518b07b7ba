#!/bin/bash
# round 4, GPU call h: XCD-aware tile order of the M = B * L gemm_nt launches (tunable 8): ops/encoder tests + A/B x3 + per-kernel stats
OUT=gpurun_out/r4h; mkdir -p $OUT
python -m pytest tests/test_hip_ops.py tests/test_hip_modules.py tests/test_hip_graphs.py -m gpu -q -p no:cacheprovider > $OUT/test.log 2>&1; echo "pytest rc=$?" > $OUT/rc.txt
B="--steps 100 --warmup 8 --no-cpu-baseline --no-secondary --no-roofline"
for i in 1 2 3; do
python bench.py $B > $OUT/bench_xcd$i.json 2> $OUT/bench_xcd$i.err
python bench.py $B --tunable 8=0 > $OUT/bench_grid$i.json 2> $OUT/bench_grid$i.err
done
export TMPDIR=/tmp
for v in 1 0; do
rocprofv3 --kernel-trace --stats -d $OUT/trace$v -o trace -- python3 bench.py $B --tunable 8=$v > $OUT/bench_trace$v.json 2> $OUT/bench_trace$v.err
python3 scripts/rocpd_stats.py $(ls $OUT/trace$v/*results.db | head -1) --iters 108 --shapes gemm_nt > $OUT/stats$v.txt
rm -rf $OUT/trace$v
done
tail -3 $OUT/test.log; for f in $OUT/bench_xcd?.json $OUT/bench_grid?.json; do python -c "import json,sys; print(sys.argv[1], json.load(open(sys.argv[1]))['ms_per_step'])" $f; done; cat $OUT/rc.txt
grep -n "80)" $OUT/stats1.txt $OUT/stats0.txt
