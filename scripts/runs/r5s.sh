#!/bin/bash
# round 5, GPU call s: the bench line + the rocprofv3 stats / PMC passes of the same command (profiles/round5_*) + the cfg3 trace
OUT=gpurun_out/r5s; mkdir -p $OUT
# (the PMC passes first: they stamp profiles/round5_pmc.json with the kernel sources' hash, which the bench line's `traffic` checks)
bash scripts/run_pmc.sh $OUT/pmc > $OUT/pmc.log 2>&1; echo "pmc rc=$?" > $OUT/rc.txt
python bench.py --steps 20 --warmup 5 > $OUT/bench_full.json 2> $OUT/bench_full.err; echo "bench rc=$?" >> $OUT/rc.txt
python3 scripts/rocpd_gaps.py $(ls $OUT/pmc/trace/*results.db | head -1) --timeline 400 > $OUT/timeline.txt 2>&1
rm -rf $OUT/pmc/trace
python bench.py --steps 1000 --warmup 8 --no-secondary --no-cpu-baseline --no-roofline > $OUT/bench_1000.json 2> $OUT/bench_1000.err
cut -c1-600 $OUT/bench_full.json; cat $OUT/rc.txt; cat $OUT/pmc/gemm_nt_by_shape.txt; tail -3 $OUT/pmc.log; cut -c1-300 $OUT/bench_1000.json
