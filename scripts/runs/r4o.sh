#!/bin/bash
# round 4, GPU call o: the bench line + the rocprofv3 stats / PMC passes of the same command (profiles/round4_*)
OUT=gpurun_out/r4o; mkdir -p $OUT
python bench.py --steps 20 --warmup 5 > $OUT/bench_full.json 2> $OUT/bench_full.err; echo "bench rc=$?" > $OUT/rc.txt
bash scripts/run_pmc.sh $OUT/pmc > $OUT/pmc.log 2>&1; echo "pmc rc=$?" >> $OUT/rc.txt
python3 scripts/rocpd_gaps.py $(ls $OUT/pmc/trace/*results.db | head -1) --timeline 400 > $OUT/timeline.txt 2>&1
rm -rf $OUT/pmc/trace
python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > $OUT/bench_after_pmc.json 2> $OUT/bench_after_pmc.err
cut -c1-600 $OUT/bench_full.json; cat $OUT/rc.txt; cat $OUT/pmc/gemm_nt_by_shape.txt; tail -3 $OUT/pmc.log
