#!/bin/bash
# round 4, GPU call g: the full GPU suite, the full bench line, the N = 1 rehearsal of the segmented path (thread_local capture)
OUT=gpurun_out/r4g; mkdir -p $OUT
python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/test.log 2>&1; echo "pytest rc=$?" > $OUT/rc.txt
python bench.py --steps 20 --warmup 5 > $OUT/bench_full.json 2> $OUT/bench_full.err; echo "bench rc=$?" >> $OUT/rc.txt
python bench.py --steps 100 --warmup 8 --no-cpu-baseline --no-secondary --no-roofline --dp-path > $OUT/bench_dp.json 2> $OUT/bench_dp.err; echo "dp rc=$?" >> $OUT/rc.txt
tail -5 $OUT/test.log; cut -c1-300 $OUT/bench_full.json $OUT/bench_dp.json; cat $OUT/rc.txt
