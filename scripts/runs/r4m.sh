#!/bin/bash
# round 4, GPU call m: kernel-level breakdown of the IL + A2C iteration (graph segments, no action read -> GPU-bound)
OUT=gpurun_out/r4m; mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python3 scripts/bench_agents.py a2c --steps 30 --warmup 8 --no-action-read > $OUT/a2c.json 2> $OUT/a2c.err
python3 scripts/rocpd_stats.py $(ls $OUT/trace/*results.db | head -1) --iters 42 > $OUT/stats.txt
python3 scripts/rocpd_gaps.py $(ls $OUT/trace/*results.db | head -1) --timeline > $OUT/timeline.txt 2>&1
rm -rf $OUT/trace
cat $OUT/a2c.json; head -60 $OUT/stats.txt
