#!/bin/bash
# round 4, GPU call n: kernel-level breakdown of the Self-Monitor (fp32 and bf16 default) and the Follower iteration
OUT=gpurun_out/r4n; mkdir -p $OUT
export TMPDIR=/tmp
for w in monitor follower; do
for d in bf16 fp32; do
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python3 scripts/bench_agents.py $w --steps 30 --warmup 8 --dtype $d --fused-only > $OUT/${w}_$d.json 2> $OUT/${w}_$d.err
python3 scripts/rocpd_stats.py $(ls $OUT/trace/*results.db | head -1) --iters 42 > $OUT/stats_${w}_$d.txt
rm -rf $OUT/trace
done; done
cat $OUT/*.json; head -45 $OUT/stats_monitor_bf16.txt
