#!/bin/bash
# round 5, GPU call al: kernel traces of the headline with / without the decoder's shadows riding (prologue / forward recurrence durations)
OUT=gpurun_out/r5al; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in a b; do
  if [ $v = a ]; then F="--no-ride-shadows"; else F=""; fi
  rocprofv3 --kernel-trace --stats -d $OUT/trace_$v -o trace -- python3 bench.py --steps 60 --warmup 8 --no-cpu-baseline --no-secondary --no-roofline $F > $OUT/bench_$v.json 2> $OUT/bench_$v.err
  python3 scripts/rocpd_stats.py $(ls $OUT/trace_$v/*results.db | head -1) --iters 72 > $OUT/stats_$v.txt 2>&1
  python3 scripts/rocpd_gaps.py $(ls $OUT/trace_$v/*results.db | head -1) --timeline 40 > $OUT/timeline_$v.txt 2>&1
  rm -rf $OUT/trace_$v
  echo "== $v [$F]"; grep -E "prologue|lstm_persist_g_fwd|per iteration" $OUT/stats_$v.txt | cut -c1-170
done
