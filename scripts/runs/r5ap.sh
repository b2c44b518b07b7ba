#!/bin/bash
# round 5, GPU call ap: what the BPTT launch pays for passengers -- placement (two groups per XCD) or their work?
OUT=gpurun_out/r5ap; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for F in "--no-ride-wgrads" "--no-ride-wgrads --tunable 13=1" "--no-ride-wgrads --tunable 13=2"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --stats -d $OUT/trace_$i -o trace -- python3 bench.py --steps 60 --warmup 8 --no-cpu-baseline --no-secondary --no-roofline $F > $OUT/bench_$i.json 2> $OUT/bench_$i.err
  python3 scripts/rocpd_stats.py $(ls $OUT/trace_$i/*results.db | head -1) --iters 72 > $OUT/stats_$i.txt 2>&1
  rm -rf $OUT/trace_$i
  echo "== [$F]"; grep -E "lstm_persist_bwd|per iteration" $OUT/stats_$i.txt | cut -c1-170
done
