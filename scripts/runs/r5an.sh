#!/bin/bash
# round 5, GPU call an: kernel trace of the headline with batches sent ahead (push)
OUT=gpurun_out/r5an; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --memory-copy-trace --stats -d $OUT/trace -o trace -- python3 bench.py --steps 60 --warmup 8 --no-cpu-baseline --no-secondary --no-roofline --batch-source push > $OUT/bench.json 2> $OUT/bench.err
python3 scripts/rocpd_stats.py $(ls $OUT/trace/*results.db | head -1) --iters 72 > $OUT/stats.txt 2>&1
python3 scripts/rocpd_gaps.py $(ls $OUT/trace/*results.db | head -1) --timeline 140 > $OUT/timeline.txt 2>&1
rm -rf $OUT/trace
grep -E "prologue|lstm_persist_g_fwd|per iteration|copy|Copy" $OUT/stats.txt | cut -c1-170
grep -n "gaps >" $OUT/timeline.txt | head -3
