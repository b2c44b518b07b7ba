#!/bin/bash
# round 5, GPU call j: kernel trace of BASELINE config 3's per-rank iteration (IL T=7 + sampled A2C T=35, host reads every action)
OUT=gpurun_out/r5j; mkdir -p $OUT
timeout -k 10 300 python scripts/bench_agents.py a2c --poll-actions --steps 20 > $OUT/a2c.json 2> $OUT/a2c.err || exit 1
cat $OUT/a2c.json | tail -3
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/trace -o k -- python3 $GRAFT_REPO_ROOT/scripts/bench_agents.py a2c --poll-actions --steps 20 > $GRAFT_REPO_ROOT/$OUT/trace.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/rocpd_stats.py $OUT/trace/k_results.db --iters 28 > $OUT/kernel_stats.txt 2>&1; head -45 $OUT/kernel_stats.txt
