#!/bin/bash
# round 5, GPU call n: kernel traces of the other BASELINE workloads (Self-Monitor bf16 / fp32, Speaker-Follower, speaker)
OUT=gpurun_out/r5n; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for w in "monitor bf16" "monitor fp32" "follower bf16" "speaker bf16"; do
  set -- $w
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/trace_$1_$2 -o k -- python3 $GRAFT_REPO_ROOT/scripts/bench_agents.py $1 --dtype $2 --steps 20 --fused-only > $GRAFT_REPO_ROOT/$OUT/$1_$2.log 2>&1 || exit 1
  tail -1 $GRAFT_REPO_ROOT/$OUT/$1_$2.log | cut -c1-200
  python3 $GRAFT_REPO_ROOT/scripts/rocpd_stats.py $GRAFT_REPO_ROOT/$OUT/trace_$1_$2/k_results.db --iters 31 > $GRAFT_REPO_ROOT/$OUT/$1_$2_kernel_stats.txt 2>&1
  rm -rf $GRAFT_REPO_ROOT/$OUT/trace_$1_$2
done
