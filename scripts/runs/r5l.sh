#!/bin/bash
# round 5, GPU call l: the sampled rollout's in-step sampler and chained backward, A/B (2 alternations), + kernel trace of the default
OUT=gpurun_out/r5l; mkdir -p $OUT
for i in 1 2; do
for v in "" "--separate-sampler" "--no-chain-backward" "--separate-sampler --no-chain-backward"; do
  n=$(echo "$v" | tr -d ' -'); [ -z "$n" ] && n=default
  timeout -k 10 200 python scripts/bench_agents.py a2c --poll-actions --steps 20 $v 2>/dev/null | tail -1 > $OUT/${n}_$i.json || exit 1
  python -c "import json,sys; print(sys.argv[1], json.load(open(sys.argv[1]))['ms_per_iteration'])" $OUT/${n}_$i.json
done; done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/trace -o k -- python3 $GRAFT_REPO_ROOT/scripts/bench_agents.py a2c --poll-actions --steps 20 > $GRAFT_REPO_ROOT/$OUT/trace.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/rocpd_stats.py $OUT/trace/k_results.db --iters 28 > $OUT/kernel_stats.txt 2>&1; head -24 $OUT/kernel_stats.txt
