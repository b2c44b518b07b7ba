#!/bin/bash
# round 5, GPU call l (re-run: the first run's A/B flags never reached the workload): the sampled rollout's in-step sampler and chained
# backward, A/B (2 alternations, 36 graph segments with polled actions), the one-graph handshake form, + the roofline block
OUT=gpurun_out/r5l; mkdir -p $OUT
for i in 1 2; do
for v in "" "--separate-sampler" "--no-chain-backward" "--separate-sampler --no-chain-backward"; do
  n=$(echo "$v" | tr -d ' -'); [ -z "$n" ] && n=default
  timeout -k 10 200 python scripts/bench_agents.py a2c --poll-actions --steps 20 $v 2>/dev/null | tail -1 > $OUT/${n}_$i.json || exit 1
  python -c "import json,sys; print(sys.argv[1], json.load(open(sys.argv[1]))['ms_per_iteration'])" $OUT/${n}_$i.json
done; done
timeout -k 10 200 python scripts/bench_agents.py a2c --handshake --roofline --steps 20 2>/dev/null | tail -1 | tee $OUT/handshake_roofline.json
