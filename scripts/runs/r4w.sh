#!/bin/bash
OUT=gpurun_out/r4w; mkdir -p $OUT
for i in 1 2; do
timeout -k 10 300 python scripts/bench_agents.py a2c --steps 30 --warmup 8 > $OUT/a2c_sync$i.json 2> $OUT/a2c_sync$i.err
timeout -k 10 300 python scripts/bench_agents.py a2c --steps 30 --warmup 8 --poll-actions > $OUT/a2c_poll$i.json 2> $OUT/a2c_poll$i.err
done
for f in $OUT/a2c_*.json; do python -c "import json,sys; j=json.load(open(sys.argv[1])); print(sys.argv[1], j['ms_per_iteration'])" $f; done; tail -2 $OUT/a2c_poll1.err
