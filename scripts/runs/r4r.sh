#!/bin/bash
# round 4, GPU call r: IL + A2C with the teacher-forced half's steps chained: test + A/B
OUT=gpurun_out/r4r; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_hip_graphs.py tests/test_hip_ops.py -m gpu -q -p no:cacheprovider -x -k "a2c or split_fp32" > $OUT/test.log 2>&1; echo "pytest rc=$?" > $OUT/rc.txt
tail -3 $OUT/test.log | cut -c1-200; cat $OUT/rc.txt
for i in 1 2; do
timeout -k 10 300 python scripts/bench_agents.py a2c --steps 30 --warmup 8 > $OUT/a2c_chain$i.json 2> $OUT/a2c_chain$i.err
timeout -k 10 300 python scripts/bench_agents.py a2c --steps 30 --warmup 8 --no-chain-il > $OUT/a2c_nochain$i.json 2> $OUT/a2c_nochain$i.err
done
for f in $OUT/a2c_*.json; do python -c "import json,sys; j=json.load(open(sys.argv[1])); print(sys.argv[1], j['ms_per_iteration'])" $f; done
