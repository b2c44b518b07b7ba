#!/bin/bash
# round 4, GPU call u: Self-Monitor / Follower steps: the LSTM gate product's slabs consumed by the pointwise launch: tests + timing
OUT=gpurun_out/r4u; mkdir -p $OUT
timeout -k 10 1100 python -m pytest tests/test_hip_agents.py tests/test_hip_full_size_agents.py tests/test_rollout_tapes.py tests/test_hip_graphs.py -m gpu -q -p no:cacheprovider > $OUT/test.log 2>&1; echo "pytest rc=$?" > $OUT/rc.txt
tail -3 $OUT/test.log | cut -c1-200; cat $OUT/rc.txt; grep -E "^FAILED|^E  " $OUT/test.log | head -10
for i in 1 2; do
python scripts/bench_agents.py monitor --steps 40 --warmup 20 --dtype bf16 > $OUT/mon_$i.json 2> $OUT/mon_$i.err
python scripts/bench_agents.py follower --steps 40 --warmup 20 --dtype bf16 --fused-only > $OUT/fol_$i.json 2> $OUT/fol_$i.err
done
cat $OUT/mon_*.json $OUT/fol_*.json
