#!/bin/bash
# round 4, GPU call l: BASELINE config 3's iteration (IL + sampled A2C, T = 35) as graph segments: test + timings
OUT=gpurun_out/r4l; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_hip_graphs.py tests/test_hip_cfg3_cfg4.py tests/test_hip_agents.py -m gpu -q -p no:cacheprovider -x -k "a2c or cfg3 or cfg4 or critic or sampl" > $OUT/test.log 2>&1; echo "pytest rc=$?" > $OUT/rc.txt
tail -5 $OUT/test.log; cat $OUT/rc.txt
if grep -q "rc=0" $OUT/rc.txt; then
timeout -k 10 300 python scripts/bench_agents.py a2c --steps 30 --warmup 8 > $OUT/a2c_segments.json 2> $OUT/a2c_segments.err
timeout -k 10 300 python scripts/bench_agents.py a2c --steps 30 --warmup 8 --no-action-read > $OUT/a2c_segments_noread.json 2> $OUT/a2c_segments_noread.err
timeout -k 10 300 python scripts/bench_agents.py a2c --steps 30 --warmup 8 --no-graph > $OUT/a2c_eager.json 2> $OUT/a2c_eager.err
timeout -k 10 300 python scripts/bench_agents.py a2c --steps 30 --warmup 8 --no-graph --no-action-read > $OUT/a2c_eager_noread.json 2> $OUT/a2c_eager_noread.err
cat $OUT/a2c_*.json; tail -3 $OUT/a2c_segments.err
fi
