#!/bin/bash
# round 4, GPU call d: chained steps (tests + A/B), speaker tests, timeline
OUT=gpurun_out/r4d; mkdir -p $OUT
python -m pytest tests/test_hip_graphs.py tests/test_hip_full_size_agents.py tests/test_hip_modules.py tests/test_hip_cfg3_cfg4.py -m gpu -q -p no:cacheprovider -k "chained or speaker or iteration_graph or envdrop_full or cfg3 or pulled or segmented" > $OUT/test.log 2>&1; echo "pytest rc=$?" > $OUT/rc.txt
B="--steps 100 --warmup 8 --no-cpu-baseline --no-secondary --no-roofline"
python bench.py $B > $OUT/bench_chain.json 2> $OUT/bench_chain.err
python bench.py $B --no-chain > $OUT/bench_nochain.json 2> $OUT/bench_nochain.err
python bench.py $B > $OUT/bench_chain2.json 2> $OUT/bench_chain2.err
python bench.py $B --no-chain > $OUT/bench_nochain2.json 2> $OUT/bench_nochain2.err
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python3 bench.py --steps 60 --warmup 8 --no-cpu-baseline --no-secondary --no-roofline > $OUT/bench_trace.json 2> $OUT/bench_trace.err
DB=$(ls $OUT/trace/*results.db $OUT/trace/*/*results.db 2>/dev/null | head -1)
python3 scripts/rocpd_gaps.py $DB --skip 0.6 --timeline 360 > $OUT/timeline.txt 2>&1
rm -rf $OUT/trace
python scripts/bench_agents.py speaker --dtype bf16 > $OUT/speaker.log 2>&1
tail -3 $OUT/test.log; cat $OUT/bench_*.json | cut -c1-200; tail -2 $OUT/speaker.log
