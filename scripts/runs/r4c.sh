#!/bin/bash
# round 4, GPU call c: tests, headline bench by batch source, dp-path, timeline of one iteration, Self-Monitor A/B, speaker
OUT=gpurun_out/r4c; mkdir -p $OUT
python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/test.log 2>&1; echo "pytest rc=$?" > $OUT/rc.txt
cp gpurun_out/parity_report.json $OUT/parity_report.json
B="--steps 100 --warmup 8 --no-cpu-baseline --no-secondary --no-roofline"
for src in pull copy device; do python bench.py $B --batch-source $src > $OUT/bench_$src.json 2> $OUT/bench_$src.err; done
python bench.py $B --dp-path > $OUT/bench_dp.json 2> $OUT/bench_dp.err
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python3 bench.py --steps 60 --warmup 8 --no-cpu-baseline --no-secondary --no-roofline > $OUT/bench_trace.json 2> $OUT/bench_trace.err
DB=$(ls $OUT/trace/*results.db $OUT/trace/*/*results.db 2>/dev/null | head -1)
python3 scripts/rocpd_gaps.py $DB --skip 0.6 --timeline 380 > $OUT/timeline.txt 2>&1
python3 scripts/rocpd_stats.py $DB --iters 72 --shapes gemm_nt > $OUT/kernel_stats.txt 2>&1
rm -rf $OUT/trace
python scripts/bf16_exceptions_ab.py monitor > $OUT/ab.log 2>&1; echo "ab rc=$?" >> $OUT/rc.txt
python scripts/bench_agents.py speaker --dtype bf16 > $OUT/speaker.log 2>&1; python scripts/bench_agents.py speaker --dtype fp32 >> $OUT/speaker.log 2>&1
python -c "
import sys, json; sys.path.insert(0,'.')
import torch, bench, vln_amd as vln
dev=torch.device('cuda:0'); store=bench.build_store(vln, dev, torch.bfloat16)
class A: batch_source='pull'; ride_gather='auto'; rollout_gather=False
tapes=[bench.make_tape(64,80,7,8,seed=2020+k,n_rows=store.N) for k in range(8)]
print(json.dumps(bench.secondary_host_in_loop(vln, dev, store, tapes, torch.bfloat16, A)))
" > $OUT/host_loop.log 2>&1
tail -3 $OUT/test.log; cat $OUT/bench_*.json | cut -c1-330; cat $OUT/host_loop.log $OUT/speaker.log | tail -5
