#!/bin/bash
# round 5, GPU call f: projected context A/B x3 after the load reordering + merged deferred dctx; kernel trace
OUT=gpurun_out/r5f; mkdir -p $OUT
B="--steps 100 --warmup 8 --no-cpu-baseline --no-secondary --no-roofline"
for i in 1 2 3; do
timeout -k 10 200 python bench.py $B > $OUT/bench_k$i.json 2> $OUT/bench_k$i.err || exit 1
timeout -k 10 200 python bench.py $B --no-project-context > $OUT/bench_nok$i.json 2> $OUT/bench_nok$i.err || exit 1
done
for f in $OUT/bench_*.json; do python -c "import json,sys; j=json.load(open(sys.argv[1])); print(sys.argv[1], j['ms_per_step'], j['config'].get('projected_context'))" $f; done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/trace -o k -- python3 $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 8 --no-cpu-baseline --no-secondary --no-roofline > $GRAFT_REPO_ROOT/$OUT/trace.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/rocpd_stats.py $OUT/trace/k_results.db --iters 58 > $OUT/kernel_stats.txt 2>&1; head -30 $OUT/kernel_stats.txt
