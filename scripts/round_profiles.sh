#!/bin/bash
# Everything profiles/roundN_* holds, in one GPU call (run from the repo root on the GPU box):
#   bash scripts/round_profiles.sh gpurun_out/r6p      then, back home:  bash scripts/collect_profiles.sh gpurun_out/r6p
# 1. the rocprofv3 --pmc passes + kernel-trace stats of bench.py (scripts/run_pmc.sh), 2. the full bench line (it quotes the PMC
# figures only if they were taken on these kernel sources), 3. a kernel timeline of one captured iteration, 4. kernel stats of the
# other BASELINE workloads (cfg3 IL + A2C, Self-Monitor fp32 / bf16, Follower, speaker).
set -e
OUT=$1
mkdir -p $OUT
export TMPDIR=/tmp
bash scripts/run_pmc.sh $OUT/pmc
python3 bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err
rocprofv3 --kernel-trace -d $OUT/tl -o tl -- python3 bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-secondary --no-roofline > /dev/null 2> $OUT/tl.err
python3 scripts/rocpd_gaps.py $(ls $OUT/tl/*results.db | head -1) --skip 0.7 --timeline 300 > $OUT/timeline.txt 2>&1 || true
rm -rf $OUT/tl
for w in "a2c --handshake" "monitor --dtype fp32" "monitor --dtype bf16" "follower --fused-only" "speaker"; do
  tag=$(echo $w | tr ' -' '__' | tr -s '_')
  rocprofv3 --kernel-trace --stats -d $OUT/w_$tag -o t -- python3 scripts/bench_agents.py $w --steps 30 --warmup 8 > $OUT/w_$tag.json 2> $OUT/w_$tag.err
  python3 scripts/rocpd_stats.py $(ls $OUT/w_$tag/*results.db | head -1) --iters $(python3 -c "print(41 if 'a2c' in '$w' else 38)") > $OUT/w_$tag.stats.txt
  rm -rf $OUT/w_$tag
done
ls -la $OUT
