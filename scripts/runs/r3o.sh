set -e
O=gpurun_out/r3o; mkdir -p $O
export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/trace -o b -- python3 bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-secondary --no-roofline > $O/bench.json 2> $O/bench.err
python3 scripts/rocpd_gaps.py $(ls $O/trace/*results.db | head -1) --skip 0.5 > $O/gaps.txt
head -32 $O/gaps.txt
