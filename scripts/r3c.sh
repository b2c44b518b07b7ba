set -e
O=gpurun_out/r3c; mkdir -p $O
export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/bench_trace -o b -- python3 bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-secondary --no-roofline > $O/bench.json 2> $O/bench.err
python3 scripts/rocpd_gaps.py $(ls $O/bench_trace/*results.db | head -1) --skip 0.6 --timeline 400 > $O/bench_gaps.txt
head -60 $O/bench_gaps.txt
VLN_PARITY_RECORD_ONLY=1 python3 -m pytest tests/test_hip_cfg3_cfg4.py -x -q > $O/cfg34.txt 2>&1 || true
tail -40 $O/cfg34.txt
