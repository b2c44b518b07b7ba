set -e
O=gpurun_out/r3b; mkdir -p $O
export TMPDIR=/tmp
python3 -m pytest tests -m gpu -x -q > $O/gputests.txt 2>&1 || { tail -30 $O/gputests.txt; exit 1; }
tail -3 $O/gputests.txt
./scripts/boundary_probe 300 g > $O/probe_graph.txt 2>&1
cat $O/probe_graph.txt
rocprofv3 --kernel-trace -d $O/probe_trace -o p -- ./scripts/boundary_probe 40 g > $O/probe_traced.txt 2>&1
python3 scripts/rocpd_gaps.py $(ls $O/probe_trace/*results.db | head -1) --skip 0.2 > $O/probe_gaps.txt
head -12 $O/probe_gaps.txt
rocprofv3 --kernel-trace -d $O/bench_trace -o b -- python3 bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-secondary --no-roofline > $O/bench.json 2> $O/bench.err
python3 scripts/rocpd_gaps.py $(ls $O/bench_trace/*results.db | head -1) --skip 0.6 --timeline 400 > $O/bench_gaps.txt
head -50 $O/bench_gaps.txt
