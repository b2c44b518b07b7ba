set -e
O=gpurun_out/r3d; mkdir -p $O
export TMPDIR=/tmp
python3 -m pytest tests/test_hip_graphs.py -x -q > $O/graphs.txt 2>&1 || { tail -40 $O/graphs.txt; exit 1; }
tail -3 $O/graphs.txt
B="--steps 100 --warmup 8 --no-cpu-baseline --no-secondary --no-roofline"
for i in 1 2; do
python3 bench.py $B --iteration-graph off > $O/bench_eager_$i.json 2> $O/bench_eager_$i.err
python3 bench.py $B > $O/bench_graph_$i.json 2> $O/bench_graph_$i.err
python3 bench.py $B --rollout-gather > $O/bench_graph_rg_$i.json 2> $O/bench_graph_rg_$i.err
python3 bench.py $B --rollout-gather --gather-branch > $O/bench_graph_branch_$i.json 2> $O/bench_graph_branch_$i.err
done
grep -H ms_per_step $O/*.json | sed 's/"metric.*"ms_per_step"/ms_per_step/' | cut -c1-120
grep -h "host submit\|captured" $O/bench_graph_1.err
