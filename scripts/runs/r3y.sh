O=$GRAFT_REPO_ROOT/gpurun_out/r3w15; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 -m pytest tests/test_hip_graphs.py -q -k "bench" > $O/t1.txt 2>&1; echo rc=$?
tail -3 $O/t1.txt
timeout -k 10 300 python3 bench.py --steps 50 --warmup 8 --no-secondary --no-cpu-baseline --no-roofline > $O/b.json 2> $O/b.err; grep -o '"ms_per_step": [0-9.]*\|"iteration_graph": [a-z]*' $O/b.json | tr '\n' ' '
