O=gpurun_out/r3v; mkdir -p $O
python3 scripts/bench_agents.py monitor --steps 50 --warmup 10 > $O/mon_g.txt 2>&1
python3 scripts/bench_agents.py monitor --steps 50 --warmup 30 --no-graph > $O/mon_e.txt 2>&1
python3 scripts/bench_agents.py follower --steps 50 --warmup 10 > $O/fol_g.txt 2>&1
python3 scripts/bench_agents.py follower --steps 50 --warmup 30 --no-graph > $O/fol_e.txt 2>&1
grep -h "ms_per_iteration\|Error\|error" $O/*.txt | cut -c1-200
