O=gpurun_out/r3r; mkdir -p $O
python3 scripts/bench_agents.py monitor --steps 30 --warmup 30 > $O/mon1.txt 2>&1
python3 scripts/bench_agents.py monitor --steps 30 --warmup 30 > $O/mon2.txt 2>&1
python3 scripts/bench_agents.py follower --steps 30 --warmup 30 > $O/fol.txt 2>&1
grep -h ms_per_iteration $O/mon1.txt $O/mon2.txt $O/fol.txt
