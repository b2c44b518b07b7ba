O=$GRAFT_REPO_ROOT/gpurun_out/r3z4; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/a2c -o t -- python3 $GRAFT_REPO_ROOT/scripts/bench_agents.py a2c --steps 20 --warmup 4 > $O/a2c.json 2> $O/a2c.err || echo fail
tail -1 $O/a2c.json | cut -c1-300
