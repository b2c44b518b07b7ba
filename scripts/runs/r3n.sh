set -e
O=gpurun_out/r3n; mkdir -p $O
timeout -k 10 300 python3 -m pytest tests/test_hip_ops.py -x -q -k "four_workgroups or one_launch" > $O/ops.txt 2>&1 || { tail -30 $O/ops.txt; exit 1; }
tail -2 $O/ops.txt
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q --deselect tests/test_hip_cfg3_cfg4.py > $O/gputests.txt 2>&1 || { tail -30 $O/gputests.txt; exit 1; }
tail -2 $O/gputests.txt
B="--steps 100 --warmup 8 --no-cpu-baseline --no-secondary --no-roofline"
python3 bench.py $B > $O/g1.json 2> $O/g1.err
python3 bench.py $B > $O/g2.json 2> $O/g2.err
grep -H -o '"ms_per_step": [0-9.]*' $O/*.json
