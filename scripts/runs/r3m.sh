set -e
O=gpurun_out/r3m; mkdir -p $O
export TMPDIR=/tmp
python3 -m pytest tests -m gpu -x -q --deselect tests/test_hip_cfg3_cfg4.py > $O/gputests.txt 2>&1 || { tail -30 $O/gputests.txt; exit 1; }
tail -2 $O/gputests.txt
B="--steps 100 --warmup 8 --no-cpu-baseline --no-secondary --no-roofline"
python3 bench.py $B > $O/g1.json 2> $O/g1.err
python3 bench.py $B > $O/g2.json 2> $O/g2.err
grep -H -o '"ms_per_step": [0-9.]*' $O/*.json
rocprofv3 --kernel-trace -d $O/trace -o b -- python3 bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-secondary --no-roofline > $O/bench.json 2> $O/bench.err
python3 scripts/rocpd_gaps.py $(ls $O/trace/*results.db | head -1) --skip 0.5 > $O/gaps.txt
grep -n "gather\|tm_to_bm\|bm_to_tm\|embed" $O/gaps.txt
