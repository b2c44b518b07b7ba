O=gpurun_out/r3x; mkdir -p $O
timeout -k 10 300 python3 -m pytest tests/test_hip_staging.py -x -q -k "riding or rollout_gather" > $O/t.txt 2>&1; echo rc=$?
tail -5 $O/t.txt
B="--steps 100 --warmup 8 --no-cpu-baseline --no-secondary --no-roofline"
run() { name=$1; shift; timeout -k 10 200 python3 bench.py $B "$@" > $O/$name.json 2> $O/$name.err; echo "$name $(grep -o '"ms_per_step": [0-9.]*' $O/$name.json)"; }
run base
run ride --ride-gather on
run rollout --rollout-gather
run base2
run ride2 --ride-gather on
