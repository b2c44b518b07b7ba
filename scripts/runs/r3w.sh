O=gpurun_out/r3w; mkdir -p $O
B="--steps 100 --warmup 8 --no-cpu-baseline --no-secondary --no-roofline"
run() { name=$1; shift; python3 bench.py $B "$@" > $O/$name.json 2> $O/$name.err; echo "$name $(grep -o '"ms_per_step": [0-9.]*' $O/$name.json)"; }
run base
run deep --tunable 5=4
run wide --tunable 1=3
run t256 --tunable 0=256
run t320 --tunable 0=320
run t448 --tunable 0=448
run t512 --tunable 0=512
run t640 --tunable 0=640
run base2
run nosplitattn --tunable 4=2
