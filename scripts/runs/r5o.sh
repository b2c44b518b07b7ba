#!/bin/bash
# round 5, GPU call o: the N > 1 path on one GPU (one-rank RCCL group): three segments vs collectives captured in the one graph vs the plain single graph
OUT=gpurun_out/r5o; mkdir -p $OUT
B="--steps 100 --warmup 8 --no-cpu-baseline --no-secondary --no-roofline"
for i in 1 2 3; do
timeout -k 10 200 python bench.py $B > $OUT/single$i.json 2> $OUT/single$i.err || exit 1
timeout -k 10 200 python bench.py $B --dp-path > $OUT/seg$i.json 2> $OUT/seg$i.err || exit 1
timeout -k 10 200 python bench.py $B --dp-path --dp-capture > $OUT/cap$i.json 2> $OUT/cap$i.err || { tail -5 $OUT/cap$i.err; }
timeout -k 10 200 python bench.py $B --no-ride-wgrads > $OUT/noride$i.json 2> $OUT/noride$i.err || exit 1
done
for f in $OUT/*.json; do python -c "import json,sys; j=json.load(open(sys.argv[1])); print(sys.argv[1], j['ms_per_step'], j['config'].get('iteration_graph'))" $f; done
