set -e
O=gpurun_out/r3p; mkdir -p $O
B="--steps 100 --warmup 8 --no-cpu-baseline --no-secondary --no-roofline"
for T in 2 4 7 10; do
python3 bench.py $B --T $T > $O/T$T.json 2> $O/T$T.err
done
for L in 20 40 80; do
python3 bench.py $B --len $L > $O/L$L.json 2> $O/L$L.err
done
grep -H -o '"ms_per_step": [0-9.]*' $O/*.json
