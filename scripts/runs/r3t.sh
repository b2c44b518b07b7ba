O=gpurun_out/r3t; mkdir -p $O
timeout -k 10 500 python3 scripts/bf16_exceptions_ab.py > $O/ab.txt 2> $O/ab.err; echo rc=$?
cat $O/ab.txt; tail -5 $O/ab.err
