#!/bin/bash
# round 5, GPU call ba: what the driver runs at round end -- smoke(), the default bench.py
OUT=gpurun_out/r5ba; mkdir -p $OUT
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout -k 10 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
cut -c1-420 $OUT/bench_default.json
