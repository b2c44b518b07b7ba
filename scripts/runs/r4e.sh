#!/bin/bash
# round 4, GPU call e: full suite, chained steps A/B, memory-copy trace of the captured iteration, fused-cell probe
OUT=gpurun_out/r4e; mkdir -p $OUT
python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/test.log 2>&1; echo "pytest rc=$?" > $OUT/rc.txt
cp gpurun_out/parity_report.json $OUT/parity_report.json; cp gpurun_out/parity_summary.txt $OUT/parity_summary.txt
B="--steps 100 --warmup 8 --no-cpu-baseline --no-secondary --no-roofline"
for i in 1 2; do
python bench.py $B > $OUT/bench_chain$i.json 2> $OUT/bench_chain$i.err
python bench.py $B --no-chain > $OUT/bench_nochain$i.json 2> $OUT/bench_nochain$i.err
done
scripts/cell_probe > $OUT/cell_probe.txt 2>&1
export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --stats -d $OUT/trace -o trace -- python3 bench.py --steps 60 --warmup 8 --no-cpu-baseline --no-secondary --no-roofline > $OUT/bench_trace.json 2> $OUT/bench_trace.err
DB=$(ls $OUT/trace/*results.db $OUT/trace/*/*results.db 2>/dev/null | head -1)
python3 scripts/rocpd_gaps.py $DB --skip 0.6 --timeline 360 > $OUT/timeline.txt 2>&1
python3 - "$DB" > $OUT/memcopies.txt 2>&1 <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
print([t for t in tabs if 'copy' in t.lower() or 'memory' in t.lower()])
for t in tabs:
    if 'memory_cop' in t.lower() or t.lower() == 'memory_copies':
        cols = [r[1] for r in cur.execute(f"pragma table_info({t})")]
        print(t, cols)
        rows = list(cur.execute(f"select * from {t} order by start"))
        print(len(rows), "copies")
        for r in rows[-40:]:
            print(r)
PY
rm -rf $OUT/trace
tail -3 $OUT/test.log; cat $OUT/bench_*chain*.json | cut -c1-200; cat $OUT/cell_probe.txt
