#!/bin/bash
# round 5, GPU call am: batches sent ahead by an async H2D copy (HostBatchFeed prefetch) vs pulled over PCIe by the first launch; x decoder shadows riding
OUT=gpurun_out/r5am; mkdir -p $OUT
for rep in 1 2 3; do
  for v in "--batch-source pull --no-ride-shadows" "--batch-source push --no-ride-shadows" "--batch-source push"; do
    echo "headline [$v]: $(timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --no-roofline $v 2>$OUT/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")"
  done
done
tail -3 $OUT/err.txt
