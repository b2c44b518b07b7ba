#!/bin/bash
# Copy what one `scripts/r4o.sh` GPU call produced (bench line, rocprofv3 kernel stats, the PMC passes) into profiles/ (tracked).
#   bash scripts/collect_profiles.sh gpurun_out/r4o
set -e
SRC=$1; P=profiles
cp $SRC/pmc/round4_pmc.json $P/round4_pmc.json
cp $SRC/pmc/gemm_nt_by_shape.txt $P/round4_gemm_nt_by_shape.txt
cp $SRC/pmc/stats.txt $P/round4_kernel_stats.txt
cp $SRC/bench_full.json $P/round4_bench_bf16.json
[ -s $SRC/timeline.txt ] && cp $SRC/timeline.txt $P/round4_timeline.txt
rm -rf $P/round4_pmc_d; mkdir -p $P/round4_pmc_d
cp $SRC/pmc/kernel_stats.csv $P/round4_pmc_d/
for d in FETCH_SIZE WRITE_SIZE TCC MFMA; do
  f=$(ls $SRC/pmc/$d/*counter_collection.csv $SRC/pmc/$d/*/*counter_collection.csv 2>/dev/null | head -1)
  gzip -c $f > $P/round4_pmc_d/${d}_counter_collection.csv.gz
done
ls -la $P/round4_pmc_d
