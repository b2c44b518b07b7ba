#!/bin/bash
# Copy what one `scripts/run_pmc.sh + the bench line` GPU call produced (bench line, rocprofv3 kernel stats, the PMC passes) into profiles/ (tracked).
#   bash scripts/collect_profiles.sh gpurun_out/r5s
set -e
SRC=$1; P=profiles
cp $SRC/pmc/round6_pmc.json $P/round6_pmc.json
cp $SRC/pmc/gemm_nt_by_shape.txt $P/round6_gemm_nt_by_shape.txt
cp $SRC/pmc/stats.txt $P/round6_kernel_stats.txt
cp $SRC/bench_full.json $P/round6_bench_bf16.json
[ -s $SRC/timeline.txt ] && cp $SRC/timeline.txt $P/round6_timeline.txt
rm -rf $P/round6_pmc_d; mkdir -p $P/round6_pmc_d
cp $SRC/pmc/kernel_stats.csv $P/round6_pmc_d/
for d in FETCH_SIZE WRITE_SIZE TCC MFMA; do
  f=$(ls $SRC/pmc/$d/*counter_collection.csv $SRC/pmc/$d/*/*counter_collection.csv 2>/dev/null | head -1)
  gzip -c $f > $P/round6_pmc_d/${d}_counter_collection.csv.gz
done
ls -la $P/round6_pmc_d
for w in a2c_handshake monitor_dtype_fp32 monitor_dtype_bf16 follower_fused_only speaker; do
  [ -s $SRC/w_$w.stats.txt ] && cp $SRC/w_$w.stats.txt $P/round6_${w}_kernel_stats.txt
done
