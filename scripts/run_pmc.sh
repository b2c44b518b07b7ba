#!/bin/bash
# The three rocprofv3 --pmc passes of bench.py (separate runs, --kernel-trace only: MI355X_MICROARCH.md, HBM section) + the
# kernel-trace stats run; run on the GPU box from the repo root:  bash scripts/run_pmc.sh gpurun_out/pmc [extra bench args]
set -e
OUT=$1; shift
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 30 --warmup 4 --no-cpu-baseline --no-secondary --no-roofline $@"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python3 bench.py --steps 100 --warmup 8 --no-cpu-baseline --no-secondary --no-roofline "$@" > $OUT/bench_trace.json 2> $OUT/bench_trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/FETCH_SIZE -o p -- python3 bench.py $ARGS > /dev/null 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/WRITE_SIZE -o p -- python3 bench.py $ARGS > /dev/null 2> $OUT/write.err
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/TCC -o p -- python3 bench.py $ARGS > /dev/null 2> $OUT/tcc.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/MFMA -o p -- python3 bench.py $ARGS > /dev/null 2> $OUT/mfma.err
python3 scripts/rocpd_stats.py $(ls $OUT/trace/*results.db | head -1) --iters 112 --shapes gemm_nt attn_fused --csv $OUT/kernel_stats.csv > $OUT/stats.txt
python3 scripts/pmc_stamp.py $OUT/FETCH_SIZE $OUT/WRITE_SIZE $(ls $OUT/MFMA/*counter_collection.csv $OUT/MFMA/*/*counter_collection.csv 2>/dev/null | head -1) bf16
cp profiles/round6_pmc.json $OUT/round6_pmc.json
python3 scripts/pmc_by_shape.py $OUT/FETCH_SIZE $OUT/WRITE_SIZE $OUT/TCC > $OUT/gemm_nt_by_shape.txt 2>&1
head -30 $OUT/stats.txt
