#!/bin/bash
# round 4, GPU call q: W_F32X (six-product fp32-grade gemm_nt) for the BN-MLP's forward Linear: ops + Self-Monitor parity tests, timing
OUT=gpurun_out/r4q; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_hip_ops.py tests/test_hip_full_size_agents.py tests/test_hip_agents.py tests/test_hip_graphs.py -m gpu -q -p no:cacheprovider > $OUT/test.log 2>&1; echo "pytest rc=$?" > $OUT/rc.txt
tail -4 $OUT/test.log | cut -c1-250; cat $OUT/rc.txt; grep -E "FAILED|Error" $OUT/test.log | head
for i in 1 2; do
python scripts/bench_agents.py monitor --steps 40 --warmup 20 --dtype bf16 > $OUT/mon_bf16_$i.json 2> $OUT/mon_bf16_$i.err
done
cat $OUT/mon_bf16_*.json
grep -n "monitor" gpurun_out/parity_summary.txt | cut -c1-220 | head -12
