O=gpurun_out/r3s; mkdir -p $O
timeout -k 10 300 python3 -m pytest tests/test_hip_dp_rccl.py -x -q -s > $O/dp.txt 2>&1; echo rc=$?
grep -h "recurrence beside\|passed\|failed\|Error" $O/dp.txt | head -20
