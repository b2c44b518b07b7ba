O=gpurun_out/r3u; mkdir -p $O
timeout -k 10 300 python3 -m pytest tests/test_hip_graphs.py -x -q > $O/graphs.txt 2>&1; echo rc=$?
tail -25 $O/graphs.txt
