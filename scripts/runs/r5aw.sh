#!/bin/bash
# round 5, GPU call aw: smoke(), the data-parallel path on one GPU (in-process and under torch.distributed.run with one rank), default bench
OUT=gpurun_out/r5aw; mkdir -p $OUT
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
echo "dp-path: $(timeout -k 10 300 python bench.py --dp-path --steps 100 --warmup 10 --no-cpu-baseline --no-secondary --no-roofline 2>$OUT/dp.err | cut -c1-200)"
echo "torchrun 1 rank: $(timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 100 --warmup 10 --no-cpu-baseline --no-secondary --no-roofline 2>$OUT/tr.err | tail -1 | cut -c1-200)"
