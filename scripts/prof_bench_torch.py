#!/usr/bin/env python3
"""Which torch (aten) ops still run inside one bench iteration?  (torch profiler, CPU + device activity.)"""
import sys
sys.path.insert(0, ".")
import torch
import bench
import vln_amd as vln
from torch.profiler import profile, ProfilerActivity

dev = torch.device("cuda:0")
tape = bench.tape_to(bench.make_tape(64, 80, 7, 8, seed=2020), dev, store_dtype=torch.bfloat16)
ag = bench.GpuAgent(vln, dev, torch.bfloat16, 1, arena=True)
for _ in range(8):
    ag.iteration(tape)
torch.cuda.synchronize()
N = 4
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(N):
        ag.iteration(tape)
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages() if e.key.startswith("aten::") or "Memcpy" in e.key or "Memset" in e.key]
rows.sort(key=lambda e: -e.count)
for e in rows[:30]:
    print(f"{e.key:40s} calls/iter={e.count / N:6.1f} cpu_us/iter={e.cpu_time_total / N:8.1f} dev_us/iter={e.device_time_total / N:8.1f}")
