#!/usr/bin/env python3
"""Which Python frames issue the device-to-device copies / fills (aten::copy_, clone, contiguous, fill_, zero_) that remain inside one
bench iteration?  (rocprofv3 shows them as __amd_rocclr_copyBuffer / fill kernels; torch profiler with stacks names the caller.)
    python scripts/find_copies.py"""
import sys, collections
sys.path.insert(0, ".")
import torch
import bench
import vln_amd as vln
from torch.profiler import profile, ProfilerActivity

dev = torch.device("cuda:0")
dtype = torch.bfloat16
store = bench.build_store(vln, dev, dtype, n_rows=600, seed=5)
tapes = [bench.tape_to(bench.make_tape(64, 80, 7, 8, seed=500 + k, n_rows=store.N), dev, store=store) for k in range(3)]
live = bench.LiveBatch(tapes, source="pull")
ag = bench.GpuAgent(vln, dev, dtype, 1, arena=True)
ag.use_live(live)
ag.clear_grads_in_step = True
ag.ride_gather = True
ag.dec.ride_wgrads = True
ag.use_clock(store)
for k in range(6):
    ag.iteration(live.load(k))
torch.cuda.synchronize()
N = 2
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for k in range(N):
        ag.iteration(live.load(k))
    torch.cuda.synchronize()
want = ("aten::copy_", "aten::fill_", "aten::zero_", "aten::clone", "aten::contiguous", "aten::add_", "aten::add", "aten::mul", "aten::sum", "aten::cat")
by = collections.Counter()
for e in prof.events():
    if e.name in want:
        frames = [f for f in (e.stack or []) if "site-packages/torch" not in f and "dist-packages/torch" not in f and "<built-in" not in f][:4]
        by[(e.name, " <- ".join(frames))] += 1
for (name, st), n in sorted(by.items(), key=lambda t: -t[1]):
    print(f"{n / N:5.1f}/iter  {name:18s} {st}")
