#!/usr/bin/env python3
"""Tall products: gemm_nt's 64-row tiles (tunable 12 = 1) against the 16-row-block tiling of gemm_rows.h, HIP-event timed.
    python scripts/rows_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vln_amd as vln

lib = vln._lib.load()
dev = torch.device("cuda:0")
shapes = [(1152, 1024, 2176, "BN-MLP forward"), (1152, 2176, 1024, "BN-MLP backward"), (5120, 256, 2048, "d embedding rows"),
          (5120, 2048, 256, "encoder input projection"), (5120, 512, 512, "projected context"), (576, 1024, 2176, "BN-MLP forward, B = 64")]
for M, N, K, what in shapes:
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) / K ** 0.5
    for name, wt, split in (("fp32", w, False), ("bf16", w.bfloat16(), False), ("f32s", w, True), ("f32x", w, "x6")):
        res = []
        for keep64 in (1, 0):
            lib.vln_set_tunable(12, keep64)
            for _ in range(3):
                vln.ops.linear_fwd(x, wt, split=split)
            ts = []
            for _ in range(20):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); vln.ops.linear_fwd(x, wt, split=split); b.record(); b.synchronize()
                ts.append(a.elapsed_time(b) * 1e3)
            ts.sort()
            res.append(ts[len(ts) // 2])
        lib.vln_set_tunable(12, 0)
        print(f"M={M:5d} N={N:5d} K={K:5d} {name}: 64-row tiles {res[0]:7.1f} us   row-block tiling {res[1]:7.1f} us   ({what})", flush=True)
