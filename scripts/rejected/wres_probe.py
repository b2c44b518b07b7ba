"""The encoder's input projection ([5120, 256] x [2048, 256]^T, bf16 weights) and its siblings: gemm_nt's tiles vs the
resident-weights kernel (gemm_wres.h; tunable 13), HIP-event timings of 200 back-to-back launches."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vln_amd as vln

lib = vln._lib.load()
dev = torch.device("cuda:0")
for M, N, K in ((5120, 2048, 256), (5120, 1024, 256), (2560, 2048, 256), (5120, 2048, 128)):
    x = torch.randn(M, K, device=dev); w = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16(); b = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev)
    res = []
    for keep in (1, 0):
        vln._lib.check(lib.vln_set_tunable(13, keep), "vln_set_tunable")
        for _ in range(5):
            vln.ops.linear_fwd(x, w, b, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            vln.ops.linear_fwd(x, w, b, out=out)
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 200 * 1e3)
    vln._lib.check(lib.vln_set_tunable(13, 0), "vln_set_tunable")
    print(f"[{M}, {K}] x [{N}, {K}]^T   gemm_nt tiles {res[0]:6.1f} us   resident weights {res[1]:6.1f} us   ({2.0 * M * N * K / res[1] / 1e6:.0f} TFLOP/s, {4.0 * M * N / res[1] / 1e6:.2f} TB/s of output)")
