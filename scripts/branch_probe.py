#!/usr/bin/env python3
"""Do two BRANCHES of a hipGraph run concurrently on this stack, and what does the fork / join cost?  Two independent chains of N
dependent skinny products each ([64, 512] x [512, 512]: launch-latency-bound, ~6 us), captured (a) on one stream, 2N launches in a
row, (b) as two branches (fork after a common head, join before a common tail).  ms per replay."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vln_amd as vln
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
w = [torch.randn(512, 512, device=dev).to(torch.bfloat16) for _ in range(2)]
x0 = torch.randn(64, 512, device=dev)


def chain(x, wi, n):
    for _ in range(n):
        x = vln.ops.linear_fwd(x, wi)
    return x


def one_stream():
    a = chain(x0, w[0], N)
    b = chain(x0, w[1], N)
    return a + b


side = torch.cuda.Stream()


def two_branches():
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        b = chain(x0, w[1], N)
    a = chain(x0, w[0], N)
    main.wait_stream(side)
    return a + b


for name, fn in (("one stream", one_stream), ("two branches", two_branches)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        g.replay()
    torch.cuda.synchronize()
    print(f"{name:14s} N={N}: {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms per replay ({2 * N} products)", flush=True)
