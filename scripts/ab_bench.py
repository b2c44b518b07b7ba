"""Interleaved A/B timing of training-iteration variants in ONE process (cdna guide rule 24: separate invocations and
separate boxes differ by >10 %).  Usage: python scripts/ab_bench.py [rounds]"""
import statistics
import sys
import time

sys.path.insert(0, ".")
import torch
import bench
import vln_amd as vln

dev = torch.device("cuda:0")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
tape = bench.tape_to(bench.make_tape(64, 80, 7, 8, 2020), dev)
lib = vln._lib.load()


def make(name):
    torch.manual_seed(0)
    ag = bench.GpuAgent(vln, dev, torch.bfloat16, 1)
    it = ag.iteration
    if name == "prepare":
        def it2(t, _it=it, _ag=ag):
            r = _it(t); _ag.dec.prepare(); return r
        return ag, it2
    if name == "overlap_wgrads":
        ag.dec.overlap_wgrads = True
    return ag, it


variants = {n: make(n) for n in ("base", "prepare", "overlap_wgrads", "per_step_lstm", "no_graphs")}
times = {n: [] for n in variants}
for n, (ag, it) in variants.items():
    for _ in range(3):
        it(tape)
torch.cuda.synchronize()
for r in range(rounds):
    for n, (ag, it) in variants.items():
        lib.vln_set_persistent(0 if n == "per_step_lstm" else 1)
        lib.vln_set_graphs(0 if n == "no_graphs" else 1)
        if n == "no_graphs":
            lib.vln_set_persistent(0)
        it(tape)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            it(tape)
        torch.cuda.synchronize()
        times[n].append((time.perf_counter() - t0) / 10 * 1e3)
lib.vln_set_persistent(1); lib.vln_set_graphs(1)
for n, v in times.items():
    print(f"{n:16s} median {statistics.median(v):.3f} ms  min {min(v):.3f}  max {max(v):.3f}")
