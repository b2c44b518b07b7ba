#!/usr/bin/env python3
"""Where does a host turn inside ONE captured iteration spend its time?  (a) a toy graph of N trivial kernels with a host turn between
each pair (graphs.HandshakeIterationGraph): us per turn, with vln_host_wait and with vln_host_wait_fetch (4.6 KB pull); (b)
trainers.EnvDropHostLoopIteration at the headline's size: eager, one graph with the actions stored by a kernel, one graph with a D2H
memcpy node per step.      python scripts/hostloop_probe.py [--steps 20]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vln_amd as vln

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=20)
a = ap.parse_args()
dev = torch.device("cuda:0")
vln._lib.load()
out = {}

for fetch in (False, True):
    clock = vln.DeviceClock(dev)
    x = torch.zeros(4096, device=dev)
    n = 16
    mail = [torch.zeros(4608, dtype=torch.uint8).pin_memory() for _ in range(n)]
    dst = [torch.zeros(4608, dtype=torch.uint8, device=dev) for _ in range(n)]
    segs = [("graph", lambda: (clock.tick(), x.add_(1.0))[1])]
    for i in range(n):
        segs += [("host", (lambda: None), (mail[i], dst[i])) if fetch else ("host", lambda: None), ("graph", lambda: x.add_(1.0))]
    hg = vln.HandshakeIterationGraph(segs, clock).capture()
    ms = vln.trainers.time_iterations(hg.replay, 50, 5)
    out["toy_us_per_turn_" + ("wait_fetch" if fetch else "wait")] = round(ms * 1e3 / n, 2)

def cpu_stat():
    d = {}
    for f in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat"):
        try:
            for line in open(f):
                k, v = line.split()
                d[k] = int(v)
        except OSError:
            pass
    return d


for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try:
        out["cgroup " + f] = open(f).read().strip()
    except OSError:
        pass
import threading
out["threads_before"] = threading.active_count()
dtype = torch.bfloat16
store = vln.synthetic.build_store(dev, dtype, 10567)
cpu_tapes = [vln.synthetic.make_tape(64, 80, 7, 8, seed=2020 + k, n_rows=store.N) for k in range(4)]
for mode in ("eager", "kernel", "plainwait"):
    torch.manual_seed(2020)
    tapes = [vln.synthetic.tape_to(t, dev, store=store) for t in cpu_tapes]
    ls = vln.LiveSteps(tapes, dev)
    it = vln.trainers.EnvDropHostLoopIteration(dev, dtype, ls, store)
    it.action_store = "kernel"
    it.fetch_in_wait = mode != "plainwait"
    for k in range(4):
        it.iteration(k)
    run = it.iteration
    if mode != "eager":
        it.capture(warmup=0)
        run = it.replay
    k0 = [0]

    def one():
        run(k0[0]); k0[0] += 1
        if mode == "sync":
            torch.cuda.synchronize()
    c0 = cpu_stat()
    out[f"hostloop_ms_{mode}"] = round(vln.trainers.time_iterations(one, a.steps, 6), 3)
    c1 = cpu_stat()
    out[f"cpu_stat_delta_{mode}"] = {k: c1[k] - c0.get(k, 0) for k in c1 if c1[k] != c0.get(k, 0)}
    out[f"os_threads_{mode}"] = len(os.listdir("/proc/self/task"))
    if mode == "kernel":
        # back-to-back replays: wall time of each, and of its parts
        import gc
        gcs = []
        gc.callbacks.append(lambda phase, info: gcs.append((phase, info.get("generation"), round(time.perf_counter() * 1e6))))
        walls, parts = [], []
        g_replay = it.graph.graph.replay
        marks = []
        it.graph.graph = type("G", (), {"replay": staticmethod(lambda: (marks.append(("launch0", time.perf_counter())), g_replay(), marks.append(("launch1", time.perf_counter())))[1])})()
        def wrap(obj, name, tag):
            f = getattr(obj, name)
            def w(*a_, **k_):
                marks.append((tag + ":in", time.perf_counter()))
                r = f(*a_, **k_)
                marks.append((tag + ":out", time.perf_counter()))
                return r
            setattr(obj, name, w)
        wrap(it.feed, "select", "select"); wrap(it.feed, "launched", "launched"); wrap(it, "_await_action", "await")
        wrap(it.graph, "_await_ack", "ack"); wrap(it.graph.clock, "replayed", "clock")
        for _ in range(12):
            marks.clear()
            t0 = time.perf_counter()
            one()
            t1 = time.perf_counter()
            walls.append(round((t1 - t0) * 1e6))
            parts.append([(k, round((v - t0) * 1e6)) for k, v in marks])
        slow = max(range(12), key=lambda i_: walls[i_])
        out["slow_iteration_marks"] = parts[slow]
        torch.cuda.synchronize()
        out["back_to_back_wall_us"] = walls
        out["gc_events"] = [(p_, g_) for p_, g_, _ in gcs if p_ == "start"]
        gc.callbacks.clear()
        gc.collect(); gc.freeze()
        walls2 = []
        for _ in range(12):
            t0 = time.perf_counter(); one(); walls2.append(round((time.perf_counter() - t0) * 1e6))
        torch.cuda.synchronize()
        gc.unfreeze()
        out["back_to_back_wall_us_gc_frozen"] = walls2
        # when does each action reach the host inside one replay?
        stamps = []
        real = it.env_step
        it.env_step = lambda t, act: (stamps.append((t, time.perf_counter())), real(t, act))[1]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        one()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        out["hostloop_action_arrival_us"] = [round((ts - t0) * 1e6, 1) for _, ts in stamps]
        out["hostloop_replay_returned_us"] = round((t1 - t0) * 1e6, 1)
        out["hostloop_device_done_us"] = round((t2 - t0) * 1e6, 1)
        it.env_step = real
    out[f"hostloop_mismatches_{mode}"] = it.mismatches
# cfg3's handshake graph for comparison: per-replay wall times
tape = vln.synthetic.tape_to(vln.synthetic.make_tape(64, 80, 35, 8, 2020, n_rows=store.N), dev, store=store)
ag = vln.trainers.EnvDropA2CIteration(dev, dtype, tape, T_il=7, graph=True, read_actions="handshake")
for _ in range(3):
    ag.iteration()
run = ag.capture()
walls = []
for _ in range(14):
    t0 = time.perf_counter(); run(); walls.append(round((time.perf_counter() - t0) * 1e6))
torch.cuda.synchronize()
out["cfg3_handshake_wall_us"] = walls
print(json.dumps(out))
