"""What cross-stream synchronisation costs next to the captured iteration (why a side-stream feature prefetch lost 115 us per
iteration): replay loop + per iteration (a) nothing, (b) event record on main, (c) record on main + wait on an idle side stream
+ record there + main waits, (d) the same plus a trivial kernel on the side stream, (e) a 100 us streaming kernel on the side
stream with NO dependency on main at all.
Measured (MI355X, round 3): nothing 1.713 ms; event record 1.718; event round trip main -> side -> main 1.872 (+158 us with
nothing to wait for); independent 128 MB copy beside the iteration 1.746 (+33 us for ~50 us of streaming); the same round trip
with hipStreamWriteValue64 / hipStreamWaitValue64 on signal memory: 2.646 ms (+930 us: removed again)."""
import sys, time
sys.path.insert(0, ".")
import torch
import bench
import vln_amd as vln

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
lib = vln._lib.load()
dtype = torch.bfloat16
torch.manual_seed(2020)
store = bench.build_store(vln, dev, dtype, 10567)
tapes = [bench.tape_to(bench.make_tape(64, 80, 7, 8, seed=2020 + k, n_rows=store.N), dev, store=store) for k in range(4)]
live = bench.LiveBatch(tapes)
ag = bench.GpuAgent(vln, dev, dtype, 1, arena=True)
ag.clear_grads_in_step = True
ag.use_clock(store)
for k in range(4):
    ag.iteration(live.load(k))
torch.cuda.synchronize()
ag.capture(live.live)
side = torch.cuda.Stream()
main = torch.cuda.current_stream()
e1, e2 = torch.cuda.Event(), torch.cuda.Event()
junk = torch.zeros(64 << 20, device=dev)        # 256 MB: a copy of half of it = ~100 us of pure streaming
tiny = torch.zeros(1024, device=dev)


def loop(per_iter, n=200):
    for k in range(10):
        live.load(k); ag.replay(); per_iter()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(n):
        live.load(k); ag.replay(); per_iter()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def b():
    e1.record(main)


def c():
    e1.record(main); side.wait_event(e1); e2.record(side); main.wait_event(e2)


def d():
    e1.record(main); side.wait_event(e1)
    with torch.cuda.stream(side):
        tiny.add_(1.0)
    e2.record(side); main.wait_event(e2)


def e():
    with torch.cuda.stream(side):
        junk[:32 << 20].copy_(junk[32 << 20:])


def f():
    with torch.cuda.stream(side):
        tiny.add_(1.0)


for name, fn in (("nothing", lambda: None), ("event record on main", b), ("record + idle side stream waits + records + main waits", c),
                 ("the same with a trivial kernel on the side stream", d), ("independent 128 MB copy on the side stream, no events", e),
                 ("independent trivial kernel on the side stream, no events", f), ("nothing (again)", lambda: None)):
    print(f"{name:62s} {loop(fn):.3f} ms per iteration", flush=True)
