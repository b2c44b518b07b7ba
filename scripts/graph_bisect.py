"""Which part of the captured iteration makes a trivial dependent kernel cost ~3.9 us inside it (1.9 us in a graph of its
own, scripts/torch_boundary_probe.py)?  For several sub-graphs of the real iteration: us per trivial launch =
(replay time with N trivial launches appended - replay time without) / N."""
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
store = bench.build_store(vln, dev, dtype, 2000)
tapes = [bench.tape_to(bench.make_tape(64, 80, 7, 8, seed=2020 + k, n_rows=store.N), dev, store=store) for k in range(2)]
live = bench.LiveBatch(tapes)
ag = bench.GpuAgent(vln, dev, dtype, 1, arena=True)
ag.clear_grads_in_step = True
ag.use_clock(store)
for k in range(4):
    ag.iteration(live.load(k))
torch.cuda.synchronize()
buf = torch.zeros(2, 64 * 512, device=dev)
N = 48


def trivial(n):
    if n:
        vln._lib.check(lib.vln_debug_trivial_chain(buf[0].data_ptr(), buf[1].data_ptr(), 64 * 512, n, 256, vln._lib.raw_stream()), "chain")


def timed(fn, reps=100):
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        fn()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


def variant(name, body):
    t0 = timed(lambda: body(0))
    t1 = timed(lambda: body(N))
    print(f"{name:58s} {t0:9.1f} us   + {N} trivial: {t1:9.1f} us   -> {(t1 - t0) / N:5.2f} us per trivial launch", flush=True)


tape = live.live
variant("trivial launches alone", lambda n: trivial(n))


def enc_only(n):
    with torch.no_grad():
        ag.enc(tape["tokens"], tape["lengths32"])
    trivial(n)


variant("encoder forward (no grad), trivial after", enc_only)


def enc_first(n):
    trivial(n)
    with torch.no_grad():
        ag.enc(tape["tokens"], tape["lengths32"])


variant("trivial first, then encoder forward", enc_first)


def opt_only(n):
    ag.opt.step(zero_grads=False)
    trivial(n)


variant("optimizer step, trivial after", opt_only)


def gemm_only(n):
    x = torch.zeros(64, 512, device=dev)
    w = ag.dec._shadow.t
    with torch.no_grad():
        for _ in range(8):
            vln.ops.linear_fwd(x, ag.dec.text_attn.linear_in.weight.detach())
    trivial(n)


variant("8 small fp32 GEMMs, trivial after", gemm_only)


def full(n):
    ag.probe_trivial = 0
    ag.iteration(tape)
    trivial(n)


variant("whole iteration, trivial after", full)


def full_mid(n):
    ag.probe_trivial = n // 2
    ag.iteration(tape)
    ag.probe_trivial = 0


variant("whole iteration, trivial inside (top + before backward)", full_mid)
