"""Where the HOST time of one decoder step goes (the forward phase is host-bound at B=64).

    python scripts/host_breakdown.py
Each line: host microseconds per call, GPU left to run asynchronously (no sync inside the loops).
"""
import sys, time
sys.path.insert(0, '.')
import torch, bench
import vln_amd as vln
from vln_amd import _lib

dev = torch.device('cuda:0')
dtype = torch.bfloat16
agent = bench.GpuAgent(vln, dev, dtype, 1, arena=True)
tape = bench.tape_to(bench.make_tape(64, 80, 7, 8, 2020), dev, store_dtype=dtype)
for _ in range(5): agent.iteration(tape)
torch.cuda.synchronize()
lib = _lib.load()
acc = {}


def wrap(name):
    f = getattr(lib, name)
    def g(*a):
        t = time.perf_counter()
        r = f(*a)
        acc[name] = acc.get(name, 0.) + time.perf_counter() - t
        acc[name + "#"] = acc.get(name + "#", 0) + 1
        return r
    setattr(lib, name, g)


for n in ("vln_envdrop_step_fwd", "vln_envdrop_step_bwd", "vln_gather_step", "vln_wgrad_grouped", "vln_colsum_grouped", "vln_attn_dctx_deferred", "vln_masked_ce_fwd",
          "vln_masked_ce_bwd", "vln_lstm_seq_fwd", "vln_lstm_seq_bwd", "vln_linear_wgrad", "vln_colsum", "vln_linear_fwd",
          "vln_transpose_cast", "vln_cast_copy", "vln_rmsprop_clip_step", "vln_embed_fwd", "vln_embed_bwd"):
    if hasattr(lib, n):
        wrap(n)

N = 20
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N): agent.iteration(tape)
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"iteration submit {1e3*(t1-t0)/N:.2f} ms")
tot = 0.
for k in sorted(k for k in acc if not k.endswith('#')):
    per_iter = acc[k] / N * 1e6
    tot += per_iter
    print(f"  C call {k:26s} {acc[k + '#']/N:5.1f} calls/iter  {acc[k]/acc[k + '#']*1e6:7.1f} us/call  {per_iter:7.1f} us/iter")
print(f"  C calls total {tot:.0f} us/iter")

s = tape["steps"][0]
ctx, h_t, c_t = agent.enc(tape["tokens"], tape["lengths32"])
torch.cuda.synchronize()


def timeit(label, fn, n=200):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): fn()
    e = time.perf_counter() - t
    torch.cuda.synchronize()
    print(f"{label:46s} {e/n*1e6:7.1f} us/call")


timeit("step_features (2 gathers)", lambda: agent.step_features(tape, s))
img, cand, kw = agent.step_features(tape, s)
with torch.no_grad():
    timeit("dec.forward no_grad", lambda: agent.dec(s["angle"], img, cand, h_t, h_t, c_t, ctx, tape["seq_mask"], **kw))
hd, cd, cx = h_t.detach().requires_grad_(True), c_t.detach().requires_grad_(True), ctx.detach()
outs = []
timeit("dec.forward grad (graph kept)", lambda: outs.append(agent.dec(s["angle"], img, cand, hd, hd, cd, cx, tape["seq_mask"], **kw)), n=50)
logits = outs[0][0]
timeit("masked_cross_entropy(..., 'sum')", lambda: vln.losses.masked_cross_entropy(logits, s["target"], s["cand_mask"], "sum"))
timeit("torch.empty", lambda: torch.empty(64, 512, device=dev))
timeit("current_stream().cuda_stream", lambda: torch.cuda.current_stream().cuda_stream)
x = torch.empty(64, 512, device=dev)
timeit("tensor.sum()", lambda: x.sum())
timeit("data_ptr", lambda: x.data_ptr())
