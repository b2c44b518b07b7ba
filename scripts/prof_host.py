"""Host-side profile of bench iterations: submit-vs-drain timing, per-phase host time, cProfile top entries.

    python scripts/prof_host.py [bf16|fp32]
"""
import cProfile, pstats, sys, time, io
sys.path.insert(0, '.')
import torch, bench
import vln_amd as vln
dev = torch.device('cuda:0')
dtype = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == 'bf16') else torch.float32
agent = bench.GpuAgent(vln, dev, dtype, 1, arena=True)
tape = bench.tape_to(bench.make_tape(64, 80, 7, 8, 2020), dev, store_dtype=dtype)
for _ in range(5): agent.iteration(tape)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): agent.iteration(tape)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"submit {1e3*(t1-t0)/20:.2f} ms/iter, drain {(t2-t1)*1e3:.2f} ms total, wall {(t2-t0)*1e3/20:.2f} ms/iter")


def phases(sync):
    """host time of each phase; with sync=True each phase is drained, so the numbers are GPU-inclusive"""
    acc = dict(zero=0., enc=0., dec=0., bwd=0., opt=0.)
    N = 10
    for _ in range(N):
        def lap(key, t):
            if sync: torch.cuda.synchronize()
            n = time.perf_counter(); acc[key] += n - t; return n
        t = time.perf_counter()
        agent.opt.zero_grad(); t = lap('zero', t)
        ctx, h_t, c_t = agent.enc(tape["tokens"], tape["lengths32"]); t = lap('enc', t)
        h_tilde, terms = h_t, []
        for s in tape["steps"]:
            img, cand, kw = agent.step_features(tape, s)
            logits, (h_t, c_t), h_tilde = agent.dec(s["angle"], img, cand, h_tilde, h_t, c_t, ctx, tape["seq_mask"], **kw)
            terms.append(vln.losses.masked_cross_entropy(logits, s["target"], s["cand_mask"], "sum"))
        loss = torch.stack(terms).sum() * bench.ML_WEIGHT / 64; t = lap('dec', t)
        loss.backward(); t = lap('bwd', t)
        agent.opt.allreduce(); agent.opt.step(); t = lap('opt', t)
        if not sync: torch.cuda.synchronize()
    print(("drained " if sync else "submit  ") + "  ".join(f"{k} {v/N*1e3:.2f} ms" for k, v in acc.items()))


phases(False); phases(True)
pr = cProfile.Profile(); pr.enable()
for _ in range(10): agent.iteration(tape)
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(32); print(s.getvalue()[:7000])
