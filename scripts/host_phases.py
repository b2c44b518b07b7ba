"""Host submit time of the phases of a bench iteration (arena + step plans active, nothing synchronised inside).
    python scripts/host_phases.py"""
import sys, time
sys.path.insert(0, '.')
import torch, bench
import vln_amd as vln
dev = torch.device('cuda:0')
agent = bench.GpuAgent(vln, dev, torch.bfloat16, 1, arena=True)
tape = bench.tape_to(bench.make_tape(64, 80, 7, 8, 2020), dev, store_dtype=torch.bfloat16)
for _ in range(8): agent.iteration(tape)
torch.cuda.synchronize()
acc = {}
def lap(k, t):
    n = time.perf_counter(); acc[k] = acc.get(k, 0.) + n - t; return n
N = 40
for _ in range(N):
    vln.ops.set_arena(agent.arena); agent.arena.begin()
    t = time.perf_counter()
    agent.opt.zero_grad(); t = lap('zero_grad', t)
    ctx, h_t, c_t = agent.enc(tape["tokens"], tape["lengths32"]); t = lap('encoder fwd', t)
    h_tilde, terms = h_t, []
    for s in tape["steps"]:
        img, cand, kw = agent.step_features(tape, s); t = lap('gather x7', t)
        logits, (h_t, c_t), h_tilde = agent.dec(s["angle"], img, cand, h_tilde, h_t, c_t, ctx, tape["seq_mask"], **kw); t = lap('decoder fwd x7', t)
        terms.append(vln.losses.masked_cross_entropy(logits, s["target"], s["cand_mask"], "sum")); t = lap('CE x7', t)
    loss = torch.stack(terms).sum() * bench.ML_WEIGHT / 64; t = lap('loss glue', t)
    loss.backward(); t = lap('backward', t)
    agent.opt.allreduce(); agent.opt.step(); t = lap('optimizer', t)
    vln.ops.set_arena(None)
torch.cuda.synchronize()
tot = sum(acc.values())
for k, v in acc.items():
    print(f"{k:18s} {v / N * 1e3:7.3f} ms/iter")
print(f"{'total submit':18s} {tot / N * 1e3:7.3f} ms/iter   plan hits {agent.dec.plan_hits}")
