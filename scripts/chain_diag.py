"""Diagnostic: which gradients differ between chained and unchained decoder steps (one eager iteration, same batch, same masks)."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
import bench, vln_amd as vln

dev = torch.device("cuda:0")
out = {}
for chain in (False, True):
    torch.manual_seed(77)
    store = bench.build_store(vln, dev, torch.float32, n_rows=300, seed=5)
    tapes = [bench.tape_to(bench.make_tape(16, 24, 4, 6, seed=500 + k, n_rows=store.N), dev, store=store) for k in range(2)]
    live = bench.LiveBatch(tapes)
    torch.manual_seed(78)
    ag = bench.GpuAgent(vln, dev, torch.float32, 1, arena=True)
    ag.use_live(live)
    ag.dec.chain_steps = chain
    ag.clear_grads_in_step = False
    ag.enc.deterministic_embedding_grad = True
    ag.ride_gather = True
    ag.use_clock(store)
    loss = ag.iteration(live.load(0))
    torch.cuda.synchronize()
    g = {"enc." + n: p.grad.clone() for n, p in ag.enc.named_parameters()}
    g.update({"dec." + n: p.grad.clone() for n, p in ag.dec.named_parameters()})
    out[chain] = (loss.detach().clone(), g)
print("loss equal:", torch.equal(out[False][0], out[True][0]), float(out[False][0]), float(out[True][0]))
for n in out[False][1]:
    a, b = out[False][1][n], out[True][1][n]
    d = (a - b).abs().max().item()
    print(f"{n:40s} max|diff| {d:.3e}  max|ref| {a.abs().max().item():.3e}  {'EQUAL' if torch.equal(a, b) else 'DIFFERENT'}")
