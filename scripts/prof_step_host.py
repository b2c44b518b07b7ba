"""cProfile of the host side of the UNCHANGED-CALLER iteration (the reference's loop on the drop-in modules: feature tensors handed
in every step, masked_fill + per-step CE, no arena, no deferred logits, no graphs; bench.py's `dropin_unchanged_caller`): where the
host-bound 2.0-2.6 ms per iteration go.      python scripts/prof_step_host.py [arena]"""
import cProfile, pstats, io, sys, time
import ctypes as C
sys.path.insert(0, '.')
import torch
import vln_amd as vln
dev = torch.device('cuda:0')
arena = len(sys.argv) > 1 and sys.argv[1] == "arena"
torch.manual_seed(2020)
agent = vln.trainers.EnvDropILIteration(dev, torch.bfloat16, 1, arena=arena, rollout_ce=arena)
agent.clear_grads_in_step = True
if "stepgraphs" in sys.argv:      # per-step hipGraphs WITHOUT an arena: relies on the caching allocator handing back the same addresses
    agent.dec.step_graphs = True
tape = vln.synthetic.tape_to(vln.synthetic.make_tape(64, 80, 7, 8, 2020), dev)
for _ in range(6): agent.iteration(tape)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30): agent.iteration(tape)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
gs = (C.c_longlong * 3)()
vln._lib.load().vln_graph_stats(gs)
print("graph stats [replays, captures, chains paused]:", list(gs))
print(f"30 iterations: host submit {(t1 - t0) / 30 * 1e3:.3f} ms each, with the device {(t2 - t0) / 30 * 1e3:.3f} ms each")
pr = cProfile.Profile()
N = 30
pr.enable()
for _ in range(N): agent.iteration(tape)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(40); print(s.getvalue()[:12000])
