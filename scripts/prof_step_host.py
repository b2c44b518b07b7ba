"""cProfile of the decoder step's host side (forward with grad + backward), 7-step rollouts repeated.
    python scripts/prof_step_host.py"""
import cProfile, pstats, io, sys, time
sys.path.insert(0, '.')
import torch, bench
import vln_amd as vln
dev = torch.device('cuda:0')
agent = vln.trainers.EnvDropILIteration(dev, torch.bfloat16, 1, arena=True)
tape = vln.synthetic.tape_to(vln.synthetic.make_tape(64, 80, 7, 8, 2020), dev, store_dtype=torch.bfloat16)
for _ in range(5): agent.iteration(tape)
torch.cuda.synchronize()
pr = cProfile.Profile()
N = 30
pr.enable()
for _ in range(N): agent.iteration(tape)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(45); print(s.getvalue()[:9000])
