"""Host-side profile of bench iterations (cProfile) + submit-vs-sync timing."""
import cProfile, pstats, sys, time, io
sys.path.insert(0, '.')
import torch, bench
import vln_amd as vln
dev = torch.device('cuda:0')
dtype = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == 'bf16') else torch.float32
agent = bench.GpuAgent(vln, dev, dtype, 1)
tape = bench.tape_to(bench.make_tape(64, 80, 7, 8, 2020), dev)
for _ in range(3): agent.iteration(tape)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): agent.iteration(tape)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"submit {1e3*(t1-t0)/10:.2f} ms/iter, drain {(t2-t1)*1e3:.2f} ms total")
pr = cProfile.Profile(); pr.enable()
for _ in range(10): agent.iteration(tape)
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(28); print(s.getvalue()[:6000])
