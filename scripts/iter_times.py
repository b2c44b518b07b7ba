import sys, time, gc
sys.path.insert(0, '.')
import torch, bench
import vln_amd as vln
dev = torch.device('cuda:0')
agent = bench.GpuAgent(vln, dev, torch.bfloat16, 1, arena=(len(sys.argv) > 1 and sys.argv[1] == 'arena'))
tape = bench.tape_to(bench.make_tape(64, 80, 7, 8, 2020), dev, store_dtype=torch.bfloat16)
agent.iteration(tape); torch.cuda.synchronize()
if len(sys.argv) > 2 and sys.argv[2] == 'nogc': gc.disable()
ts = [time.perf_counter()]
mem = []
for i in range(40):
    agent.iteration(tape)
    ts.append(time.perf_counter())
    mem.append(torch.cuda.memory_reserved() >> 20)
torch.cuda.synchronize()
print(" ".join(f"{(b - a) * 1e3:.1f}" for a, b in zip(ts, ts[1:])))
print("reserved MB:", mem[::4], "gc counts", gc.get_count(), gc.get_stats()[2])
