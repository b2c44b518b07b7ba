"""cProfile of the Self-Monitor iteration's host side (scripts/bench_agents.py's monitor workload)."""
import cProfile, pstats, io, sys
sys.path.insert(0, '.')
sys.argv = [sys.argv[0], "none"]
import importlib.util
spec = importlib.util.spec_from_file_location("ba", "scripts/bench_agents.py")
ba = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ba)
import torch
ba.args.steps, ba.args.warmup = 10, 4
pr = cProfile.Profile()
orig_timed = ba.timed
def timed(fn):
    for _ in range(4): fn()
    torch.cuda.synchronize()
    pr.enable()
    for _ in range(10): fn()
    pr.disable()
    torch.cuda.synchronize()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(45); print(s.getvalue()[:9000])
    return 1.0
ba.timed = timed
ba.run_monitor()
