import sys, subprocess
sys.argv = ["bench_agents.py", "none"]
sys.path.insert(0, "scripts"); sys.path.insert(0, ".")
import torch
exec(open("scripts/bench_agents.py").read().split("if args.which in")[0])
from torch.profiler import profile, ProfilerActivity
args.steps, args.warmup = 3, 3
def go():
    run_monitor()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    go()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=70))
print(vln.functional.GRAD_IN_PLACE_STATS)
