"""torch.profiler view of the Self-Monitor workload (which aten ops the Python glue adds around the C calls)."""
import sys
sys.path.insert(0, "scripts"); sys.path.insert(0, ".")
import torch
import bench_agents as W
from torch.profiler import profile, ProfilerActivity

W.configure(steps=3, warmup=3, dtype="bf16", arena=False)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
    W.run_monitor()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=40, max_name_column_width=60))
print(W.vln.functional.GRAD_IN_PLACE_STATS)
