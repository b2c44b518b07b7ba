#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (…_results.db): per-kernel totals and, with --shapes, per-grid breakdown.

    python scripts/rocpd_stats.py gpurun_out/x/prof/x_results.db --iters 130 [--shapes gemm_nt attn] [--csv out.csv]
"""
import argparse, sqlite3, csv

ap = argparse.ArgumentParser()
ap.add_argument("db")
ap.add_argument("--iters", type=float, default=1.0, help="iterations the run contained (per-iteration columns)")
ap.add_argument("--shapes", nargs="*", default=[])
ap.add_argument("--csv")
a = ap.parse_args()
cur = sqlite3.connect(a.db).cursor()
rows = list(cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
n = sum(r[1] for r in rows)
print(f"total kernel time {tot/1e6:.2f} ms, {n} launches; per iteration {tot/1e6/a.iters:.3f} ms, {n/a.iters:.1f} launches")
for r in rows[:50]:
    print(f"{r[0][:64]:64s} n/it={r[1]/a.iters:6.1f} us/it={r[2]/1e3/a.iters:8.1f} avg={r[3]/1e3:8.2f}us min={r[4]/1e3:7.2f} max={r[5]/1e3:8.2f} {100*r[2]/tot:5.1f}%")
if a.csv:
    with open(a.csv, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            w.writerow([r[0], r[1], r[2], f"{r[3]:.1f}", f"{100*r[2]/tot:.2f}", r[4], r[5]])
for pat in a.shapes:
    print(f"-- {pat}")
    q = ("select name, grid_x/workgroup_x, grid_y/workgroup_y, grid_z/workgroup_z, count(*), avg(end-start), min(end-start), "
         f"vgpr_count, lds_size from kernels where name like '%{pat}%' group by 1,2,3,4 order by count(*)*avg(end-start) desc")
    for r in list(cur.execute(q))[:16]:
        print(f"   {r[0][:28]:28s} grid=({r[1]},{r[2]},{r[3]}) n/it={r[4]/a.iters:5.1f} avg={r[5]/1e3:7.2f}us min={r[6]/1e3:6.2f} "
              f"us/it={r[4]*r[5]/a.iters/1e3:7.1f} vgpr={r[7]} lds={r[8]}")
