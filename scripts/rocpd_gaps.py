#!/usr/bin/env python3
"""Where the wall time of a launch stream goes, from a rocprofv3 rocpd database: for every kernel (ordered by start time) its
own duration (end - start) and the GAP in front of it (its start - the previous kernel's end; negative = overlap), summed per
kernel name, plus a time line of one steady-state window.

    python scripts/rocpd_gaps.py x_results.db [--skip 0.5] [--timeline 200]
"""
import argparse
import sqlite3
from collections import defaultdict

ap = argparse.ArgumentParser()
ap.add_argument("db")
ap.add_argument("--skip", type=float, default=0.5, help="fraction of the run to skip (warm-up, captures)")
ap.add_argument("--timeline", type=int, default=0, help="print this many consecutive launches of the steady state")
ap.add_argument("--big-gap-us", type=float, default=20.0, help="gaps above this are host stalls / sync points: listed apart")
a = ap.parse_args()
cur = sqlite3.connect(a.db).cursor()
rows = list(cur.execute("select name, start, end, grid_x/workgroup_x, grid_y/workgroup_y, grid_z/workgroup_z from kernels order by start"))
rows = rows[int(len(rows) * a.skip):]
dur, gap, cnt = defaultdict(float), defaultdict(float), defaultdict(int)
big = 0.0
nbig = 0
for i in range(1, len(rows)):
    n, s, e = rows[i][0], rows[i][1], rows[i][2]
    g = (s - rows[i - 1][2]) / 1e3
    key = n.split("(")[0][-48:]
    if g > a.big_gap_us:
        big += g
        nbig += 1
        g = 0.0
    dur[key] += (e - s) / 1e3
    gap[key] += g
    cnt[key] += 1
wall = (rows[-1][2] - rows[0][1]) / 1e3
td, tg = sum(dur.values()), sum(gap.values())
print(f"{len(rows)} launches, wall {wall:.1f} us: kernel durations {td:.1f} us ({100 * td / wall:.1f} %), gaps {tg:.1f} us "
      f"({100 * tg / wall:.1f} %), {nbig} gaps > {a.big_gap_us} us = {big:.1f} us")
print(f"{'kernel':48s} {'n':>6s} {'dur us':>8s} {'gap us':>8s} {'dur+gap':>8s}   share")
for k in sorted(dur, key=lambda k: -(dur[k] + gap[k])):
    print(f"{k:48s} {cnt[k]:6d} {dur[k] / cnt[k]:8.2f} {gap[k] / cnt[k]:8.2f} {(dur[k] + gap[k]) / cnt[k]:8.2f}   {100 * (dur[k] + gap[k]) / wall:5.1f} %")
if a.timeline:
    t0 = rows[0][1]
    print("\n   start us   dur us   gap us  grid            kernel")
    for i in range(1, min(len(rows), a.timeline + 1)):
        n, s, e, gx, gy, gz = rows[i]
        print(f"{(s - t0) / 1e3:11.2f} {(e - s) / 1e3:8.2f} {(s - rows[i - 1][2]) / 1e3:8.2f}  ({gx},{gy},{gz})".ljust(48) + n.split("(")[0][-60:])
