#!/usr/bin/env python3
"""Per-kernel MFMA utilisation from a rocprofv3 `--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE` pass.

    python scripts/pmc_mfma.py profiles/round1_pmc_c/MFMA_BUSY_counter_collection.csv > profiles/round1_mfma_util.json

Per dispatch: SQ_VALU_MFMA_BUSY_CYCLES = cycles a SIMD's MFMA pipe was busy, summed over all SIMDs (16 per
v_mfma_f32_16x16x32_bf16: the weight-gradient launch's 23.3 GFLOP = 1.42 M MFMAs give 19 cycles each here);
GRBM_GUI_ACTIVE = cycles the dispatch kept the chip busy, summed over the 8 XCDs (MI355X_MICROARCH.md, DVFS note).
util = MFMA busy cycles / (chip cycles x 256 CUs x 4 SIMDs) = the fraction of the dense MFMA peak's issue slots used -- for a
launch that occupies 64 of 256 CUs at most 0.25.  Reported for the kernels that issue MFMAs at all.
"""
import collections, csv, json, re, sys

acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"vln::(\w+)", r["Kernel_Name"])
    if not m:
        continue
    k = m.group(1)
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        cnt[k] += 1
        acc[k]["ns"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
out = {}
for k, c in acc.items():
    mf, gui = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), c.get("GRBM_GUI_ACTIVE", 0.0)
    if mf <= 0 or gui <= 0:
        continue
    chip_cycles = gui / 8.0
    out[k] = {"launches": cnt[k], "avg_us": round(c["ns"] / cnt[k] / 1e3, 2), "mfma_busy_cycles_per_launch": round(mf / cnt[k]),
              "chip_cycles_per_launch": round(chip_cycles / cnt[k]), "sq_busy_cycles_per_launch": round(c.get("SQ_BUSY_CYCLES", 0.0) / cnt[k]),
              "mfma_util": round(mf / (chip_cycles * 1024.0), 4)}
print(json.dumps(dict(sorted(out.items(), key=lambda kv: -kv[1]["mfma_busy_cycles_per_launch"] * kv[1]["launches"])), indent=1))
