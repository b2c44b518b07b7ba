#!/usr/bin/env python3
"""gemm_nt by launch shape from rocprofv3 --pmc passes of bench.py: HBM-side traffic (FETCH_SIZE x 1 KiB x 2
[gfx950 wide-read correction] + WRITE_SIZE x 1 KiB, MI355X_MICROARCH.md HBM section), L2 hit rate (TCC_HIT_sum / (TCC_HIT_sum +
TCC_MISS_sum)), against the ALGORITHMIC bytes of the product (weights once + X + the split-K slabs it writes) and its duration.

    python scripts/pmc_by_shape.py <FETCH_SIZE dir> <WRITE_SIZE dir> <TCC dir> > profiles/round6_gemm_nt_by_shape.txt

A launch is identified by (weight type of the kernel template, grid size in threads); the EnvDrop headline's shapes are named below
(B = 64, H = 512, F = 2176, AE = 64, L = 80).  Shapes with the same workgroup count and weight type share a row."""
import collections, csv, glob, re, sys

# workgroups -> [(what, M, N, K)] ; the split-K factor is workgroups / ceil(N / 64) / ceil(M / 64)
SHAPES = {
    (8 * 4): [("text query / d(text query): H x H", 64, 512, 512)],
    (43 * 8): [("d xcat = dgates W_cat: 4H -> AE+F+H", 64, 2752, 2048)],
    (32 * 11): [("LSTM gates = xcat W_cat^T: AE+F+H -> 4H", 64, 2048, 2752)],
    (34 * 4): [("visual query: H -> F", 64, 2176, 512), ("d(visual query): F -> H (8 x 17)", 64, 512, 2176)],
    (8 * 8): [("linear_out: 2H -> H", 64, 512, 1024), ("d tcat: H -> 2H (16 x 4)", 64, 1024, 512)],
    (4 * 80): [("d embedding rows: 4Hd*2 -> E, M = L*B", 5120, 256, 2048)],
    (32 * 80): [("encoder input projection: E -> 4Hd*2, M = L*B", 5120, 2048, 256)],
    (34 * 7): [("rollout logits query: H -> F, M = T*B", 448, 2176, 512)],
    (8 * 6 * 7): [("rollout logit branch backward: F -> H, M = T*B", 448, 512, 2176)],
    (8 * 80): [("projected context K = ctx W_in: H -> H, M = L*B (round 5)", 5120, 512, 512)],
    (8 * 64): [("projected context K = ctx W_in on 80-row tiles (gemm_rows.h, round 5): H -> H, M = L*B", 5120, 512, 512)],
    (8 * 10): [("the rollout's text queries W_in hd_t on 48-row tiles (gemm_rows.h): H -> H, M = T*B (round 5)", 448, 512, 512)],
    (8 * 7): [("the rollout's text queries W_in hd_t for the context gradient: H -> H, M = T*B (round 5)", 448, 512, 512)],
    104: [("the rollout's text queries W_in hd_t on gemm_rows<3> tiles (vln_gemm_rows_tiling(448, 512) = 104 tiles): H -> H, M = T*B", 448, 512, 512)],
    (4 * 64): [("d embedding rows on gemm_rows<5> tiles: 4Hd*2 -> E, M = L*B", 5120, 256, 2048)],
}


def rows(d, counters):
    out = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "gemm_nt_kernel" not in n and "gemm_rows_kernel" not in n:
                continue
            wt = "f32s" if "f32s_raw" in n else ("bf16" if "unsigned short" in n else "f32")
            key = (wt, int(r["Grid_Size"]) // int(r["Workgroup_Size"]))
            c = r["Counter_Name"]
            if c in counters:
                a = out[key][c]
                a[0] += float(r["Counter_Value"]); a[1] += 1
                t = out[key]["_dur"]
                t[0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; t[1] += 1
    return out


fetch = rows(sys.argv[1], ("FETCH_SIZE",))
write = rows(sys.argv[2], ("WRITE_SIZE",))
tcc = rows(sys.argv[3], ("TCC_HIT_sum", "TCC_MISS_sum")) if len(sys.argv) > 3 else {}
print(f"{'weights':7s} {'wgs':>5s} {'launches':>8s} {'us (pmc run)':>12s} {'fetch MB':>9s} {'write MB':>9s} {'traffic MB':>10s} {'algo MB':>8s} {'ratio':>6s} {'L2 hit':>7s}  shape")
tot_t, tot_a = 0.0, 0.0
for key in sorted(set(fetch) | set(write), key=lambda k: (k[1], k[0])):
    f = fetch[key]["FETCH_SIZE"]; w = write[key]["WRITE_SIZE"]
    nf, nw = max(f[1], 1), max(w[1], 1)
    fb = f[0] / nf * 1024 * 2.0; wb = w[0] / nw * 1024
    dur = fetch[key]["_dur"]; us = dur[0] / max(dur[1], 1)
    names = SHAPES.get(key[1])
    if names is None:          # a launch shape this table does not know: reported as such, never priced against 0 algorithmic bytes
        print(f"{key[0]:7s} {key[1]:5d} {f[1]:8d} {us:12.2f} {fb / 1e6:9.2f} {wb / 1e6:9.2f} {(fb + wb) / 1e6:10.2f} {'n/a':>8s} {'n/a':>6s} {'':>7s}  (launch shape not in scripts/pmc_by_shape.py::SHAPES)")
        continue
    ws = 2 if key[0] == "bf16" else 4
    algo = []
    for what, M, N, K in names:
        nsplit = max(1, key[1] // max(1, ((N + 63) // 64) * ((M + 63) // 64)))
        algo.append(ws * N * K + 4 * M * K + 4 * M * N * nsplit)
    al = sum(algo) / len(algo)
    hit = ""
    if key in tcc:
        h, m = tcc[key]["TCC_HIT_sum"][0], tcc[key]["TCC_MISS_sum"][0]
        hit = f"{h / max(h + m, 1):.3f}"
    tot_t += (fb + wb) * f[1]; tot_a += al * f[1]
    print(f"{key[0]:7s} {key[1]:5d} {f[1]:8d} {us:12.2f} {fb / 1e6:9.2f} {wb / 1e6:9.2f} {(fb + wb) / 1e6:10.2f} {al / 1e6:8.2f} {(fb + wb) / max(al, 1):6.2f} {hit:>7s}  "
          + " | ".join(n[0] for n in names))
print(f"all gemm_nt launches: traffic / algorithmic (slabs counted as algorithmic output) = {tot_t / max(tot_a, 1):.2f}")
