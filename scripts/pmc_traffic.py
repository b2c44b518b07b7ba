"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into per-kernel HBM-side bytes per launch.

Units and gfx950 corrections per /opt/skills/guides/MI355X_MICROARCH.md §HBM: the counters are in KiB; on gfx950
FETCH_SIZE reports exactly HALF the bytes of wide (16 B/lane) coalesced reads -> doubled here; WRITE_SIZE is exact
for 16-B streaming stores.  Infinity-Cache hits are counted (fabric-side counters), so "traffic" is bytes that
left the XCD L2s, not necessarily DRAM.

    python scripts/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE bf16 > profiles/traffic.json
"""
import collections
import csv
import glob
import json
import re
import sys

NAMES = {"gemm_nt_kernel": "gemm_nt", "gemm_nt_n16_kernel": "gemm_nt", "gemm_rows_kernel": "gemm_nt", "gemm_tn_kernel": "gemm_tn", "gemm_tn_x3_kernel": "gemm_tn",
         "wgrad_packed_kernel": "gemm_tn", "wgrad_grouped_x3_kernel": "gemm_tn", "attn_dot_kernel": "attn_dot",
         "attn_wsum_kernel": "attn_wsum", "attn_bwd_kernel": "attn_bwd", "lstm_persist_fwd_kernel": "lstm_rec_fwd", "lstm_persist_g_fwd_kernel": "lstm_rec_fwd", "lstm_persist_g_bwd_kernel": "lstm_rec_bwd",
         "gather_step_prep_kernel": "gather_step",
         "lstm_persist_bwd_kernel": "lstm_rec_bwd", "lstm_rec_fwd_kernel": "lstm_rec_fwd", "lstm_rec_bwd_kernel": "lstm_rec_bwd",
         "feat_dropout_kernel": "feat_dropout", "lstm_pw_fwd_kernel": "lstm_pointwise", "reduce_epilogue_kernel": "reduce_epilogue",
         "wgrad_pack_kernel": "wgrad_pack", "wgrad_pack_gemm_kernel": "wgrad_pack", "gather_step_kernel": "gather_step"}


def kernel_key(name):
    m = re.search(r"vln::(\w+)", name)
    if not m:
        return None
    k = m.group(1)
    if k == "attn_fused_kernel":          # one-launch attention rows: forward counts as attn_wsum, backward as attn_bwd
        return "attn_bwd" if re.search(r",\s*true\s*>", name) else "attn_wsum"
    return NAMES.get(k)


def load(d):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            key = kernel_key(r["Kernel_Name"])
            if key is None:
                continue
            a = acc[key]
            a[0] += float(r["Counter_Value"]); a[1] += 1
    return acc


fetch, write = load(sys.argv[1]), load(sys.argv[2])
dtype = sys.argv[3] if len(sys.argv) > 3 else "bf16"
out = {}
for k in sorted(set(fetch) | set(write)):
    f = fetch[k][0] / max(fetch[k][1], 1) * 1024 * 2.0     # KiB -> bytes, x2 gfx950 wide-read correction
    w = write[k][0] / max(write[k][1], 1) * 1024
    out[k] = {"bytes_per_launch": round(f + w), "fetch_bytes_corrected": round(f), "write_bytes": round(w),
              "launches_sampled": fetch[k][1]}
print(json.dumps({dtype: out}, indent=1))
