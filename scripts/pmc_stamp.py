#!/usr/bin/env python3
"""Merge the rocprofv3 --pmc passes of `bench.py` into profiles/round6_pmc.json, stamped with the hash of the kernel sources they
were taken on (bench.py refuses the figures when the sources have changed since):

    python scripts/pmc_stamp.py <dir with FETCH_SIZE pass> <dir with WRITE_SIZE pass> <MFMA pass counter_collection.csv> [dtype]

traffic[dtype][kernel] = HBM-side bytes per launch (FETCH_SIZE x 1 KiB x 2 [gfx950 wide-read correction] + WRITE_SIZE x 1 KiB,
scripts/pmc_traffic.py); mfma_util[dtype][kernel] = SQ_VALU_MFMA_BUSY_CYCLES / (chip cycles x 1024 SIMDs) (scripts/pmc_mfma.py),
under the names bench.py's per-kernel timers use."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

fetch, write, mfma_csv = sys.argv[1:4]
dtype = sys.argv[4] if len(sys.argv) > 4 else "bf16"
traffic = json.loads(subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "pmc_traffic.py"), fetch, write, dtype],
                                    capture_output=True, text=True, check=True).stdout)
mf = json.loads(subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "pmc_mfma.py"), mfma_csv], capture_output=True, text=True,
                               check=True).stdout)
names = {"gemm_nt_kernel": "gemm_nt", "wgrad_packed_kernel": "gemm_tn", "lstm_persist_g_fwd_kernel": "lstm_rec_fwd",
         "lstm_persist_bwd_kernel": "lstm_rec_bwd", "attn_fused_kernel": "attn_wsum"}
util = {}
for k, v in mf.items():
    util.setdefault(names.get(k, k), v["mfma_util"])
out = {"csrc_sha": bench.csrc_sha(), "traffic": traffic, "mfma_util": {dtype: util}, "mfma_detail": mf}
path = os.path.join(ROOT, "profiles", "round6_pmc.json")
json.dump(out, open(path, "w"), indent=1)
print(f"wrote {path} for kernel sources {out['csrc_sha']}")
