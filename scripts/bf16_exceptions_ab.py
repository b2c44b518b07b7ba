"""decide the bf16 exceptions by measurement.  For each set of decoder weight matrices streamed in fp32
instead of bf16 (EnvDropDecoder.fp32_weights): (a) the full-size EnvDrop parity test (B=64, 36 x 2176, H=512, 3 steps, dropout
on) against the UNROUNDED fp64 oracle in record-only mode -> which tensors exceed north_star's 1e-2, (b) ms per training
iteration of the headline bench.   python scripts/bf16_exceptions_ab.py"""
import json, os, subprocess, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
os.environ["VLN_PARITY_RECORD_ONLY"] = "1"
import torch
import vln_amd as vln
import parity
import test_hip_modules as T

SETS = [(), ("w_vin",), ("w_vin", "w_tin"), ("w_cat",), ("w_vin", "w_cat"), ("w_vin", "w_cat", "w_tin", "w_tout", "w_c")]
out = []
for fs in SETS:
    vln.EnvDropDecoder.default_fp32_weights = frozenset(fs)
    parity.RECORDS.clear()
    T._full_size_envdrop(vln, torch.bfloat16)
    recs = [r for r in parity.RECORDS if r["what"].startswith("bf16 unrounded")]
    over = sorted(((r["what"].replace("bf16 unrounded: ", ""), r["err"]) for r in recs if r["err"] >= 1e-2), key=lambda x: -x[1])
    worst = max(r["err"] for r in recs)
    # ms per iteration (graph mode, 60 steps) in a child process with the same class default
    code = ("import sys; sys.path.insert(0,'.'); import vln_amd as v; v.EnvDropDecoder.default_fp32_weights=frozenset(%r); "
            "import bench; sys.argv=['bench.py','--steps','60','--warmup','8','--no-cpu-baseline','--no-secondary','--no-roofline']; bench.main()" % (fs,))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    ms = json.loads(r.stdout.strip().splitlines()[-1])["ms_per_step"] if r.returncode == 0 else None
    row = dict(fp32_weights=list(fs), comparisons=len(recs), over_1e2=len(over), worst=worst, over=over[:12], ms_per_step=ms)
    out.append(row)
    print(json.dumps(row), flush=True)
vln.EnvDropDecoder.default_fp32_weights = frozenset()
