"""Decide the bf16 mode's fp32-streamed matrices by measurement.  For each set of weight matrices
streamed in fp32 instead of bf16, the BASELINE-size parity test of the agent runs against the UNROUNDED fp64 oracle in
record-only mode -> which tensors exceed north_star's 1e-2 -- and the ms per training iteration is timed:

  envdrop   EnvDropDecoder.fp32_weights, tests/test_hip_modules.py::_full_size_envdrop (cfg1, B=64, 3 steps, dropout on) + bench.py
  cfg3      the same sets through tests/test_hip_cfg3_cfg4.py::_iteration (IL T=7 + A2C T=35, incl. the critic's gradients)
  monitor   MonitorDecoder.fp32_weights, tests/test_hip_full_size_agents.py::_monitor_full (cfg2, B=128) + scripts/bench_agents.py

    python scripts/bf16_exceptions_ab.py [envdrop] [cfg3] [monitor]      (default: all three); one JSON line per row
"""
import json, os, subprocess, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests"); sys.path.insert(0, "scripts")
os.environ["VLN_PARITY_RECORD_ONLY"] = "1"
import torch
import vln_amd as vln
import parity

which = [a for a in sys.argv[1:] if not a.startswith("-")] or ["envdrop", "cfg3", "monitor"]
SMOKE = ("loss",)                  # cancelling-sum scalars of the tests: relative error not meaningful


def row(tag, fs, prefix="bf16 unrounded: ", **extra):
    recs = [r for r in parity.RECORDS if r["what"].startswith(prefix)]
    names = lambda r: r["what"].replace(prefix, "")
    over = sorted(((names(r), round(r["err"], 5)) for r in recs if r["err"] >= 1e-2 and names(r) not in SMOKE), key=lambda x: -x[1])
    worst = max((r["err"] for r in recs if names(r) not in SMOKE), default=0.0)
    top = sorted(((names(r), round(r["err"], 5)) for r in recs if names(r) not in SMOKE), key=lambda x: -x[1])[:5]
    out = dict(agent=tag, fp32_weights=list(fs), comparisons=len(recs), over_1e2=len(over), worst=round(worst, 5), over=over[:14], top5=top, **extra)
    print(json.dumps(out), flush=True)
    return out


def child_ms(code):
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    if r.returncode != 0:
        return None
    return json.loads(r.stdout.strip().splitlines()[-1])


ENV_SETS = [(), ("w_vin",), ("w_vin", "w_tin"), ("w_cat",), ("w_vin", "w_tin", "w_cat"), ("w_vin", "w_cat", "w_tin", "w_tout", "w_c")]
default_env = vln.EnvDropDecoder.default_fp32_weights
if "envdrop" in which:
    import test_hip_modules as T
    for fs in ENV_SETS:
        vln.EnvDropDecoder.default_fp32_weights = frozenset(fs)
        parity.RECORDS.clear()
        T._full_size_envdrop(vln, torch.bfloat16)
        code = ("import sys; sys.path.insert(0,'.'); import vln_amd as v; v.EnvDropDecoder.default_fp32_weights=frozenset(%r); "
                "import bench; sys.argv=['bench.py','--steps','60','--warmup','8','--no-cpu-baseline','--no-secondary','--no-roofline']; bench.main()" % (fs,))
        j = child_ms(code)
        row("envdrop_cfg1", fs, ms_per_step=None if j is None else j["ms_per_step"])
if "cfg3" in which:
    import test_hip_cfg3_cfg4 as T3
    for fs in [(), ("w_vin", "w_tin"), ("w_vin", "w_tin", "w_cat"), ("w_vin", "w_cat", "w_tin", "w_tout", "w_c")]:
        vln.EnvDropDecoder.default_fp32_weights = frozenset(fs)
        parity.RECORDS.clear()
        T3._iteration(vln, torch.bfloat16, "sum", only="bf16 unrounded")
        row("envdrop_cfg3_il_a2c", fs)
vln.EnvDropDecoder.default_fp32_weights = default_env
if "monitor" in which:
    import test_hip_full_size_agents as TA
    default_mon = vln.MonitorDecoder.default_fp32_weights
    MON_SETS = [(), ("mlp",), ("mlp", "w_vh", "w_tin"), ("mlp", "w_cat", "w_vh", "w_tin"), ("mlp", "w_cat", "w_vh", "w_tin", "w_a", "w_m")]
    for fs in MON_SETS:
        vln.MonitorDecoder.default_fp32_weights = frozenset(fs)
        parity.RECORDS.clear()
        TA._monitor_full(vln, torch.bfloat16, train=True, merged=True)
        same = [r for r in parity.RECORDS if r["what"].startswith("bf16 same-weights: ") and "loss" not in r["what"]]
        code = ("import sys; sys.path.insert(0,'.'); sys.path.insert(0,'scripts'); import vln_amd as v; v.MonitorDecoder.default_fp32_weights=frozenset(%r); "
                "import json, bench_agents as b; b.configure(steps=40, warmup=20, dtype='bf16'); v.functional.set_grad_in_place(True); v.functional.set_rollout_wgrads(True); "
                "print(json.dumps(b.run_monitor()))" % (fs,))
        j = child_ms(code)
        row("self_monitor_cfg2", fs, ms_per_iteration=None if j is None else j["ms_per_iteration"],
            same_weights_worst=round(max(r["err"] for r in same), 5), same_weights_worst_what=max(same, key=lambda r: r["err"])["what"])
    vln.MonitorDecoder.default_fp32_weights = default_mon
    code = ("import sys; sys.path.insert(0,'.'); sys.path.insert(0,'scripts'); import vln_amd as v; import json, bench_agents as b; "
            "b.configure(steps=40, warmup=20, dtype='fp32'); v.functional.set_grad_in_place(True); v.functional.set_rollout_wgrads(True); print(json.dumps(b.run_monitor()))")
    print(json.dumps(dict(agent="self_monitor_cfg2", mode="fp32", **(child_ms(code) or {}))), flush=True)
