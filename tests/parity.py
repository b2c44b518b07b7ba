"""Parity bookkeeping shared by the `-m gpu` tests (test infrastructure).

north_star's bar: logits AND gradients of the HIP path match the reference CPU path within 1e-4 (fp32) / 1e-2 (bf16).
The error of a tensor is its max-abs deviation relative to the reference tensor's max-abs value (`rel_err`); every
comparison made through `check()` is recorded, printed at the end of the session (worst tensor per test) and written to
`gpurun_out/parity_report.json`, so the ACHIEVED errors -- not only pass / fail -- are on record for every run.

FP32 = 1e-4 and BF16 = 1e-2 are the only tolerances the parity tests use for outputs and gradients; a tensor that cannot
meet them carries an explicit `tol=` with the reason next to the call (and a row in DESIGN.md section 2).
"""
import json
import os

import torch

FP32 = 1e-4
BF16 = 1e-2

RECORDS = []          # dicts: test, what, err, l2, tol
RECORD_ONLY = os.environ.get("VLN_PARITY_RECORD_ONLY", "0") == "1"     # survey runs: record every error, fail nothing


def tol_of(dtype):
    return FP32 if dtype in (torch.float32, None) else BF16


def rel_err(a, b, floor=1e-6):
    a, b = torch.as_tensor(a).detach().double().cpu(), torch.as_tensor(b).detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(floor)).item() if a.numel() else 0.0


def rel_l2(a, b):
    a, b = torch.as_tensor(a).detach().double().cpu(), torch.as_tensor(b).detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-12)).item() if a.numel() else 0.0


def _test_name():
    return os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0]


def check(a, b, tol, what, floor=1e-6):
    """max |a - b| / max(max |b|, floor) < tol; the achieved error is recorded either way.  `floor`: scale below which the
    reference tensor counts as zero (a gradient that vanishes in exact arithmetic is rounding noise on both sides)."""
    assert tuple(torch.as_tensor(a).shape) == tuple(torch.as_tensor(b).shape), f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    e = rel_err(a, b, floor)
    RECORDS.append({"test": _test_name(), "what": what, "err": e, "l2": rel_l2(a, b), "tol": tol})
    if not RECORD_ONLY:
        assert e < tol, f"{what}: max-abs error {e:.3e} (relative to the tensor's max) >= {tol}"
    return e


# Parameter gradients that are ZERO IN EXACT ARITHMETIC in the parity tests' set-ups: both sides hold rounding noise only, so the
# error is taken relative to `scale_floor` x the module's largest gradient instead of the tensor's own (vanishing) scale.  They
# are listed BY NAME (round 4): every other tensor is held to its own scale, so a small but real gradient that is wrong fails.
#   * MLPwithBN (BatchNorm, then Linear -> BatchNorm -> Dropout -> ReLU per hidden layer, units.py:210-242): every Linear bias AND
#     the first BatchNorm's beta sit directly in front of a train-mode BatchNorm, whose batch mean removes any per-feature
#     constant -- `mlp.0.bias` (beta of the input BatchNorm), `mlp.1.bias`, `mlp.5.bias` (the Linear layers of the one- and
#     two-hidden-layer configurations);
#   * VisualSoftDotAttention.linear_in_v.bias: adds the same constant to every view's logit, softmax is shift-invariant;
#   * ActionScoring.linear_out.bias: adds the same constant to every candidate's logit under a CE / softmax loss.
EXACT_ZERO_GRADS = ("mlp.0.bias", "mlp.1.bias", "mlp.5.bias", "linear_in_v.bias", "decode_action.linear_out.bias")


def grad_floor(name, gmax, scale_floor=1e-2, zero_grads=EXACT_ZERO_GRADS):
    """`floor` argument of check() for the gradient of parameter `name`: the module-wide floor only for the listed exact zeros."""
    return scale_floor * gmax if any(name == z or name.endswith("." + z) for z in zero_grads) else 1e-30


def check_grads(named_params, ref_grads, tol, prefix="grad", scale_floor=1e-2, zero_grads=EXACT_ZERO_GRADS):
    """Every parameter gradient against its reference, each relative to ITS OWN max-abs value -- except the gradients listed in
    `zero_grads` (zero in exact arithmetic, rounding noise on both sides), which are judged on `scale_floor` x the largest
    gradient of the module.  A reference gradient that is exactly zero and not listed must be matched by an exact zero."""
    named_params = list(named_params)
    refs = {n: (ref_grads[n] if ref_grads.get(n) is not None else None) for n, _ in named_params}
    gmax = max([float(r.abs().max()) for r in refs.values() if r is not None and r.numel()] + [1e-30])
    for n, p in named_params:
        r = refs[n]
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        if r is None:
            r = torch.zeros_like(g, device="cpu")
        if float(r.abs().max()) == 0.0 and not any(n == z or n.endswith("." + z) for z in zero_grads):
            assert float(g.abs().max()) == 0.0, f"{prefix}[{n}]: the reference gradient is exactly zero, got max {float(g.abs().max()):.3e}"
            continue
        check(g, r, tol, f"{prefix}[{n}]", floor=grad_floor(n, gmax, scale_floor, zero_grads))


def summary_lines():
    worst = {}
    for r in RECORDS:
        k = r["test"]
        if k not in worst or r["err"] / r["tol"] > worst[k]["err"] / worst[k]["tol"]:
            worst[k] = r
    lines = []
    for k in sorted(worst):
        r = worst[k]
        lines.append(f"{k:<100s} worst {r['what']:<40s} err {r['err']:.2e}  l2 {r['l2']:.2e}  tol {r['tol']:.0e}  ({r['err'] / r['tol']:.2f} of tol)")
    return lines


def write_report(root):
    if not RECORDS:
        return None
    out = os.path.join(root, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    tag = "" if torch.cuda.is_available() else "_cpu"
    path = os.path.join(out, f"parity_report{tag}.json")
    with open(path, "w") as f:
        json.dump(RECORDS, f, indent=0)
    with open(os.path.join(out, f"parity_summary{tag}.txt"), "w") as f:
        f.write("\n".join(summary_lines()) + "\n")
    return path


# ---- bf16 mode: two oracles -------------------------------------------------------------------------------------------------
# In bf16 mode the kernels STREAM bf16 copies of every weight matrix (and of the image / candidate / context tensors where a
# module says so); activations, accumulation, recurrent state, softmax and gradients stay fp32.  Two comparisons pin that:
#   * "same weights": the fp64 oracle evaluated on the bf16-ROUNDED weights the kernels stream (straight-through, so the
#     gradients still land on the fp64 masters).  What remains is accumulation order: held to SAME_BF16 -- this is the check of
#     the kernels' arithmetic;
#   * "unrounded": the fp64 oracle on the fp32 masters = north_star's bf16 bound (1e-2).  The distance is the rounding of the
#     weights themselves (2^-9 relative per weight), amplified by softmax / recurrence / ReLU switches; tensors that cannot meet
#     1e-2 carry an explicit tolerance and a row in DESIGN.md section 2.
SAME_BF16 = 1e-4
# ... and for PARAMETER gradients under the default weight-gradient form of the bf16 mode (ops.set_wgrad_precision("bf16"): both
# operands of dW = dY^T X enter the MFMA as plain bf16, 2^-9 relative rounding each): measured 2-5e-3 on the full-size tests;
# with "split" operands (hi + lo planes, three MFMAs) the gradients meet SAME_BF16 like everything else.
SAME_BF16_GRAD = 8e-3


def same_bf16_grad_tol():
    import vln_amd
    return SAME_BF16 if vln_amd.ops.get_wgrad_precision() == "split" else SAME_BF16_GRAD


def bf16_round_st(t):
    """bf16(t) with the gradient of t (straight-through): the oracle computes with the number the kernel streams."""
    return t + (t.detach().float().bfloat16().to(t.dtype) - t.detach())


def bf16_weights(P, skip=()):
    """Parameter dict as the bf16 mode streams it: every 2-D `*weight*` matrix rounded, everything else (biases, BatchNorm,
    embedding rows, 1-row heads listed in `skip`) left in full precision."""
    return {k: (bf16_round_st(v) if (torch.is_tensor(v) and v.is_floating_point() and v.dim() == 2 and "weight" in k and k not in skip)
                else v) for k, v in P.items()}


# ---- per-matrix fp32 overrides of the bf16 mode: which state_dict keys the same-weights oracle must NOT round -----------------
ENVDROP_FP32_KEYS = {"w_vin": ("visual_attn.linear_in.weight",), "w_tin": ("text_attn.linear_in.weight",),
                     "w_tout": ("text_attn.linear_out.weight",), "w_c": ("cand_attn.weight",), "w_cat": ("lstm.weight_ih", "lstm.weight_hh")}
MONITOR_FP32_KEYS = {"mlp": ("proj_navigable_mlp.mlp.1.weight",), "w_tin": ("text_attn.linear_in.weight",),
                     "w_vh": ("visual_attn.linear_in_h.weight",), "w_cat": ("lstm.weight_ih", "lstm.weight_hh"),
                     "w_a": ("action_linear.weight",), "w_m": ("monitor_linear.weight",)}


def fp32_streamed_keys(dec, table):
    """state_dict keys of the matrices `dec.fp32_weights` streams in fp32 (left unrounded by bf16_weights(skip=...))."""
    return tuple(k for name in sorted(dec.fp32_weights) for k in table[name])
