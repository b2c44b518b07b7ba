"""GPU: the Speaker-Follower and Self-Monitoring decoders and the attention units (HIP operators through the C ABI)
against the golden vectors captured from the reference, state_dict loaded strict.  fp32 tolerance 1e-4 for outputs and gradients (tests/parity.py)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from parity import check, check_grads, grad_floor, FP32, BF16

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def vln():
    import vln_amd
    vln_amd._lib.load()
    return vln_amd


def dev(d):
    return {k: v.to(DEV) for k, v in d.items()}


@pytest.mark.parametrize("tag", ["full", "ctxonly", "visual"])
def test_softdot_units(vln, tag):
    G = load_golden("softdot_" + tag)
    I = dev(G["inp"])
    Q = I["h"].shape[1]
    D = I["ctx"].shape[2]
    att = vln.SoftDotAttention(Q, context_only=(tag != "full"), context_dim=(D if tag == "visual" else None))
    att.load_state_dict(G["param"], strict=True); att.to(DEV)
    h = I["h"].clone().requires_grad_(True); ctx = I["ctx"].clone().requires_grad_(True)
    out, attn = att(h, ctx, None if tag == "visual" else I["mask"])
    check(out, G["out"]["out"], 1e-4, "out"); check(attn, G["out"]["attn"], 1e-4, "attn")
    ((out * I["r"]).sum() + (attn * I["ra"]).sum()).backward()
    check_grads(att.named_parameters(), G["grad"], 1e-4)
    check(h.grad, G["grad"]["h"], 1e-4, "dh"); check(ctx.grad, G["grad"]["ctx"], 1e-4, "dctx")


@pytest.mark.parametrize("tag", ["follower", "monitor"])
def test_visualdot_units(vln, tag):
    G = load_golden("visualdot_" + tag)
    I = dev(G["inp"])
    vdim = I["v"].shape[2] if tag == "follower" else None
    dot = G["param"]["linear_in_h.weight"].shape[0]
    att = vln.VisualSoftDotAttention(I["h"].shape[1], vdim, dot)
    att.load_state_dict(G["param"], strict=True); att.to(DEV)
    h = I["h"].clone().requires_grad_(True); v = I["v"].clone().requires_grad_(True)
    out, attn = att(h, v, I["mask"] if tag == "monitor" else None)
    check(out, G["out"]["out"], 1e-4, "out"); check(attn, G["out"]["attn"], 1e-4, "attn")
    ((out * I["r"]).sum() + (attn * I["ra"]).sum()).backward()
    check_grads(att.named_parameters(), G["grad"], 1e-4)
    check(h.grad, G["grad"]["h"], 1e-4, "dh"); check(v.grad, G["grad"]["v"], 1e-4, "dv")


@pytest.mark.parametrize("name", ["follower_step", "follower_chain3"])
def test_follower_golden(vln, name):
    G = load_golden(name)
    cfg, I = G["cfg"], dev(G["inp"])
    F = int(cfg["IMG"]) + int(cfg["ANG"])
    dec = vln.AttnDecoderLSTM(int(cfg["H"]), 0.5, action_embed_size=F, feature_size=F)
    dec.load_state_dict(G["param"], strict=True); dec.to(DEV).eval()
    ctx = I["ctx"].clone().requires_grad_(True)
    h = I["h0"].clone().requires_grad_(True); c = I["c0"].clone().requires_grad_(True)
    h0, c0 = h, c
    loss = 0.
    for t in range(int(cfg["steps"])):
        logit, (h, c), (ac, av) = dec(I[f"img{t}"], I[f"a_prev{t}"], I[f"cand{t}"], h, c, ctx, I["ctx_mask"])
        check(logit, G["out"][f"logit{t}"], 1e-4, f"logit{t}")
        check(h, G["out"][f"h1_{t}"], 1e-4, "h1"); check(c, G["out"][f"c1_{t}"], 1e-4, "c1")
        check(ac, G["out"][f"alpha_c{t}"], 1e-4, "alpha_c"); check(av, G["out"][f"alpha_v{t}"], 1e-4, "alpha_v")
        loss = loss + (logit * I[f"rl{t}"]).sum()
    loss = loss + (h * I["rf"]).sum() + (c * I["rc"]).sum()
    loss.backward()
    check_grads(dec.named_parameters(), G["grad"], 1e-4)
    check(ctx.grad, G["grad"]["ctx"], 1e-4, "dctx"); check(h0.grad, G["grad"]["h0"], 1e-4, "dh0"); check(c0.grad, G["grad"]["c0"], 1e-4, "dc0")


@pytest.mark.parametrize("name", ["monitor_step_train", "monitor_step_eval"])
def test_monitor_golden(vln, name):
    G = load_golden(name)
    cfg, I = G["cfg"], dev(G["inp"])
    F = int(cfg["IMG"]) + int(cfg["ANG"])
    training = bool(cfg["training"])
    dec = vln.MonitorDecoder(int(cfg["H"]), 0.5, int(cfg["L"]), mlp_dims=[int(cfg["M"])], action_embed_size=F, feature_size=F)
    dec.load_state_dict(G["param"], strict=True); dec.to(DEV)
    if training:            # the golden was captured in train mode with every Dropout at p = 0 (BatchNorm batch stats on)
        dec.train()
        dec.drop_ratio = 0.0; dec.position.p = 0.0
        for m in dec.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
    else:
        dec.eval()
    ctx = I["ctx"].clone().requires_grad_(True)
    h = I["h0"].clone().requires_grad_(True); c = I["c0"].clone().requires_grad_(True)
    (logit, prog), (h1, c1), (ca, va) = dec(None, I["a_prev"], I["cand"], h, c, ctx, I["ctx_mask"], I["cand_mask"])
    for k, v in (("logit", logit), ("prog", prog), ("h1", h1), ("c1", c1), ("ctx_attn", ca), ("cand_attn", va)):
        check(v, G["out"][k], 1e-4, k)
    ((logit * I["rl"]).sum() + (prog * I["rp"]).sum() + (h1 * I["rh"]).sum() + (c1 * I["rc"]).sum()).backward()
    check_grads(dec.named_parameters(), G["grad"], 1e-4)
    check(ctx.grad, G["grad"]["ctx"], 1e-4, "dctx"); check(h.grad, G["grad"]["h0"], 1e-4, "dh0"); check(c.grad, G["grad"]["c0"], 1e-4, "dc0")
    if training:            # two running-stat updates per step (previous action rows, then B*C candidate rows)
        sd = dec.state_dict()
        for k in ("proj_navigable_mlp.mlp.0.running_mean", "proj_navigable_mlp.mlp.0.running_var",
                  "proj_navigable_mlp.mlp.2.running_mean", "proj_navigable_mlp.mlp.2.running_var"):
            check(sd[k], G["param_after"][k], 1e-4, k)
        assert int(sd["proj_navigable_mlp.mlp.0.num_batches_tracked"]) == 2


@pytest.mark.parametrize("cdt", [torch.float32, torch.bfloat16])
@pytest.mark.usefixtures("split_wgrads")
def test_monitor_fused_step_equals_operator_path(vln, cdt):
    """MonitorDecoder with the one-node core (functional.MonitorCoreFn) against the operator-by-operator path it replaces,
    in TRAINING mode with every dropout on (both draw the same Philox masks): outputs, state, attention weights, all
    parameter gradients and the gradients of ctx / h0 / c0, over a two-step chain."""
    B, C, L, H, M, F = 24, 7, 20, 64, 128, 256
    g = torch.Generator().manual_seed(77)
    ctx0 = torch.randn(B, L, H, generator=g); h00 = torch.randn(B, H, generator=g) * 0.5; c00 = torch.randn(B, H, generator=g) * 0.5
    a_prev = torch.randn(B, F, generator=g).abs(); cands = [torch.randn(B, C, F, generator=g).abs() for _ in range(2)]
    lens = torch.randint(5, L + 1, (B,), generator=g); ctx_mask = torch.arange(L)[None, :] >= lens[:, None]
    nc = torch.randint(2, C + 1, (B,), generator=g); cmask = torch.arange(C)[None, :] >= nc[:, None]
    r = [torch.randn(B, C, generator=g), torch.randn(B, generator=g), torch.randn(B, H, generator=g), torch.randn(B, H, generator=g),
         torch.randn(B, L, generator=g)]
    res = []
    torch.manual_seed(5)
    ref = vln.MonitorDecoder(H, 0.5, L, mlp_dims=[32, M], action_embed_size=F, feature_size=F, compute_dtype=cdt)
    sd = {k: v.clone() for k, v in ref.state_dict().items()}
    for fused in (True, False):
        dec = vln.MonitorDecoder(H, 0.5, L, mlp_dims=[32, M], action_embed_size=F, feature_size=F, compute_dtype=cdt)
        dec.load_state_dict(sd); dec.to(DEV).train()
        dec.fused_step = fused
        ctx = ctx0.to(DEV).requires_grad_(True); h = h00.to(DEV).requires_grad_(True); c = c00.to(DEV).requires_grad_(True)
        hh, cc, ap, total, outs = h, c, a_prev.to(DEV), 0.0, []
        for t in range(2):
            (logit, prog), (hh, cc), (ww, mw) = dec(None, ap, cands[t].to(DEV), hh, cc, ctx, ctx_mask.to(DEV), cmask.to(DEV))
            total = total + (logit.masked_fill(cmask.to(DEV), 0.0) * r[0].to(DEV)).sum() + (prog * r[1].to(DEV)).sum() + (ww * r[4].to(DEV)).sum()
            outs += [logit, prog, ww, mw]
            ap = cands[t][:, 0].to(DEV)
        total = total + (hh * r[2].to(DEV)).sum() + (cc * r[3].to(DEV)).sum()
        total.backward()
        res.append((outs + [hh, cc], {n: p.grad.clone() for n, p in dec.named_parameters()}, [ctx.grad, h.grad, c.grad],
                    {k: v.clone() for k, v in dec.state_dict().items() if "running" in k}))
    tol = 2e-4 if cdt == torch.float32 else 2e-2
    gscale = max(v.abs().max().item() for v in res[1][1].values())
    def close(a, b, what, floor=1e-6):
        err = (a.double() - b.double()).abs().max().item() / max(b.double().abs().max().item(), floor)
        assert err < tol, (what, err)
    for i, (a, b) in enumerate(zip(res[0][0], res[1][0])):
        close(a, b, f"out{i}")
    for n in res[0][1]:      # gradients that are zero in exact arithmetic (a bias in front of a BatchNorm) are fp32 noise:
        close(res[0][1][n], res[1][1][n], "grad " + n, floor=grad_floor(n, gscale))      # judged on the scale of the real gradients
    for i, (a, b) in enumerate(zip(res[0][2], res[1][2])):
        close(a, b, f"input grad {i}")
    for k in res[0][3]:
        close(res[0][3][k], res[1][3][k], k)


@pytest.mark.parametrize("cdt", [torch.float32, torch.bfloat16])
@pytest.mark.usefixtures("split_wgrads")
def test_monitor_c_call_step_equals_python_driven_node(vln, cdt):
    """`vln_monitor_step_fwd/bwd` (the step after the BN-MLP as ONE C call each way, csrc/monitor.hip) issues the launch sequence
    that functional.MonitorCoreFn drives from Python: outputs bit-identical, gradients to summation-order rounding, in training
    mode with every dropout on, over a two-step chain with attention-map gradients flowing in."""
    B, C, L, H, M, F = 24, 7, 20, 64, 128, 256
    g = torch.Generator().manual_seed(78)
    ctx0 = torch.randn(B, L, H, generator=g); h00 = torch.randn(B, H, generator=g) * 0.5; c00 = torch.randn(B, H, generator=g) * 0.5
    a_prev = torch.randn(B, F, generator=g).abs(); cands = [torch.randn(B, C, F, generator=g).abs() for _ in range(2)]
    lens = torch.randint(5, L + 1, (B,), generator=g); ctx_mask = torch.arange(L)[None, :] >= lens[:, None]
    nc = torch.randint(2, C + 1, (B,), generator=g); cmask = torch.arange(C)[None, :] >= nc[:, None]
    r = [torch.randn(B, C, generator=g), torch.randn(B, generator=g), torch.randn(B, H, generator=g), torch.randn(B, H, generator=g),
         torch.randn(B, L, generator=g), torch.randn(B, C, generator=g)]
    torch.manual_seed(5)
    sd = {k: v.clone() for k, v in vln.MonitorDecoder(H, 0.5, L, mlp_dims=[32, M], action_embed_size=F, feature_size=F).state_dict().items()}
    res = []
    for c_step in (True, False):
        dec = vln.MonitorDecoder(H, 0.5, L, mlp_dims=[32, M], action_embed_size=F, feature_size=F, compute_dtype=cdt)
        dec.load_state_dict(sd); dec.to(DEV).train()
        dec.c_step = c_step
        ctx = ctx0.to(DEV).requires_grad_(True); h = h00.to(DEV).requires_grad_(True); c = c00.to(DEV).requires_grad_(True)
        hh, cc, ap, total, outs = h, c, a_prev.to(DEV), 0.0, []
        for t in range(2):
            (logit, prog), (hh, cc), (ww, mw) = dec(None, ap, cands[t].to(DEV), hh, cc, ctx, ctx_mask.to(DEV), cmask.to(DEV))
            total = total + (logit.masked_fill(cmask.to(DEV), 0.0) * r[0].to(DEV)).sum() + (prog * r[1].to(DEV)).sum() \
                + (ww * r[4].to(DEV)).sum() + (mw * r[5].to(DEV)).sum()
            outs += [logit, prog, ww, mw]
            ap = cands[t][:, 0].to(DEV)
        total = total + (hh * r[2].to(DEV)).sum() + (cc * r[3].to(DEV)).sum()
        total.backward()
        res.append((outs + [hh, cc], {n: p.grad.clone() for n, p in dec.named_parameters()}, [ctx.grad, h.grad, c.grad]))
    for i, (a, b) in enumerate(zip(res[0][0], res[1][0])):
        assert torch.equal(a, b), f"output {i}"
    gscale = max(v.abs().max().item() for v in res[1][1].values())
    for n in res[0][1]:
        check(res[0][1][n], res[1][1][n], 2e-5, f"grad[{n}]", floor=grad_floor(n, gscale))
    for i, (a, b) in enumerate(zip(res[0][2], res[1][2])):
        check(a, b, 2e-5, f"input grad {i}")


@pytest.mark.parametrize("cdt", [torch.float32, torch.bfloat16])
@pytest.mark.usefixtures("split_wgrads")
def test_follower_fused_step_equals_operator_path(vln, cdt):
    """AttnDecoderLSTM as one autograd node (functional.FollowerCoreFn) against the operator-by-operator path it replaces,
    in TRAINING mode with both dropouts on (same Philox masks): logits, state, both attention maps, every parameter
    gradient (incl. the factorised d linear_in_v / d linear_act) and the gradients of ctx / h0 / c0, over a two-step chain
    with ragged candidate counts and attention-map gradients flowing in."""
    B, V, C, L, H, F = 12, 36, 6, 17, 64, 96
    g = torch.Generator().manual_seed(99)
    ctx0 = torch.randn(B, L, H, generator=g); h00 = torch.randn(B, H, generator=g) * 0.5; c00 = torch.randn(B, H, generator=g) * 0.5
    imgs = [torch.randn(B, V, F, generator=g).abs() for _ in range(2)]
    cands = [torch.randn(B, C, F, generator=g).abs() for _ in range(2)]
    a_prev = torch.randn(B, F, generator=g).abs()
    lens = torch.randint(5, L + 1, (B,), generator=g); ctx_mask = torch.arange(L)[None, :] >= lens[:, None]
    r = [torch.randn(B, C, generator=g), torch.randn(B, H, generator=g), torch.randn(B, H, generator=g), torch.randn(B, L, generator=g),
         torch.randn(B, V, generator=g)]
    torch.manual_seed(6)
    ref = vln.AttnDecoderLSTM(H, 0.5, F, F, compute_dtype=cdt)
    sd = {k: v.clone() for k, v in ref.state_dict().items()}
    res = []
    for fused in (True, False):
        dec = vln.AttnDecoderLSTM(H, 0.5, F, F, compute_dtype=cdt)
        dec.load_state_dict(sd); dec.to(DEV).train()
        dec.fused_step = fused
        ctx = ctx0.to(DEV).requires_grad_(True); h = h00.to(DEV).requires_grad_(True); c = c00.to(DEV).requires_grad_(True)
        hh, cc, ap, total, outs = h, c, a_prev.to(DEV), 0.0, []
        for t in range(2):
            logit, (hh, cc), (ww, vw) = dec(imgs[t].to(DEV), ap, cands[t].to(DEV), hh, cc, ctx, ctx_mask.to(DEV))
            total = total + (logit * r[0].to(DEV)).sum() + (ww * r[3].to(DEV)).sum() + (vw * r[4].to(DEV)).sum()
            outs += [logit, ww, vw]
            ap = cands[t][:, 0].to(DEV)
        total = total + (hh * r[1].to(DEV)).sum() + (cc * r[2].to(DEV)).sum()
        total.backward()
        res.append((outs + [hh, cc], {n: p.grad.clone() for n, p in dec.named_parameters()}, [ctx.grad, h.grad, c.grad]))
    tol = 2e-4 if cdt == torch.float32 else 2e-2
    gscale = max(v.abs().max().item() for v in res[1][1].values())

    def close(a, b, what, floor=1e-6):
        err = (a.double() - b.double()).abs().max().item() / max(b.double().abs().max().item(), floor)
        assert err < tol, (what, err)
    for i, (a, b) in enumerate(zip(res[0][0], res[1][0])):
        close(a, b, f"out{i}")
    for n in res[0][1]:      # d linear_in_v.bias is zero in exact arithmetic (softmax rows): judged on the scale of real gradients
        close(res[0][1][n], res[1][1][n], "grad " + n, floor=grad_floor(n, gscale))
    for i, (a, b) in enumerate(zip(res[0][2], res[1][2])):
        close(a, b, f"input grad {i}")


@pytest.mark.parametrize("cdt", [torch.float32, torch.bfloat16])
@pytest.mark.usefixtures("split_wgrads")
def test_follower_c_call_step_equals_python_driven_node(vln, cdt):
    """`vln_follower_step_fwd/bwd` (csrc/follower.hip) against functional.FollowerCoreFn: the same launch sequence issued by the
    library -- outputs bit-identical; gradients to summation-order rounding (d linear_in_v takes sum_v dl_v img_v from the
    attention backward's own pass instead of a second sweep over the panorama)."""
    B, V, C, L, H, F = 12, 36, 6, 17, 64, 96
    g = torch.Generator().manual_seed(98)
    ctx0 = torch.randn(B, L, H, generator=g); h00 = torch.randn(B, H, generator=g) * 0.5; c00 = torch.randn(B, H, generator=g) * 0.5
    imgs = [torch.randn(B, V, F, generator=g).abs() for _ in range(2)]
    cands = [torch.randn(B, C, F, generator=g).abs() for _ in range(2)]
    a_prev = torch.randn(B, F, generator=g).abs()
    lens = torch.randint(5, L + 1, (B,), generator=g); ctx_mask = torch.arange(L)[None, :] >= lens[:, None]
    r = [torch.randn(B, C, generator=g), torch.randn(B, H, generator=g), torch.randn(B, H, generator=g), torch.randn(B, L, generator=g),
         torch.randn(B, V, generator=g)]
    torch.manual_seed(6)
    sd = {k: v.clone() for k, v in vln.AttnDecoderLSTM(H, 0.5, F, F).state_dict().items()}
    res = []
    for c_step in (True, False):
        dec = vln.AttnDecoderLSTM(H, 0.5, F, F, compute_dtype=cdt)
        dec.load_state_dict(sd); dec.to(DEV).train()
        dec.c_step = c_step
        ctx = ctx0.to(DEV).requires_grad_(True); h = h00.to(DEV).requires_grad_(True); c = c00.to(DEV).requires_grad_(True)
        hh, cc, ap, total, outs = h, c, a_prev.to(DEV), 0.0, []
        for t in range(2):
            logit, (hh, cc), (ww, vw) = dec(imgs[t].to(DEV), ap, cands[t].to(DEV), hh, cc, ctx, ctx_mask.to(DEV))
            total = total + (logit * r[0].to(DEV)).sum() + (ww * r[3].to(DEV)).sum() + (vw * r[4].to(DEV)).sum()
            outs += [logit, ww, vw]
            ap = cands[t][:, 0].to(DEV)
        total = total + (hh * r[1].to(DEV)).sum() + (cc * r[2].to(DEV)).sum()
        total.backward()
        res.append((outs + [hh, cc], {n: p.grad.clone() for n, p in dec.named_parameters()}, [ctx.grad, h.grad, c.grad]))
    for i, (a, b) in enumerate(zip(res[0][0], res[1][0])):
        # (bit for bit until round 5; the C call now hands its projected query to the attention as split-K slabs, the Python-driven
        #  node as one matrix: the same products in another summation order)
        check(a, b, 2e-5, f"output {i}")
    gscale = max(v.abs().max().item() for v in res[1][1].values())
    for n in res[0][1]:
        check(res[0][1][n], res[1][1][n], 2e-5, f"grad[{n}]", floor=grad_floor(n, gscale))
    for i, (a, b) in enumerate(zip(res[0][2], res[1][2])):
        check(a, b, 2e-5, f"input grad {i}")


@pytest.mark.parametrize("kind", ["follower", "monitor"])
@pytest.mark.usefixtures("split_wgrads")
def test_fused_nodes_grad_in_place(vln, kind):
    """functional.set_grad_in_place: the fused nodes add their Linear gradients into an existing p.grad (no AccumulateGrad
    launches) -- after two backward passes p.grad equals the default route's, bit for bit where the launches are the same
    and to rounding where accumulation order differs."""
    B, V, C, L, H, F = 8, 36, 5, 12, 64, 96
    g = torch.Generator().manual_seed(123)
    ctx0 = torch.randn(B, L, H, generator=g); h0 = torch.randn(B, H, generator=g) * 0.5; c0 = torch.randn(B, H, generator=g) * 0.5
    img = torch.randn(B, V, F, generator=g).abs(); cand = torch.randn(B, C, F, generator=g).abs(); ap = torch.randn(B, F, generator=g).abs()
    lens = torch.randint(4, L + 1, (B,), generator=g); ctx_mask = (torch.arange(L)[None, :] >= lens[:, None]).to(DEV)
    cmask = (torch.arange(C)[None, :] >= torch.randint(2, C + 1, (B,), generator=g)[:, None]).to(DEV)
    r = torch.randn(B, C, generator=g).to(DEV)
    torch.manual_seed(8)
    make = (lambda: vln.AttnDecoderLSTM(H, 0.5, F, F)) if kind == "follower" else \
        (lambda: vln.MonitorDecoder(H, 0.5, L, mlp_dims=[32, 128], action_embed_size=F, feature_size=F))
    sd = {k: v.clone() for k, v in make().state_dict().items()}
    out = []
    try:
        for inplace in (False, True):
            vln.functional.set_grad_in_place(inplace)
            dec = make(); dec.load_state_dict(sd); dec.to(DEV).train()
            for p in dec.parameters():
                p.grad = torch.zeros_like(p)
            for _ in range(2):                                      # two backward passes: the second one accumulates either way
                if kind == "follower":
                    logit, (hh, cc), _ = dec(img.to(DEV), ap.to(DEV), cand.to(DEV), h0.to(DEV), c0.to(DEV), ctx0.to(DEV), ctx_mask)
                else:
                    (logit, prog), (hh, cc), _ = dec(None, ap.to(DEV), cand.to(DEV), h0.to(DEV), c0.to(DEV), ctx0.to(DEV), ctx_mask, cmask)
                    logit = logit.masked_fill(cmask, 0.0) + prog[:, None]
                ((logit * r).sum() + hh.sum() * 0.3 + cc.sum() * 0.2).backward()
            out.append({n: p.grad.clone() for n, p in dec.named_parameters()})
    finally:
        vln.functional.set_grad_in_place(False)
    scale = max(v.abs().max().item() for v in out[0].values())
    for n in out[0]:
        err = (out[0][n].double() - out[1][n].double()).abs().max().item()
        assert err <= 2e-5 * max(out[0][n].abs().max().item(), 1e-2 * scale), (n, err)


# ---- speaker modules (units.py:286-395) ------------------------------------------------------------------
def _holder_name(n):
    """reference parameter name -> attribute path here (the nn.LSTM parameter holder sits one level down)"""
    for pre in ("post_lstm.", "lstm."):
        if n.startswith(pre):
            return pre + "rnn." + n[len(pre):]
    return n


@pytest.mark.parametrize("kind", ["bi", "uni"])
def test_speaker_encoder_golden(vln, kind):
    G = load_golden("speaker_encoder_" + kind)
    cfg, I = G["cfg"], dev(G["inp"])
    enc = vln.SpeakerEncoder(int(cfg["F"]), int(cfg["H"]), 0.5, bool(int(cfg["bidir"])), int(cfg["ANG"]), 0.3)
    enc.load_state_dict(G["param"], strict=True)                       # the reference's key names
    enc.to(DEV).eval()
    ctx = enc(I["act"].clone(), I["feat"].clone(), None)
    check(ctx, G["out"]["ctx"], 1e-4, "ctx")
    (ctx * I["r"]).sum().backward()
    names = dict(enc.named_parameters())
    for n, g in G["grad"].items():
        check(names[_holder_name(n)].grad, g, 1e-4, n)


def test_speaker_decoder_golden(vln):
    G = load_golden("speaker_decoder")
    cfg, I = G["cfg"], dev(G["inp"])
    dec = vln.SpeakerDecoder(int(cfg["VOC"]), int(cfg["E"]), 0, int(cfg["H"]), 0.5)
    dec.load_state_dict(G["param"], strict=True)
    dec.to(DEV).eval()
    ctx = I["ctx"].clone().requires_grad_(True)
    B, H = ctx.shape[0], ctx.shape[2]
    z = torch.zeros(1, B, H, device=DEV)
    logit, h1, c1 = dec(I["words"], ctx, I["mask"], z, z)
    for a, k in ((logit, "logit"), (h1, "h1"), (c1, "c1")):
        check(a, G["out"][k], 1e-4, k)
    ((logit * I["r"]).sum() + (h1 * 0.3).sum() + (c1 * 0.2).sum()).backward()
    names = dict(dec.named_parameters())
    for n, g in G["grad"].items():
        if n == "ctx":
            check(ctx.grad, g, 1e-4, "dctx")
            continue
        p = names[_holder_name(n)]
        if p.grad is None:                                             # baseline_projection is unused by forward
            assert float(g.abs().max()) == 0.0, n
        else:
            check(p.grad, g, 1e-4, n)
    with torch.no_grad():                                              # one word from a carried (non-zero) state
        l2, h2, c2 = dec(I["words"][:, :1], ctx, I["mask"], I["hs"], I["cs"])
    for a, k in ((l2, "step_logit"), (h2, "step_h"), (c2, "step_c")):
        check(a, G["out"][k], 1e-4, k)


# ---- speaker loop: teacher forcing, greedy / sampled inference, back-translation hook (speaker.py:235-376) ------------
def _speaker_from_golden(vln, G):
    cfg = G["cfg"]
    F, H, ANG, VOC, E = (int(cfg[k]) for k in ("F", "H", "ANG", "VOC", "E"))
    enc = vln.SpeakerEncoder(F, H, 0.5, True, ANG, 0.3)
    dec = vln.SpeakerDecoder(VOC, E, 0, H, 0.5)
    enc.load_state_dict(G["enc"], strict=True)
    dec.load_state_dict(G["dec"], strict=True)
    return vln.Speaker(enc.to(DEV), dec.to(DEV), max_decode=int(cfg["MAXD"]))


def test_speaker_loop_golden(vln):
    G = load_golden("speaker_loop")
    I, out = dev(G["inp"]), G["out"]
    spk = _speaker_from_golden(vln, G)
    lengths = G["inp"]["lengths"].tolist()
    # eval-mode numbers are what the golden holds: run the "train" entry with the modules' dropout probabilities at 0
    spk.encoder.drop_ratio = spk.encoder.feat_drop_ratio = spk.decoder.drop_ratio = 0.0
    loss = spk.teacher_forcing(I["can"].clone(), I["img"].clone(), lengths, I["insts"], train=True)
    check(loss, out["loss"], 1e-4, "teacher-forcing loss")
    loss.backward()
    ne, nd = dict(spk.encoder.named_parameters()), dict(spk.decoder.named_parameters())
    for n, g in G["grad_enc"].items():
        check(ne[_holder_name(n)].grad, g, 1e-4, "grad encoder." + n)
    for n, g in G["grad_dec"].items():
        p = nd[_holder_name(n)]
        check(p.grad if p.grad is not None else torch.zeros_like(p), g, 1e-4, "grad decoder." + n)
    per_word = spk.teacher_forcing(I["can"].clone(), I["img"].clone(), lengths, I["insts"], train=False, for_listener=True)
    check(per_word, out["per_word"], 1e-4, "un-reduced losses")
    l, word_accu, sent_accu = spk.teacher_forcing(I["can"].clone(), I["img"].clone(), lengths, I["insts"], train=False)
    assert abs(l - out["loss"].item()) < 1e-4 * max(1.0, abs(l))
    gt = G["inp"]["insts"][:, 1:]
    ok = (out["predict"][:, :-1] == gt) & (gt != 0)
    assert abs(word_accu - ok.sum().item() / (gt != 0).sum().item()) < 1e-9
    assert abs(sent_accu - (ok.sum(1) == (gt != 0).sum(1)).sum().item() / gt.shape[0]) < 1e-9
    # greedy inference under a shared environment-dropout mask: the same words as the reference's modules chose
    words = spk.infer_batch(I["can"].clone(), I["img"].clone(), lengths, featdropmask=I["noise"])
    assert (words == out["words"].numpy()).all(), (words, out["words"])
    # the back-translation hook; eval-mode follower -> the mask is all ones -> the unmasked greedy words
    fol = vln.EnvDropDecoder(32, 0.5, 0.3, 16, int(G["cfg"]["ANG"]), int(G["cfg"]["F"])).to(DEV).eval()
    insts, noise = vln.back_translate(spk, fol, I["can"].clone(), I["img"].clone(), lengths)
    assert torch.equal(noise.cpu(), torch.ones(int(G["cfg"]["F"]) - int(G["cfg"]["ANG"])))
    plain = out["words_plain"].numpy()
    assert (insts[:, 1:1 + plain.shape[1]] [:, :-1] == plain[:, :-1]).all() and (insts[:, 0] == 3).all()
    assert all(2 in row for row in insts)
    fol.train()
    m = vln.env_drop_mask(fol)
    vals = set(m.unique().cpu().tolist())
    assert vals <= {0.0, float(torch.tensor(1.0 / 0.7, dtype=torch.float32))} and len(vals) == 2
    assert not torch.equal(m, vln.env_drop_mask(fol))                   # a fresh mask per batch


def test_speaker_loop_sampling_vs_oracle(vln):
    """Sampled inference in train mode (the speaker's own RL path, speaker.py:337-345): the returned log-probs /
    entropies carry gradients and equal the oracle's for the SAME words (dropout off; draws cannot be RNG-matched)."""
    from oracle import rollout as R
    G = load_golden("speaker_loop")
    I = dev(G["inp"])
    spk = _speaker_from_golden(vln, G)
    spk.encoder.drop_ratio = spk.encoder.feat_drop_ratio = spk.decoder.drop_ratio = 0.0
    lengths = G["inp"]["lengths"].tolist()
    H, ANG, MAXD = int(G["cfg"]["H"]), int(G["cfg"]["ANG"]), int(G["cfg"]["MAXD"])
    words, logp, hid, ent = spk.infer_batch(I["can"].clone(), I["img"].clone(), lengths, sampling=True, train=True)
    n = words.shape[1]
    assert logp.shape == (4, n) and hid.shape == (4, n, H) and ent.shape == (4, n) and logp.requires_grad
    assert (words != 1).all()                                           # <UNK> masked out
    # what the model fed back: the sampled word, also for rows that had already ended -> rebuild from the pad pattern
    ora = R.SpeakerOracle(G["enc"], G["dec"], True)
    fed = words.copy()
    # rows that ended were padded on the CPU copy only; their fed-back words are unknown to the caller, so compare the
    # steps up to and including each row's <EOS>
    live = np.ones_like(words, dtype=bool)
    for b in range(words.shape[0]):
        e = np.where(words[b] == 2)[0]
        if len(e):
            live[b, e[0] + 1:] = False
    # replay the oracle with the same words while they are live (after the end any word keeps the state finite)
    fed[~live] = 4
    ow, step_logits = R.speaker_infer_batch(ora.encode, ora.decode, G["inp"]["can"], G["inp"]["img"], lengths, H, n, angle=ANG,
                                            inject_words=fed)
    lp_ref = torch.log_softmax(step_logits, dim=2).gather(2, torch.as_tensor(fed)[:, :, None]).squeeze(2)
    p = torch.softmax(step_logits, dim=2)
    ent_ref = -(p * torch.log(p.clamp(1.1920929e-07, 1 - 1.1920929e-07))).sum(2)
    sel = torch.as_tensor(live)
    check(logp.detach().cpu()[sel], lp_ref[sel], 2e-4, "log-probs of the sampled words")
    check(ent.detach().cpu()[sel], ent_ref[sel], 2e-4, "entropies")
    (-(logp * 0.5).sum() - 0.01 * ent.sum() + hid.sum() * 1e-3).backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in spk.decoder.projection.parameters())
    assert spk.encoder.lstm.rnn.weight_ih_l0.grad.abs().sum() > 0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("R,train", [(1024, True), (128, True), (1000, False)])
def test_bn_mlp_c_call_equals_python_driven_launches(vln, dtype, R, train):
    """MLPwithBN (units.py:210-242) as ONE C call each way (vln_bn_mlp_fwd / bwd) against the same launches driven from Python
    (functional.set_bn_mlp_c_call(False)): output, running statistics, counters, input gradient and every parameter gradient,
    with dropout and padded-row zeroing on, tall (row-chunked BatchNorm) and short inputs, training and eval."""
    from vln_amd import functional as Fh
    from vln_amd.decoders import MLPwithBN
    g = torch.Generator().manual_seed(R)
    F = 64 + 128
    x0 = (torch.randn(R, F, generator=g).abs() * 0.5).to(DEV)
    rz = (torch.rand(R, generator=g) < 0.2).to(DEV)
    r = torch.randn(R, 256, generator=g).to(DEV)
    res = []
    try:
        for c_call in (True, False):
            Fh.set_bn_mlp_c_call(c_call)
            torch.manual_seed(3)
            mlp = MLPwithBN(F, (32, 256), dropout=0.5, use_bn=True, relu=True).to(DEV)
            for m in mlp.modules():
                if hasattr(m, "compute_dtype"):
                    m.compute_dtype = dtype
            mlp.train(train)
            out = []
            for it in range(2):                      # second pass: gradients accumulate, running statistics move again
                x = x0.clone().requires_grad_(True)
                y = mlp(x, row_zero=rz)
                (y * r).sum().backward()
                out.append((y.detach().clone(), x.grad.clone()))
            res.append((out, {n: p.grad.clone() for n, p in mlp.named_parameters()},
                        {n: b.clone() for n, b in mlp.named_buffers()}))
    finally:
        Fh.set_bn_mlp_c_call(True)
    for (ya, dxa), (yb, dxb) in zip(res[0][0], res[1][0]):
        check(ya, yb, 1e-6, "y"); check(dxa, dxb, 2e-5, "dx")
    for n in res[0][1]:
        check(res[0][1][n], res[1][1][n], 2e-5 if dtype == torch.float32 else 8e-3, f"grad[{n}]")
    for n in res[0][2]:
        if res[0][2][n].dtype.is_floating_point:
            check(res[0][2][n], res[1][2][n], 1e-6, f"buffer[{n}]")
        else:
            assert torch.equal(res[0][2][n], res[1][2][n])


@pytest.mark.parametrize("cdt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("agent", ["monitor", "follower"])
@pytest.mark.usefixtures("split_wgrads")
def test_rollout_level_parameter_gradients_equal_per_step(vln, agent, cdt):
    """functional.RolloutWgrads: the per-step C calls skip their weight / bias gradient launches, the operands of all steps
    sit in arena slots, and ONE segmented pack + contraction + column sum per group runs when the backward pass ends.  Same
    outputs (bit for bit) and the same gradients as the per-step launches up to the summation order over steps -- over a
    three-step chain in training mode, twice in a row (the second rollout reuses the slots)."""
    T = 3
    g = torch.Generator().manual_seed(123)
    if agent == "monitor":
        B, C, L, H, M, F = 24, 7, 20, 64, 128, 256
        make = lambda: vln.MonitorDecoder(H, 0.5, L, mlp_dims=[32, M], action_embed_size=F, feature_size=F, compute_dtype=cdt)
    else:
        B, V, C, L, H, F = 12, 36, 6, 17, 64, 96
        make = lambda: vln.AttnDecoderLSTM(H, 0.5, F, F, compute_dtype=cdt)
        imgs = [torch.randn(B, V, F, generator=g).abs().to(DEV) for _ in range(T)]
    ctx0 = torch.randn(B, L, H, generator=g); h00 = torch.randn(B, H, generator=g) * 0.5; c00 = torch.randn(B, H, generator=g) * 0.5
    a_prev = torch.randn(B, F, generator=g).abs().to(DEV); cands = [torch.randn(B, C, F, generator=g).abs().to(DEV) for _ in range(T)]
    lens = torch.randint(5, L + 1, (B,), generator=g); ctx_mask = (torch.arange(L)[None, :] >= lens[:, None]).to(DEV)
    nc = torch.randint(2, C + 1, (B,), generator=g); cmask = (torch.arange(C)[None, :] >= nc[:, None]).to(DEV)
    rl = torch.randn(B, C, generator=g).to(DEV); rh = torch.randn(B, H, generator=g).to(DEV); rp = torch.randn(B, generator=g).to(DEV)
    torch.manual_seed(7)
    sd = {k: v.clone() for k, v in make().state_dict().items()}
    F_ = vln.functional
    res = []
    try:
        F_.set_grad_in_place(True)
        for deferred in (False, True):
            F_.set_rollout_wgrads(deferred)
            F_.ROLLOUT_WGRADS.stats[:] = [0, 0]
            dec = make(); dec.load_state_dict(sd); dec.to(DEV).train()
            per_rollout = []
            n_roll = 3 if deferred else 2
            for rollout in range(n_roll):
                if rollout == 2:
                    # as a backward pass that RAISED leaves it: the engine dropped the queued flush, `_queued` stayed set.
                    # The next forward step must notice, drop the stale jobs and let this rollout's backward queue its own flush.
                    F_.ROLLOUT_WGRADS._queued = True
                    stale0 = F_.ROLLOUT_WGRADS.stale_dropped
                dec._calls = 0                                  # the same dropout masks in both rollouts and both modes
                for m_ in dec.modules():
                    if hasattr(m_, "_calls"):
                        m_._calls = 0
                for p in dec.parameters():
                    p.grad = torch.zeros_like(p)
                ctx = ctx0.to(DEV).requires_grad_(True); h = h00.to(DEV).requires_grad_(True); c = c00.to(DEV).requires_grad_(True)
                hh, cc, ap, total, outs = h, c, a_prev, 0.0, []
                for t in range(T):
                    if agent == "monitor":
                        (logit, prog), (hh, cc), _ = dec(None, ap, cands[t], hh, cc, ctx, ctx_mask, cmask)
                        total = total + (logit.masked_fill(cmask, 0.0) * rl).sum() + (prog * rp).sum()
                        outs += [logit, prog]
                    else:
                        logit, (hh, cc), _ = dec(imgs[t], ap, cands[t], hh, cc, ctx, ctx_mask)
                        total = total + (logit * rl).sum()
                        outs += [logit]
                    ap = cands[t][:, 0]
                total = total + (hh * rh).sum() + (cc * rh).sum()
                total.backward()
                torch.cuda.synchronize()
                per_rollout.append(([o.detach().clone() for o in outs], {n: p.grad.clone() for n, p in dec.named_parameters()},
                                    [ctx.grad.clone(), h.grad.clone(), c.grad.clone()]))
            res.append(per_rollout)
            if deferred:
                groups = 3 if agent == "monitor" else 1          # the step + the BN-MLP's two calls per step (B and B*C rows)
                assert F_.ROLLOUT_WGRADS.stats == [3 * T * groups, 3 * groups], F_.ROLLOUT_WGRADS.stats
                assert F_.ROLLOUT_WGRADS.stale_dropped == stale0 + 1
            else:
                assert F_.ROLLOUT_WGRADS.stats == [0, 0]
    finally:
        F_.set_rollout_wgrads(False)
        F_.set_grad_in_place(False)
    for rollout in range(2):
        a, b = res[1][rollout], res[0][rollout]
        for i, (x, y) in enumerate(zip(a[0], b[0])):
            assert torch.equal(x, y), f"rollout {rollout} output {i}"
        gscale = max(v.abs().max().item() for v in b[1].values())
        for n in b[1]:
            # The input BatchNorm's bias (mlp.0.bias) has a gradient of zero in exact arithmetic: both sides hold rounding noise, and
            # since round 5 not the same noise -- the rollout-level path forms it as sum_n db[n] W[n,k] from the first layer's bias
            # gradient and the fp32 weights (vln_bn0_grads_from_wgrad), the per-step path as the column sums of dz W, whose bf16-mode
            # product carries 2^-16 per term.  Judged on 1e-2 of the module's largest gradient like every exact zero.
            tol_n = 5e-4 if n.endswith("mlp.0.bias") else 2e-5
            check(a[1][n], b[1][n], tol_n, f"grad[{n}]", floor=grad_floor(n, gscale))
        for i, (x, y) in enumerate(zip(a[2], b[2])):
            assert torch.equal(x, y), f"rollout {rollout} input grad {i}"
    for n in res[1][0][1]:                                      # the second rollout reproduces the first (slots reused correctly)
        assert torch.equal(res[1][0][1][n], res[1][1][1][n]), n
        assert torch.equal(res[1][0][1][n], res[1][2][1][n]), f"{n}: after a stale queued flush"


@pytest.mark.parametrize("cdt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(128, 1024), (24, 168), (40, 300)], ids=["B128xC8", "one_chunk_each", "ragged_chunks"])
@pytest.mark.usefixtures("split_wgrads")
def test_bn_mlp_two_batches_in_one_call_equal_two_calls(vln, cdt, shape):
    """MLPwithBN.forward_pair (vln_bn_mlp.R1 > 0): the reference projects the previous action (B rows) and the candidates
    (B*C rows) with the same BN-MLP in two calls (policy.py:140-149); the fused call normalises each batch with its OWN
    statistics, updates the running statistics twice in that order, and runs the Linear layers once over all rows.  Against
    the two calls: outputs, running statistics and counters, input-side nothing (features carry no gradient), every parameter
    gradient -- in training mode with the MLP's dropout on (same masks: offsets consumed in the same order, indexed per batch)
    and in eval mode."""
    R1, R2 = shape
    F, dims = 256, [32, 128]
    g = torch.Generator().manual_seed(11)
    x1 = torch.randn(R1, F, generator=g).abs().to(DEV); x2 = torch.randn(R2, F, generator=g).abs().to(DEV)
    rz = (torch.rand(R2, generator=g) < 0.2).to(DEV)
    r1 = torch.randn(R1, dims[-1], generator=g).to(DEV); r2 = torch.randn(R2, dims[-1], generator=g).to(DEV)
    torch.manual_seed(3)
    from vln_amd.decoders import MLPwithBN
    make = lambda: MLPwithBN(F, dims, use_bn=True, dropout=0.3, relu=True)
    sd = {k: v.clone() for k, v in make().state_dict().items()}
    res = {}
    for mode in ("two", "pair", "pair_cat"):
        mlp = make(); mlp.load_state_dict(sd); mlp.to(DEV)
        mlp.inputs_in_place = mode != "pair_cat"                  # round 5: both batches read where they are (vln_bn_mlp.x2) / concatenated first
        n_set = 0
        for m_ in mlp.modules():
            if hasattr(m_, "compute_dtype"):
                m_.compute_dtype = cdt; n_set += 1
        assert n_set >= len(dims)                                 # every Linear streams its weights in `cdt`
        out = []
        for training in (True, True, False):                     # two training calls (running statistics move twice), one eval
            mlp.train(training)
            for p in mlp.parameters():
                p.grad = None
            if mode == "two":
                y1 = mlp(x1); y2 = mlp(x2, row_zero=rz)
            else:
                y1, y2 = mlp.forward_pair(x1, x2, row_zero2=rz)
            if training:
                ((y1 * r1).sum() + (y2 * r2).sum()).backward()
            torch.cuda.synchronize()
            out.append((y1.detach().clone(), y2.detach().clone(), {n: b.clone() for n, b in mlp.named_buffers()},
                        {n: p.grad.clone() for n, p in mlp.named_parameters()} if training else {}))
        res[mode] = out
    for i, (a, b) in enumerate(zip(res["pair"], res["two"])):
        check(a[0], b[0], 2e-5, f"call {i} y1"); check(a[1], b[1], 2e-5, f"call {i} y2")
        assert float(a[1][rz].abs().sum()) == 0.0
        for n in b[2]:
            if "num_batches_tracked" in n:
                assert torch.equal(a[2][n], b[2][n]), n
            else:
                check(a[2][n], b[2][n], 2e-5, f"call {i} buffer {n}")
        gscale = max([v.abs().max().item() for v in b[3].values()] + [1e-30])
        for n in b[3]:
            check(a[3][n], b[3][n], 5e-5, f"call {i} grad[{n}]", floor=grad_floor(n, gscale))
    # the in-place form reads the same numbers through two pointers: bit-identical to the concatenated copy
    for i, (a, b) in enumerate(zip(res["pair"], res["pair_cat"])):
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]), f"call {i}: outputs differ between the in-place and the concatenated input"
        for n in b[2]:
            assert torch.equal(a[2][n], b[2][n]), f"call {i}: buffer {n}"
        for n in b[3]:
            assert torch.equal(a[3][n], b[3][n]), f"call {i}: grad[{n}]"
    # ... and, like a tensor autograd saved, the second batch must not change between the forward and the backward
    mlp = make(); mlp.load_state_dict(sd); mlp.to(DEV).train()
    x2m = x2.clone()
    y1, y2 = mlp.forward_pair(x1, x2m, row_zero2=rz)
    x2m.mul_(2.0)
    with pytest.raises(RuntimeError, match="modified in place"):
        ((y1 * r1).sum() + (y2 * r2).sum()).backward()


@pytest.mark.parametrize("cdt", [torch.float32, torch.bfloat16])
@pytest.mark.usefixtures("split_wgrads")
def test_monitor_merged_projections_equal_two_bn_mlp_calls(vln, cdt):
    """MonitorDecoder.merge_projections: the step's two BN-MLP calls as one two-batch call -- logits, progress, state, attention
    maps, BatchNorm running statistics and every gradient of a three-step chain against the reference's two calls per step
    (same dropout masks), also with the parameter gradients formed once per rollout (RolloutWgrads: one group less)."""
    T, B, C, L, H, M, F = 3, 24, 7, 20, 64, 128, 256
    g = torch.Generator().manual_seed(321)
    ctx0 = torch.randn(B, L, H, generator=g); h00 = torch.randn(B, H, generator=g) * 0.5; c00 = torch.randn(B, H, generator=g) * 0.5
    a_prev = torch.randn(B, F, generator=g).abs().to(DEV); cands = [torch.randn(B, C, F, generator=g).abs().to(DEV) for _ in range(T)]
    lens = torch.randint(5, L + 1, (B,), generator=g); ctx_mask = (torch.arange(L)[None, :] >= lens[:, None]).to(DEV)
    nc = torch.randint(2, C + 1, (B,), generator=g); cmask = (torch.arange(C)[None, :] >= nc[:, None]).to(DEV)
    rl = torch.randn(B, C, generator=g).to(DEV); rh = torch.randn(B, H, generator=g).to(DEV); rp = torch.randn(B, generator=g).to(DEV)
    torch.manual_seed(7)
    make = lambda: vln.MonitorDecoder(H, 0.5, L, mlp_dims=[32, M], action_embed_size=F, feature_size=F, compute_dtype=cdt)
    sd = {k: v.clone() for k, v in make().state_dict().items()}
    F_ = vln.functional
    res = []
    try:
        for merged, rollout_wgrads in ((False, False), (True, False), (True, True)):
            F_.set_grad_in_place(rollout_wgrads); F_.set_rollout_wgrads(rollout_wgrads)
            F_.ROLLOUT_WGRADS.stats[:] = [0, 0]
            dec = make(); dec.load_state_dict(sd); dec.to(DEV).train()
            dec.merge_projections = merged
            for p in dec.parameters():
                p.grad = torch.zeros_like(p)
            ctx = ctx0.to(DEV).requires_grad_(True); h = h00.to(DEV).requires_grad_(True); c = c00.to(DEV).requires_grad_(True)
            hh, cc, ap, total, outs = h, c, a_prev, 0.0, []
            for t in range(T):
                (logit, prog), (hh, cc), (ww, mw) = dec(None, ap, cands[t], hh, cc, ctx, ctx_mask, cmask)
                total = total + (logit.masked_fill(cmask, 0.0) * rl).sum() + (prog * rp).sum()
                outs += [logit, prog, ww, mw]
                ap = cands[t][:, 0]
            total = total + (hh * rh).sum() + (cc * rh).sum()
            total.backward()
            torch.cuda.synchronize()
            if rollout_wgrads:
                assert F_.ROLLOUT_WGRADS.stats == [2 * T, 2], F_.ROLLOUT_WGRADS.stats      # two groups: the step, the ONE BN-MLP call
            res.append(([o.detach().clone() for o in outs + [hh, cc]], {n: p.grad.clone() for n, p in dec.named_parameters()},
                        [ctx.grad.clone(), h.grad.clone(), c.grad.clone()], {n: b.clone() for n, b in dec.named_buffers()}))
    finally:
        F_.set_rollout_wgrads(False); F_.set_grad_in_place(False)
    ref = res[0]
    gscale = max(v.abs().max().item() for v in ref[1].values())
    for k, got in enumerate(res[1:]):
        for i, (a, b) in enumerate(zip(got[0], ref[0])):
            check(a, b, 5e-5, f"variant {k} output {i}")
        for n in ref[1]:
            check(got[1][n], ref[1][n], 1e-4, f"variant {k} grad[{n}]", floor=grad_floor(n, gscale))
        for i, (a, b) in enumerate(zip(got[2], ref[2])):
            check(a, b, 1e-4, f"variant {k} input grad {i}")
        for n in ref[3]:
            if "num_batches_tracked" in n:
                assert torch.equal(got[3][n], ref[3][n]), n
            elif ref[3][n].is_floating_point():
                check(got[3][n], ref[3][n], 2e-5, f"variant {k} buffer {n}")


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_follower_iteration_with_batch_only_work_hoisted_equals_the_per_step_form(vln, dtype):
    """Round 6 (trainers.FollowerIteration): under teacher forcing the previous-action rows and the candidates' projection
    `linear_act(a_t_cands)` of every step depend on the batch only; one launch each up front (`ops.select_rows_multi`,
    `AttnDecoderLSTM.project_candidates` + `vln_follower_step.context_ready`) and the rollout's mean CE as one launch each way
    (`RolloutCE.mean_per_step`) give the losses and gradients of the per-step form (another K split in the projection, another
    summation order in the loss: 2e-5 of each tensor's range)."""
    from parity import check
    dev_ = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    B, L, T, C, F = 16, 12, 4, 6, 96
    tokens = torch.randint(4, 60, (B, L), generator=g)
    lens = torch.sort(torch.randint(4, L + 1, (B,), generator=g), descending=True).values; lens[0] = L
    for i, n in enumerate(lens.tolist()):
        tokens[i, n:] = 0
    steps = []
    for t in range(T):
        ncand = torch.randint(2, C + 1, (B,), generator=g)
        cmask = torch.arange(C)[None, :] >= ncand[:, None]
        steps.append(dict(img=(torch.randn(B, 36, F, generator=g).abs() * 0.5).to(dev_), cmask=cmask.to(dev_),
                          cand=(torch.randn(B, C, F, generator=g).abs() * 0.5 * (~cmask)[..., None]).to(dev_),
                          target=(torch.rand(B, generator=g) * ncand.float()).long().to(dev_)))
    batch = dict(tokens=tokens.to(dev_), lens32=lens.to(dev_, torch.int32), steps=steps)
    res = []
    for hoist in (False, True):
        torch.manual_seed(3)
        it = vln.trainers.FollowerIteration(dev_, dtype, vocab=60, embed=32, hidden=64, feature_size=F, lr=1e-3, graph=False, rollout_ce=hoist)
        it.enc.deterministic_embedding_grad = True
        it.opt_e.lr = it.opt_d.lr = 0.0          # keep the weights: Adam's g / sqrt(v) turns last-bit gradient noise into +-lr steps
        it.load(batch)
        assert ("cand_all" in it.live) and it.hoist_batch_only_work == hoist
        losses_ = [it.iteration().detach().clone() for _ in range(3)]        # (three different dropout draws)
        torch.cuda.synchronize()
        res.append((torch.stack(losses_), it.opt_e.flat_g.clone(), it.opt_d.flat_g.clone()))
    # bf16 mode forms its weight gradients from operands rounded to bf16: a 1e-7 change of an activation flips roundings of 2^-9,
    # i.e. 1e-4 .. 1e-3 of a gradient tensor's range between two equally valid runs (the mode's bound against the oracle is 1e-2)
    tol = 2e-5 if dtype == torch.float32 else 3e-3
    check(res[1][0], res[0][0], 2e-5, "losses of three iterations")
    check(res[1][1], res[0][1], tol, "encoder gradients of the third iteration")
    check(res[1][2], res[0][2], tol, "decoder gradients of the third iteration")
