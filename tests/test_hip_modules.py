"""GPU: the drop-in nn.Modules (HIP path through the C ABI) against
  (a) the committed golden vectors captured from the reference, loading the
      reference state_dict with strict=True (checks key/shape compatibility),
  (b) the CPU oracle at BASELINE sizes with dropout ON, injecting the exact
      Philox masks the kernels used.
Tolerances (north_star): fp32 1e-4, bf16 1e-2 for outputs AND gradients (max-abs error relative to the tensor's
max, tests/parity.py); the achieved errors are printed at the end of the run."""
import os

import pytest
import torch

from conftest import load_golden
from parity import check, check_grads, rel_err, bf16_weights, bf16_round_st, same_bf16_grad_tol, grad_floor, FP32, BF16, SAME_BF16

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def vln():
    import vln_amd
    vln_amd._lib.load()
    return vln_amd


def dev(d):
    return {k: v.to(DEV) for k, v in d.items()}


@pytest.mark.parametrize("name", ["encoder_envdrop", "encoder_follower", "encoder_monitor"])
def test_encoder_golden(vln, name):
    G = load_golden(name)
    cfg, I = G["cfg"], dev(G["inp"])
    enc = vln.EncoderLSTM(int(cfg["vocab"]), int(cfg["E"]), int(cfg["H"]), 0, 0.5, bool(cfg["bidir"]), int(cfg["layers"]))
    enc.load_state_dict(G["param"], strict=True)
    enc.to(DEV).eval()
    ctx, h, c = enc(I["tokens"], G["inp"]["lengths"])
    check(ctx, G["out"]["ctx"], 1e-4, "ctx"); check(h, G["out"]["h"], 1e-4, "h"); check(c, G["out"]["c"], 1e-4, "c")
    for i, n in enumerate(G["inp"]["lengths"].tolist()):
        if n < ctx.shape[1]:
            assert ctx[i, n:].abs().max().item() == 0.0
    loss = (ctx * I["r1"]).sum() + (h * I["r2"]).sum() + (c * I["r3"]).sum()
    loss.backward()
    for n, p in enc.named_parameters():
        check(p.grad, G["grad"][n], 1e-4, f"grad[{n}]")


@pytest.mark.parametrize("name", ["envdrop_step", "envdrop_chain3"])
def test_envdrop_golden(vln, name):
    G = load_golden(name)
    cfg, I = G["cfg"], dev(G["inp"])
    dec = vln.EnvDropDecoder(int(cfg["H"]), 0.5, 0.3, int(cfg["AE"]), int(cfg["ANG"]), int(cfg["IMG"]) + int(cfg["ANG"]))
    dec.load_state_dict(G["param"], strict=True)
    dec.to(DEV).eval()
    ctx = I["ctx"].clone().requires_grad_(True)
    h_tilde = I["h_tilde0"].clone().requires_grad_(True); c = I["c0"].clone().requires_grad_(True)
    ht0, c0 = h_tilde, c
    h_t = torch.zeros_like(c)
    loss = 0.
    for t in range(int(cfg["steps"])):
        logit, (h_t, c), h_tilde = dec(I[f"a{t}"], I[f"img{t}"].clone(), I[f"cand{t}"].clone(), h_tilde, h_t, c, ctx,
                                       I["ctx_mask"], False)
        check(logit, G["out"][f"logit{t}"], 1e-4, f"logit{t}")
        check(h_t, G["out"][f"h1_{t}"], 1e-4, "h1"); check(c, G["out"][f"c1_{t}"], 1e-4, "c1")
        check(h_tilde, G["out"][f"h_tilde{t}"], 1e-4, "h_tilde")
        loss = loss + (logit * I[f"rl{t}"]).sum() + (h_t * I[f"rh{t}"]).sum() * 0.1
    loss = loss + (h_tilde * I["rf"]).sum() + (c * I["rc"]).sum()
    check(loss, G["out"]["loss"], 1e-4, "loss")
    loss.backward()
    for n, p in dec.named_parameters():
        check(p.grad, G["grad"][n], 1e-4, f"grad[{n}]")
    check(ctx.grad, G["grad"]["ctx"], 1e-4, "dctx")
    check(ht0.grad, G["grad"]["h_tilde0"], 1e-4, "dh_tilde0"); check(c0.grad, G["grad"]["c0"], 1e-4, "dc0")


def test_envdrop_inplace_logit_mask_and_stop(vln):
    """Caller-side in-place masked_fill_ on the returned logits must flow through autograd; STOP logit == 0."""
    G = load_golden("envdrop_step")
    cfg, I = G["cfg"], dev(G["inp"])
    dec = vln.EnvDropDecoder(int(cfg["H"]), 0.5, 0.3, int(cfg["AE"]), int(cfg["ANG"]), int(cfg["IMG"]) + int(cfg["ANG"]))
    dec.load_state_dict(G["param"]); dec.to(DEV).eval()
    logit, _, _ = dec(I["a0"], I["img0"].clone(), I["cand0"].clone(), I["h_tilde0"], None, I["c0"], I["ctx"], I["ctx_mask"])
    lens = (5, 4, 3, 2)
    for i, n in enumerate(lens):
        assert logit[i, n - 1].item() == 0.0
    cmask = torch.arange(5, device=DEV)[None, :] >= torch.tensor(lens, device=DEV)[:, None]
    logit.masked_fill_(cmask, -float("inf"))
    tgt = torch.tensor([1, 3, 2, -1], device=DEV)
    loss = torch.nn.functional.cross_entropy(logit, tgt, ignore_index=-1, reduction="sum")
    loss.backward()
    from oracle import torch_port as O
    P = {k: v.clone().requires_grad_(True) for k, v in G["param"].items()}
    J = G["inp"]
    lo, *_ = O.envdrop_step(P, J["a0"], J["img0"], J["cand0"], J["h_tilde0"], J["c0"], J["ctx"], J["ctx_mask"])
    ref = O.masked_cross_entropy(lo, tgt.cpu(), cmask.cpu(), "sum")
    check(loss, ref, 1e-4, "ce loss")
    ref.backward()
    for n, p in dec.named_parameters():
        check(p.grad, P[n].grad, 1e-4, f"grad[{n}]")


def test_no_grad_inference_path_matches_training_graph_path(vln):
    """Greedy evaluation (BaseAgent.test, feedback="argmax") runs the decoder under no_grad / without building the
    stash: the plain C call must give the same numbers as the autograd-wrapped one."""
    G = load_golden("envdrop_chain3")
    cfg, I = G["cfg"], dev(G["inp"])
    dec = vln.EnvDropDecoder(int(cfg["H"]), 0.5, 0.3, int(cfg["AE"]), int(cfg["ANG"]), int(cfg["IMG"]) + int(cfg["ANG"]))
    dec.load_state_dict(G["param"], strict=True); dec.to(DEV).eval()
    outs = []
    for grad in (True, False):
        with torch.set_grad_enabled(grad):
            h_tilde, c = I["h_tilde0"], I["c0"]
            seq = []
            for t in range(3):
                logit, (h1, c), h_tilde = dec(I[f"a{t}"], I[f"img{t}"].clone(), I[f"cand{t}"].clone(), h_tilde, None, c,
                                              I["ctx"], I["ctx_mask"])
                seq += [logit.detach().clone(), h1.detach().clone(), h_tilde.detach().clone()]
            assert logit.requires_grad == grad
        outs.append(seq)
    for a, b in zip(*outs):
        assert torch.equal(a, b)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_ragged_and_unaligned_shapes(vln, dtype):
    """Edge cases: batch not a multiple of the 16-row MFMA block, hidden size not a multiple of 16, a single
    candidate (only the STOP slot), instructions longer than one wave (L > 64), length-1 instructions, unaligned
    feature sizes -- every kernel's guarded / scalar path, encoder (2 layers, per-step recurrence) + EnvDrop step."""
    from oracle import torch_port as O
    B, L, V, C, H, IMG, ANG, AE, E, vocab = 5, 70, 7, 1, 40, 72, 24, 12, 20, 30
    F = IMG + ANG
    g = torch.Generator().manual_seed(17)
    torch.manual_seed(17)
    enc = vln.EncoderLSTM(vocab, E, H, 0, 0.5, True, 2, compute_dtype=dtype).to(DEV).eval()
    dec = vln.EnvDropDecoder(H, 0.5, 0.3, AE, ANG, F, compute_dtype=dtype).to(DEV).eval()
    lens = torch.tensor([70, 33, 9, 2, 1])
    tokens = torch.zeros(B, L, dtype=torch.long)
    for i, n in enumerate(lens.tolist()):
        tokens[i, :n] = torch.randint(4, vocab, (n,), generator=g)
    mask = tokens == 0
    a = torch.sin(torch.randn(B, ANG, generator=g)); img = torch.randn(B, V, F, generator=g).abs()
    cand = torch.zeros(B, C, F)                                      # the only candidate is STOP: an all-zero row
    ctx, h, c = enc(tokens.to(DEV), lens)
    logit, (h1, c1), ht = dec(a.to(DEV), img.to(DEV), cand.to(DEV), h, None, c, ctx, mask.to(DEV))
    assert logit.shape == (B, 1) and logit.abs().max().item() == 0.0  # STOP logit is exactly 0
    r = torch.randn(B, H, generator=g)
    ((ht * r.to(DEV)).sum() + c1.sum() + (ctx ** 2).sum() * 0.01).backward()
    Pe = {k: v.detach().cpu().double().requires_grad_(True) for k, v in enc.state_dict().items()}
    Pd = {k: v.detach().cpu().double().requires_grad_(True) for k, v in dec.state_dict().items()}
    cx, ho, co = O.encoder_forward(Pe, tokens, lens.tolist(), num_layers=2, bidirectional=True)
    imo = img.double() if dtype == torch.float32 else img.bfloat16().double()
    lo, (h1o, c1o), hto, _ = O.envdrop_step(Pd, a.double(), imo, cand.double(), ho, co, cx, mask)
    ((hto * r.double()).sum() + c1o.sum() + (cx ** 2).sum() * 0.01).backward()
    tol = 1e-4 if dtype == torch.float32 else 1e-2
    check(ctx, cx, tol, "ctx"); check(ht, hto, tol, "h_tilde"); check(c1, c1o, tol, "c1")
    for mod, src in ((enc, Pe), (dec, Pd)):
        for n, p in mod.named_parameters():
            if p.grad is None:                                        # cand_attn gets no gradient from this loss
                assert src[n].grad is None or src[n].grad.abs().max().item() == 0.0, n
                continue
            ref = src[n].grad if src[n].grad is not None else torch.zeros_like(src[n])
            if ref.abs().max().item() < 1e-12:
                assert p.grad.abs().max().item() < 1e-6, n
            else:
                check(p.grad, ref, tol, f"grad[{n}]")


def test_critic_golden(vln):
    G = load_golden("critic")
    I = dev(G["inp"])
    cr = vln.Critic(64, 0.5)
    cr.load_state_dict(G["param"], strict=True); cr.to(DEV).eval()
    s = I["state"].clone().requires_grad_(True)
    v = cr(s)
    check(v, G["out"]["value"], 1e-4, "value")
    (v * I["r"]).sum().backward()
    for n, p in cr.named_parameters():
        check(p.grad, G["grad"][n], 1e-4, n)
    check(s.grad, G["grad"]["state"], 1e-4, "dstate")


# bf16 vs the UNROUNDED fp64 oracle = north_star's bound (1e-2) for every logit, state and gradient.  Every streamed weight carries
# a 2^-9 relative rounding; the two attention QUERY projections (visual_attn.linear_in: K = 2176 products in front of a softmax
# over 36 views; text_attn.linear_in) turn theirs into 1.4e-2 of d h_tilde / d visual_attn.linear_in, so the module streams those
# two matrices in fp32 BY DEFAULT (EnvDropDecoder.default_fp32_weights, round 4; +2.5 % time) and the default mode is held to
# 1e-2 with NO exception (measured worst 6.5e-3, profiles/round3_notes.md section 7).  The all-bf16 mode (round 3's default) stays
# available as `fp32_weights = frozenset()`: its two tensors over the bound are listed here and asserted at their measured size in
# test_envdrop_full_size_all_bf16_weights -- a documented NON-default mode, timed as bench.py's `all_bf16_weights_ms_per_step`.
ENVDROP_ALL_BF16_EXC = {"dh_tilde0": 3e-2, "grad[visual_attn.linear_in.weight]": 3e-2, "h1_": 2e-2, "dc0": 2e-2, "logit": 1.5e-2,
                        "h_tilde": 1.5e-2, "dctx": 1.5e-2, "grad[": 1.5e-2}
ENCODER_BF16_EXC = {}          # round 3 measured <= 2.6e-3 on every tensor: north_star's 1e-2 without exceptions


def _tol_for(exc, tol, what):
    for k, v in (exc or {}).items():
        if what == k or what.startswith(k):
            return v
    return tol


# Round 5, projected context (EnvDropDecoder.project_context): the text logits are taken on K = ctx W_in formed from the FP32 context,
# while the step's backward forms dq = sum_l dl[l] ctx[l] -- the dY rows of d text_attn.linear_in -- on the bf16 STREAM copy it
# holds for the d alpha dots.  A restatement that scores on the fp32 context (score_ctx) therefore sees that one gradient with
# ctx's 2^-9 rounding in it: measured 2.4e-3 (inside the default weight-gradient form's own 8e-3; it only shows in the "split" form).
PROJECTED_DW_IN = 4e-3


def _variants(compute_dtype, exc, projected=False):
    """[(name, tolerance, same-weights?, exceptions)]: fp32 = one oracle at 1e-4; bf16 = the oracle on the bf16-rounded
    weights the kernels stream at SAME_BF16, and the oracle on the unrounded masters at north_star's 1e-2 (tests/parity.py)."""
    if compute_dtype == torch.float32:
        return [("fp32", FP32, False, None)]
    same_exc = {"grad[": same_bf16_grad_tol()}
    if projected:
        same_exc = {"grad[text_attn.linear_in.weight]": max(same_bf16_grad_tol(), PROJECTED_DW_IN), **same_exc}
    return [("bf16 same-weights", SAME_BF16, True, same_exc), ("bf16 unrounded", BF16, False, exc)]


def _full_size_envdrop(vln, compute_dtype, T=3, train=True, fp32_weights=None, bf16_exc=None):
    from oracle import torch_port as O
    B, L, V, C, H, IMG, ANG, AE = 64, 80, 36, 8, 512, 2048, 128, 64
    F = IMG + ANG
    g = torch.Generator().manual_seed(2020)
    torch.manual_seed(2020)          # default parameter init comes from the global RNG: pin it (test-order independent)
    dec = vln.EnvDropDecoder(H, 0.5, 0.3, AE, ANG, F, compute_dtype=compute_dtype).to(DEV)
    dec.train(train)
    if fp32_weights is None:             # the module's default: the two attention query projections streamed in fp32
        fp32_weights = tuple(sorted(dec.fp32_weights))
    else:
        dec.fp32_weights = frozenset(fp32_weights)
    fp32_names = {"w_vin": "visual_attn.linear_in.weight", "w_tin": "text_attn.linear_in.weight", "w_tout": "text_attn.linear_out.weight",
                  "w_c": "cand_attn.weight", "w_cat": None}
    skip_round = ("act_embed.0.weight",) + tuple(fp32_names[k] for k in fp32_weights if fp32_names[k]) + \
        (("lstm.weight_ih", "lstm.weight_hh") if "w_cat" in fp32_weights else ())
    ctx = (torch.randn(B, L, H, generator=g) * 0.5)
    lens = torch.randint(8, L + 1, (B,), generator=g); lens[0] = L
    ctx_mask = torch.arange(L)[None, :] >= lens[:, None]
    ht = torch.tanh(torch.randn(B, H, generator=g)); c = torch.randn(B, H, generator=g) * 0.5
    ctx_d = ctx.to(DEV).requires_grad_(True); ht_d = ht.to(DEV).requires_grad_(True); c_d = c.to(DEV).requires_grad_(True)
    V_ = []
    for name, tol, same, exc in _variants(compute_dtype, {} if bf16_exc is None else bf16_exc, projected=dec.project_context and dec.split_attention):
        V_.append(dict(name=name, tol=tol, same=same, exc=exc, loss=0.,
                       P={k: v.detach().cpu().double().requires_grad_(True) for k, v in dec.state_dict().items()},
                       ctx=ctx.double().requires_grad_(True), ht=ht.double().requires_grad_(True), c=c.double().requires_grad_(True)))
        V_[-1]["state"] = (V_[-1]["ht"], V_[-1]["c"])
    if compute_dtype == torch.bfloat16:
        # RECORDED, not asserted: the same comparison with the oracle on the UN-rounded fp32 feature rows --
        # north_star's "same seeded inputs" read literally; the asserted variants give the oracle the bf16 rows the kernels stream
        # (the features are data the caller hands over in that form).  tol = 1: nothing can fail, everything lands in the report.
        V_.append(dict(name="bf16 unrounded, fp32 feature rows (recorded)", tol=1.0, same=False, exc={}, loss=0., feat32=True,
                       P={k: v.detach().cpu().double().requires_grad_(True) for k, v in dec.state_dict().items()},
                       ctx=ctx.double().requires_grad_(True), ht=ht.double().requires_grad_(True), c=c.double().requires_grad_(True)))
        V_[-1]["state"] = (V_[-1]["ht"], V_[-1]["c"])
    hd, cd = ht_d, c_d
    loss_d = 0.
    p, pf = (0.5, 0.3) if train else (0.0, 0.0)
    for t in range(T):
        a = torch.sin(torch.randn(B, ANG, generator=g) * 3)
        img = torch.randn(B, V, F, generator=g).abs() * 0.5; cand = torch.randn(B, C, F, generator=g).abs() * 0.5
        ncand = torch.randint(3, C + 1, (B,), generator=g)
        for i in range(B):
            cand[i, ncand[i] - 1:] = 0
        img_d, cand_d = img.to(DEV), cand.to(DEV)
        off = dec._step_counter + 1
        logit, (h1, cd), hd = dec(a.to(DEV), img_d, cand_d, hd, None, cd, ctx_d, ctx_mask.to(DEV))
        seed = dec.dropout_seed
        m = lambda site, n, pp: vln.ops.dropout_mask(n, seed, off * 8 + site, pp, DEV).cpu().double()
        drop = {"act": m(0, B * AE, p).view(B, AE), "hprev": m(1, B * H, p).view(B, H), "h1": m(2, B * H, p).view(B, H),
                "htilde": m(3, B * H, p).view(B, H)}
        img_o = O.feature_dropout(img.double(), m(4, B * V * IMG, pf).view(B, V, IMG), ANG)
        cand_o = O.feature_dropout(cand.double(), m(5, B * C * IMG, pf).view(B, C, IMG), ANG)
        # in-place contract: the caller's tensors now hold the dropped features
        check(img_d, img_o, 1e-6, "img in place"); check(cand_d, cand_o, 1e-6, "cand in place")
        img_32, cand_32 = img_o, cand_o
        if compute_dtype == torch.bfloat16:       # the features are DATA: both bf16 oracles see the rounded rows the kernels stream
            img_o = img_o.float().bfloat16().double(); cand_o = cand_o.float().bfloat16().double()
        rl = torch.randn(B, C, generator=g)
        loss_d = loss_d + (logit * rl.to(DEV)).sum() + h1.sum() * 0.01
        for v in V_:
            ho, co = v["state"]
            Pv = bf16_weights(v["P"], skip=skip_round) if v["same"] else v["P"]   # the 128 -> 64 embedding (and fp32_weights) run in fp32
            cx = bf16_round_st(v["ctx"]) if v["same"] else v["ctx"]          # the text attention streams a bf16 copy of ctx
            # ... and scores on K = ctx W_in formed from the FP32 context when the module projects the context (round 5)
            sc = v["ctx"] if (v["same"] and dec.scores_on_projected_context(ctx_d)) else None
            io_, co_ = (img_32, cand_32) if v.get("feat32") else (img_o, cand_o)
            lo, (h1o, co), ho, _ = O.envdrop_step(Pv, a.double(), io_, co_, ho, co, cx, ctx_mask, drop=drop, score_ctx=sc)
            v["state"] = (ho, co)
            for got, ref, what in ((logit, lo, f"logit{t}"), (h1, h1o, f"h1_{t}"), (hd, ho, f"h_tilde{t}")):
                check(got, ref, _tol_for(v["exc"], v["tol"], what), f"{v['name']}: {what}")
            v["loss"] = v["loss"] + (lo * rl.double()).sum() + h1o.sum() * 0.01
    loss_d = loss_d + hd.sum() * 0.1 + cd.sum() * 0.1
    loss_d.backward()
    for v in V_:
        ho, co = v["state"]
        (v["loss"] + ho.sum() * 0.1 + co.sum() * 0.1).backward()
        gmax = max(float(q.grad.abs().max()) for q in v["P"].values() if q.grad is not None)
        for n, prm in dec.named_parameters():
            check(prm.grad, v["P"][n].grad, _tol_for(v["exc"], v["tol"], f"grad[{n}]"), f"{v['name']}: grad[{n}]", floor=grad_floor(n, gmax))
        for got, ref, what in ((ctx_d.grad, v["ctx"].grad, "dctx"), (ht_d.grad, v["ht"].grad, "dh_tilde0"), (c_d.grad, v["c"].grad, "dc0")):
            check(got, ref, _tol_for(v["exc"], v["tol"], what), f"{v['name']}: {what}")


def test_envdrop_full_size_fp32_dropout_on(vln):
    _full_size_envdrop(vln, torch.float32)


def test_envdrop_full_size_fp32_eval(vln):
    _full_size_envdrop(vln, torch.float32, train=False)


def test_envdrop_full_size_bf16(vln):
    """BASELINE config 1 in the DEFAULT bf16 mode (bf16-streamed features / context / weights except the two attention query
    projections, fp32 accumulate, dropout on): vs the fp64 oracle on the SAME rounded weights (1e-4) and vs the fp64 oracle on the
    UNROUNDED parameters at north_star's 1e-2 for every logit, state and gradient -- no exceptions."""
    _full_size_envdrop(vln, torch.bfloat16)


def test_envdrop_full_size_all_bf16_weights(vln):
    """The NON-default all-bf16 mode (`fp32_weights = frozenset()`): same-weights oracle at 1e-4; against the unrounded oracle
    two tensors sit at 1.4e-2 (ENVDROP_ALL_BF16_EXC) -- why the default streams the query projections in fp32."""
    _full_size_envdrop(vln, torch.bfloat16, fp32_weights=(), bf16_exc=ENVDROP_ALL_BF16_EXC)


def test_encoder_full_size_bf16(vln):
    _encoder_full(vln, torch.bfloat16)


def test_encoder_full_size(vln):
    _encoder_full(vln, torch.float32)


def _encoder_full(vln, compute_dtype):
    from oracle import torch_port as O
    B, L, E, H, vocab = 64, 80, 256, 512, 992
    g = torch.Generator().manual_seed(7)
    torch.manual_seed(7)
    enc = vln.EncoderLSTM(vocab, E, H, 0, 0.5, True, 1, compute_dtype=compute_dtype).to(DEV).eval()
    lens = torch.sort(torch.randint(8, L + 1, (B,), generator=g), descending=True).values; lens[0] = L
    tokens = torch.zeros(B, L, dtype=torch.long)
    for i, n in enumerate(lens.tolist()):
        tokens[i, :n] = torch.randint(4, vocab, (n,), generator=g)
    ctx, h, c = enc(tokens.to(DEV), lens)
    r1, r2 = torch.randn(B, L, H, generator=g), torch.randn(B, H, generator=g)
    ((ctx * r1.to(DEV)).sum() + (h * r2.to(DEV)).sum() + c.sum()).backward()
    for name, tol, same, exc in _variants(compute_dtype, ENCODER_BF16_EXC):
        P = {k: v.detach().cpu().double().requires_grad_(True) for k, v in enc.state_dict().items()}
        Pv = bf16_weights(P, skip=("embedding.weight",)) if same else P        # embedding rows are gathered in fp32
        co, ho, cco = O.encoder_forward(Pv, tokens, lens.tolist(), num_layers=1, bidirectional=True)
        for got, ref, what in ((ctx, co, "ctx"), (h, ho, "h"), (c, cco, "c")):
            check(got, ref, _tol_for(exc, tol, what), f"{name}: {what}")
        ((co * r1.double()).sum() + (ho * r2.double()).sum() + cco.sum()).backward()
        gmax = max(float(q.grad.abs().max()) for q in P.values() if q.grad is not None)
        for n, prm in enc.named_parameters():
            check(prm.grad, P[n].grad, _tol_for(exc, tol, f"grad[{n}]"), f"{name}: grad[{n}]", floor=grad_floor(n, gmax))


@pytest.mark.parametrize("dtype,B,inproj", [(torch.float32, 64, False), (torch.bfloat16, 64, False), (torch.bfloat16, 144, False),
                                            (torch.float32, 192, False), (torch.bfloat16, 256, False),
                                            (torch.bfloat16, 64, True), (torch.float32, 64, True), (torch.bfloat16, 144, True)])
@pytest.mark.usefixtures("split_wgrads")
def test_persistent_recurrence_equals_per_step_launches(vln, dtype, B, inproj):
    """The single-launch persistent bi-LSTM (in-kernel cross-workgroup hand-off) must reproduce the per-step
    launch chain bit for bit, forward and backward, and report a clean status word.  B > 128 (round 6): the
    one-workgroup-per-(slice, direction, 16 rows) grid no longer fits the 256 CUs, and the launch runs in PASSES -- 144 rows = 9 row
    blocks -> 2 passes of 5 (the last pass one block short), 192 -> 2 x 6, 256 -> 2 x 8 -- instead of falling back to 2 x L per-step
    launches (encoder.hip: persist_passes).  inproj (round 6, an A/B option: EncoderLSTM.inproj): the persistent launch forms the input
    projection x_t W_ih^T + b itself, in gemm_nt's MFMA order -- still bit-identical to the chain that reads the GEMM's output."""
    lib = vln._lib.load()
    L, E, H, vocab = 80, 256, 512, 992
    g = torch.Generator().manual_seed(3)
    enc = vln.EncoderLSTM(vocab, E, H, 0, 0.5, True, 1, compute_dtype=dtype).to(DEV).train()
    lens = torch.sort(torch.randint(1, L + 1, (B,), generator=g), descending=True).values; lens[0] = L
    tokens = torch.zeros(B, L, dtype=torch.long)
    for i, n in enumerate(lens.tolist()):
        tokens[i, :n] = torch.randint(4, vocab, (n,), generator=g)
    r = torch.randn(B, L, H, generator=g).to(DEV)
    outs = []
    enc.inproj = inproj
    for persistent in (1, 0):
        lib.vln_set_persistent(persistent)
        enc._calls = 0                                  # same dropout stream for both runs
        enc.zero_grad(set_to_none=True)
        ctx, h, c = enc(tokens.to(DEV), lens)
        ((ctx * r).sum() + h.sum() + (c * c).sum()).backward()
        torch.cuda.synchronize()
        if persistent:
            assert enc.persistent_status() == 0
            if B > 128:       # the persistent path WAS taken: its launch ran in passes (a per-step chain leaves no hand-off tallies)
                assert enc.ran_persistent(), "B > 128 fell back to the per-step launch chain"
        outs.append([ctx.detach().clone(), h.detach().clone(), c.detach().clone()] +
                    [p.grad.detach().clone() for p in enc.parameters()])
    lib.vln_set_persistent(1)
    names = ["ctx", "h", "c"] + [n for n, _ in enc.named_parameters()]
    for n, a, b in zip(names, *outs):
        if n in ("ctx", "h", "c"):       # same MFMA order, same pointwise expressions: bit-identical forward
            assert torch.equal(a, b), n
        else:                            # backward: the two kernels may contract a*b+c differently (last bit of
            check(a, b, 2e-5, n)         # dgates) and the embedding scatter-add uses float atomics


@pytest.mark.parametrize("B", [64, 50, 144])
def test_encoder_weight_gradients_inside_the_bptt_launch_equal_the_packed_contraction(vln, B):
    """Round 6 (an A/B option, off by default: it measured slower -- EncoderLSTM.wgrad_inlaunch): in bf16 mode the bi-LSTM's own weight
    gradients d W_hh = sum_t dgates_t^T h_{t-1}, d W_ih = sum_t dgates_t^T x_t can be
    accumulated INSIDE the persistent BPTT launch by four extra waves per workgroup (vln_lstm_seq_bwd_w + vln_lstm_wgrad_reduce)
    instead of by the pack + contraction launches over the L * B rows (vln_wgrad_grouped, precision 2): the same bf16 x bf16
    products with fp32 accumulation in another summation order -- every other output of the backward bit for bit, the four weight
    gradients within 2e-5 of each other (relative to the tensor's maximum).  B = 50: a ragged last row block; B = 144: two passes
    (the partials are summed over a workgroup's row blocks in registers).  Gradients ACCUMULATE into existing .grad buffers like the
    packed form (second backward)."""
    lib = vln._lib.load()
    L, E, H, vocab = 80, 256, 512, 992
    g = torch.Generator().manual_seed(31)
    torch.manual_seed(31)
    enc = vln.EncoderLSTM(vocab, E, H, 0, 0.5, True, 1, compute_dtype=torch.bfloat16).to(DEV).train()
    lens = torch.sort(torch.randint(1, L + 1, (B,), generator=g), descending=True).values; lens[0] = L
    tokens = torch.zeros(B, L, dtype=torch.long)
    for i, n in enumerate(lens.tolist()):
        tokens[i, :n] = torch.randint(4, vocab, (n,), generator=g)
    r = torch.randn(B, L, H, generator=g).to(DEV)
    assert vln.ops.get_wgrad_precision() == "bf16"
    outs = []
    for inl in (False, True):
        enc.wgrad_inlaunch = inl
        enc._calls = 0
        enc.zero_grad(set_to_none=True)
        for rep in range(2):                            # the second backward accumulates into the first one's .grad tensors
            ctx, h, c = enc(tokens.to(DEV), lens)
            ((ctx * r).sum() + h.sum() + (c * c).sum()).backward()
        torch.cuda.synchronize()
        assert enc.persistent_status() == 0
        outs.append({n: p.grad.detach().clone() for n, p in enc.named_parameters()})
    vln._lib.check(lib.vln_persistent_check(), "vln_persistent_check")
    for n in outs[0]:
        a, b = outs[0][n], outs[1][n]
        if n.startswith("lstm.weight"):
            check(b, a, 2e-5, f"in-launch {n} (B = {B})")
            assert not torch.equal(a, torch.zeros_like(a))
        elif n != "embedding.weight":                   # (the embedding scatter-add uses float atomics)
            assert torch.equal(a, b), n


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_backward_recurrence_hand_off_in_the_xcd_l2_equals_the_write_through_hand_off(vln, dtype):
    """Round 5: when every workgroup of a dependency group of the backward recurrence verified (XCC_ID bits ORed into the group's
    flag line before the first arrival) that the group runs on ONE XCD, its partial-product stores stay in that XCD's L2 instead
    of being written through -- the same numbers by a faster route.  Three runs of the same backward: the default (one XCD per
    group -> plain stores), always write-through (tunable 14 = 1), and groups dealt ACROSS the XCDs in dispatch order (tunable
    7 = 1: the check finds several XCC_IDs and the stores stay write-through): bit-identical gradients, clean status words, twice
    in a row (the mask word is reset with the counters by the last workgroup through)."""
    lib = vln._lib.load()
    B, L, E, H, vocab = 64, 80, 256, 512, 992
    g = torch.Generator().manual_seed(31)
    enc = vln.EncoderLSTM(vocab, E, H, 0, 0.5, True, 1, compute_dtype=dtype).to(DEV).train()
    enc.deterministic_embedding_grad = True
    lens = torch.sort(torch.randint(1, L + 1, (B,), generator=g), descending=True).values; lens[0] = L
    tokens = torch.zeros(B, L, dtype=torch.long)
    for i, n in enumerate(lens.tolist()):
        tokens[i, :n] = torch.randint(4, vocab, (n,), generator=g)
    r = torch.randn(B, L, H, generator=g).to(DEV)
    import ctypes as C
    outs, tallies = [], []

    def tally(fwd=False):
        a, b = C.c_uint32(), C.c_uint32()
        ws = enc._sync_ws(torch.device(DEV), B, H // 2, 2)
        f = lib.vln_lstm_fwd_handoff_stats if fwd else lib.vln_lstm_handoff_stats
        vln._lib.check(f(ws[0], C.byref(a), C.byref(b)), "vln_lstm_handoff_stats")
        return a.value, b.value
    fouts, ftallies = [], []
    try:
        for t14, t7 in ((0, 0), (1, 0), (0, 1), (0, 0)):
            vln._lib.check(lib.vln_set_tunable(14, t14), "vln_set_tunable"); vln._lib.check(lib.vln_set_tunable(7, t7), "vln_set_tunable")
            for rep in range(2):
                enc._calls = 0
                enc.zero_grad(set_to_none=True)
                fbefore = tally(True)
                ctx, h, c = enc(tokens.to(DEV), lens)
                torch.cuda.synchronize()
                fafter = tally(True)
                fouts.append((ctx.detach().clone(), h.detach().clone(), c.detach().clone()))
                ftallies.append((t14, t7, fafter[0] - fbefore[0], fafter[1] - fbefore[1]))
                before = tally()
                ((ctx * r).sum() + h.sum() + (c * c).sum()).backward()
                torch.cuda.synchronize()
                after = tally()
                assert enc.persistent_status() == 0
                outs.append([p.grad.detach().clone() for p in enc.parameters()])
                tallies.append((t14, t7, after[0] - before[0], after[1] - before[1]))
    finally:
        lib.vln_set_tunable(14, 0); lib.vln_set_tunable(7, 0)
    vln._lib.check(lib.vln_persistent_check(), "vln_persistent_check")
    names = [n for n, _ in enc.named_parameters()]
    for k, o in enumerate(outs[1:]):
        for n, a, b in zip(names, outs[0], o):
            assert torch.equal(a, b), f"run {k + 1}: grad[{n}] differs from the first run's"
    # the FORWARD recurrence's granule hand-off takes the same decision the same way (round 5): same outputs bit for bit
    for k, o in enumerate(fouts[1:]):
        for a, b in zip(fouts[0], o):
            assert torch.equal(a, b), f"run {k + 1}: the forward's outputs differ from the first run's"
    for t14, t7, loc, span in ftallies:
        if t14:
            assert (loc, span) == (0, 0), ftallies
        elif t7:
            assert loc + span == 8 and span > 0, ftallies
        else:
            assert (loc, span) == (8, 0), ftallies
    # what the launches decided (vln_lstm_handoff_stats): 2 directions x 4 row blocks = 8 dependency groups per backward
    for t14, t7, loc, span in tallies:
        if t14:
            assert (loc, span) == (0, 0), tallies                   # switch off: no check, write-through
        elif t7:
            assert loc + span == 8 and span > 0, tallies            # groups dealt across the XCDs: found out, write-through
        else:
            assert (loc, span) == (8, 0), tallies                   # one XCD per group: verified, stores kept in its L2


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_encoder_dx_posted_into_the_weight_gradients_pack_launch_is_bit_identical(vln, dtype):
    """ops.linear_fwd_post / vln_linear_fwd_post (round 5): the encoder backward's d x = dgates W_ih rides as extra workgroups of the
    weight gradients' pack launch (same tiles, same kernel body as the stand-alone product).  Every parameter gradient -- the
    embedding's, which is what d x feeds -- equals the separate launches' bit for bit; in fp32 mode the weight gradients take the exact
    form, nobody takes the post and the flush issues it alone: the same again.  A second post while one is pending is refused."""
    B, L, E, H, vocab = 64, 80, 256, 512, 992
    g = torch.Generator().manual_seed(7)
    enc = vln.EncoderLSTM(vocab, E, H, 0, 0.5, True, 1, compute_dtype=dtype).to(DEV).train()
    enc.deterministic_embedding_grad = True
    lens = torch.sort(torch.randint(1, L + 1, (B,), generator=g), descending=True).values; lens[0] = L
    tokens = torch.zeros(B, L, dtype=torch.long)
    for i, n in enumerate(lens.tolist()):
        tokens[i, :n] = torch.randint(4, vocab, (n,), generator=g)
    r = torch.randn(B, L, H, generator=g).to(DEV)
    outs, fwds = [], []
    for post in (True, False, True):
        enc.dx_with_wgrads = post
        enc.layout_with_bridge = post          # ... and the context's two layout changes in the bridge products' launches (vln_layout_post)
        enc._calls = 0
        enc.zero_grad(set_to_none=True)
        ctx, h, c = enc(tokens.to(DEV), lens)
        fwds.append((ctx.detach().clone(), h.detach().clone(), c.detach().clone()))
        ((ctx * r).sum() + h.sum() + (c * c).sum()).backward()
        torch.cuda.synchronize()
        outs.append([p.grad.detach().clone() for p in enc.parameters()])
    names = [n for n, _ in enc.named_parameters()]
    for o in outs[1:]:
        for n, a, b in zip(names, outs[0], o):
            assert torch.equal(a, b), f"grad[{n}] differs between the posted and the stand-alone d x product"
    for f in fwds[1:]:
        for a, b in zip(fwds[0], f):
            assert torch.equal(a, b), "the forward's outputs differ between the posted and the stand-alone layout change"
    assert float(outs[0][names.index("embedding.weight")].abs().sum()) > 0
    x = torch.randn(128, 64, device=DEV); w = torch.randn(32, 64, device=DEV); y = torch.empty(128, 32, device=DEV)
    vln.ops.linear_fwd_post(x, w, y)
    with pytest.raises(vln.VlnError, match="pending"):
        vln.ops.linear_fwd_post(x, w, y)
    vln.ops.linear_fwd_post_flush(torch.device(DEV), 128, 32)
    assert torch.equal(y, vln.ops.linear_fwd(x, w))
    # a caller that raised between its post and its flush: the next call that posts forgets what it left (vln_posted_drop)
    y2 = torch.full((128, 32), 3.0, device=DEV)
    vln.ops.linear_fwd_post(x, w, y2)
    assert vln._lib.load().vln_posted_drop() == 1 and vln._lib.load().vln_posted_drop() == 0
    vln.ops.linear_fwd_post_flush(torch.device(DEV), 128, 32)          # nothing pending: no launch
    torch.cuda.synchronize()
    assert bool((y2 == 3.0).all())
    vln._lib.check(vln._lib.load().vln_persistent_check(), "vln_persistent_check")


def test_counter_backward_on_a_dirty_or_foreign_sync_workspace(vln):
    """the counter-protocol backward leaves its group counters zero itself and skips the fill launch -- which
    only holds for a header it left behind.  (a) a caller-supplied workspace whose header was never zeroed, (b) a header the
    counter-protocol FORWARD of another mode wrote: both must get the fill (encoder.hip header_clean), i.e. no timeout and the
    same gradients as on a clean workspace."""
    lib = vln._lib.load()
    B, L, E, H, vocab = 32, 12, 64, 512, 200
    g = torch.Generator().manual_seed(5)
    lens = torch.sort(torch.randint(1, L + 1, (B,), generator=g), descending=True).values; lens[0] = L
    tokens = torch.zeros(B, L, dtype=torch.long)
    for i, n in enumerate(lens.tolist()):
        tokens[i, :n] = torch.randint(4, vocab, (n,), generator=g)
    r = torch.randn(B, L, H, generator=g).to(DEV)
    torch.manual_seed(9)
    enc = vln.EncoderLSTM(vocab, E, H, 0, 0.0, True, 1).to(DEV).train()
    enc.deterministic_embedding_grad = True

    def run():
        enc.zero_grad(set_to_none=True)
        ctx, h, c = enc(tokens.to(DEV), lens)
        ((ctx * r).sum() + h.sum() + (c * c).sum()).backward()
        torch.cuda.synchronize()
        vln._lib.check(lib.vln_persistent_check(), "vln_persistent_check")
        assert enc.persistent_status() == 0
        return [p.grad.detach().clone() for p in enc.parameters()]

    names = [n for n, _ in enc.named_parameters()]

    def same(tag, got):
        bad = [(n, float((a - b).abs().max())) for n, a, b in zip(names, ref, got) if not torch.equal(a, b)]
        assert not bad, f"{tag}: gradients differ from the clean-workspace run: {bad}"

    ref = run()
    # (a) a fresh, larger workspace the library has never seen, header full of garbage (granule area zero as documented)
    need = int(lib.vln_lstm_sync_ws_bytes(B, H // 2, 2))
    w = torch.zeros((need + 3) // 4 + 64, dtype=torch.int32, device=DEV)
    w[:2048] = 0x01010101
    w[32] = 0
    enc._sync_buf, enc._sync_mode = w, None
    same("dirty header", run())
    # (a') round 6: the SAME address with a header the caller has written to -- what torch's allocator produces when it hands a freed
    # buffer's address to a new one (seen once in the full suite: the library remembered the address as clean, skipped the fill and the
    # BPTT ran on garbage counters).  The module announces a buffer it has not used before with vln_lstm_sync_ws_forget.
    w[:2048] = 0x01010101
    w[32] = 0
    enc._sync_mode = None
    same("the same address, dirtied by the caller", run())
    # (b) mode 2 (counter forward + counter backward): a forward WITHOUT its backward leaves counters behind, then mode 1
    try:
        lib.vln_set_persistent(2)
        with torch.no_grad():
            enc(tokens.to(DEV), lens)
        torch.cuda.synchronize()
    finally:
        lib.vln_set_persistent(1)
    same("a foreign forward's counters", run())
    same("its own header", run())                  # and the header it left itself needs no fill


def test_fp32_weight_gradients_split_planes_against_the_exact_fp32_mfma(vln):
    """Round 6: fp32 compute mode forms its weight gradients on the bf16 MFMA with both operands split hi + lo (three products, one
    packed contraction per rollout) by default; `ops.set_wgrad_precision_fp32("exact")` keeps the fp32 MFMA of rounds 1-5.  The two
    forms of every parameter gradient of an EnvDrop IL iteration agree to 5e-5 of the tensor's max (the oracle tests hold 1e-4
    with the default)."""
    from parity import check
    dev_ = torch.device(DEV)
    tape = vln.synthetic.tape_to(vln.synthetic.make_tape(16, 24, 3, 6, seed=4), dev_)
    default = "exact" if os.environ.get("VLN_WGRAD_FP32") == "exact" else "split"       # (the process-wide override keeps rounds 1-5's form)
    assert vln.ops.get_wgrad_precision_fp32() == default
    res = {}
    try:
        for mode in ("split", "exact"):
            vln.ops.set_wgrad_precision_fp32(mode)
            torch.manual_seed(11)
            ag = vln.trainers.EnvDropILIteration(dev_, torch.float32, 1)
            ag.enc._calls = 0; ag.dec._step_counter = 0
            ag.enc.deterministic_embedding_grad = True
            ag.opt.lr = 0.0
            loss = ag.iteration(tape)
            torch.cuda.synchronize()
            res[mode] = (loss.detach().clone(), {n: p.grad.detach().clone() for m in (ag.enc, ag.dec) for n, p in m.named_parameters()})
    finally:
        vln.ops.set_wgrad_precision_fp32(default)
    assert torch.equal(res["split"][0], res["exact"][0])          # the forward does not depend on it
    differ = 0
    for n, g in res["exact"][1].items():
        check(res["split"][1][n], g, 5e-5, f"grad[{n}] split planes vs exact fp32 MFMA")
        differ += int(not torch.equal(res["split"][1][n], g))
    assert differ > 0                                            # (the switch does select another kernel)


def test_training_iteration_side_stream_overlap_is_transparent(vln):
    """trainers.EnvDropILIteration.iteration with the deferred weight-gradient GEMMs on a side stream (gradients accumulate into the
    flat bucket views) must give exactly the gradients of the serial configuration; store- and tensor-fed features
    must give the same loss when dropout is off."""
    dev_ = torch.device(DEV)
    cpu_tape = vln.synthetic.make_tape(16, 24, 3, 6, seed=4)
    tape = vln.synthetic.tape_to(cpu_tape, dev_)
    res = []
    for overlap in (True, False):
        torch.manual_seed(11)
        ag = vln.trainers.EnvDropILIteration(dev_, torch.float32, 1)
        ag.dec.overlap_wgrads = overlap
        ag.enc._calls = 0; ag.dec._step_counter = 0
        ag.enc.deterministic_embedding_grad = True    # no float atomics anywhere: every gradient must match bit for bit
        ag.opt.lr = 0.0                                   # keep the weights: RMSprop's g/sqrt(g^2) amplifies last-bit noise
        loss = ag.iteration(tape)
        torch.cuda.synchronize()
        res.append((loss.detach().clone(), [p.grad.detach().clone() for p in ag.dec.parameters()]))
    assert torch.equal(res[0][0], res[1][0])
    for a, b in zip(res[0][1], res[1][1]):
        assert torch.equal(a, b)
    # feature store (indices -> gather) vs pre-built tensors: same episode data, dropout off -> same loss
    store_tape = vln.synthetic.tape_to(cpu_tape, dev_, store_dtype=torch.float32)
    losses = []
    for tp in (tape, store_tape):
        torch.manual_seed(11)
        ag = vln.trainers.EnvDropILIteration(dev_, torch.float32, 1)
        ag.enc.eval(); ag.dec.eval(); ag.opt.lr = 0.0
        losses.append(ag.iteration(tp).detach().clone())
    check(losses[0], losses[1], 1e-6, "store vs tensor loss")


def test_step_graphs_with_device_side_dropout_offset_are_transparent(vln):
    """`EnvDropDecoder.step_graphs`: the step's Philox offset is read from device memory and each step is captured /
    replayed as a hipGraph (vln_envdrop_step.offset_dev).  Dropout ON: loss and every gradient must be bit-identical to
    the plain-launch path, over two iterations (second one exercises re-capture or replay)."""
    import ctypes
    dev_ = torch.device(DEV)
    tape = vln.synthetic.tape_to(vln.synthetic.make_tape(16, 24, 3, 6, seed=5), dev_)
    res = []
    for graphs in (True, False):
        torch.manual_seed(13)
        ag = vln.trainers.EnvDropILIteration(dev_, torch.bfloat16, 1)
        ag.dec.step_graphs = graphs
        ag.enc._calls = 0; ag.dec._step_counter = 0
        ag.enc.deterministic_embedding_grad = True    # no float atomics anywhere: every gradient must match bit for bit
        ag.opt.lr = 0.0
        out = []
        for _ in range(2):
            loss = ag.iteration(tape)
            torch.cuda.synchronize()
            # every gradient of both modules: all reductions on the path (incl. the embedding's) have a fixed order
            out.append((loss.detach().clone(), [p.grad.detach().clone() for p in list(ag.dec.parameters()) + list(ag.enc.parameters())]))
        res.append(out)
    for (la, ga), (lb, gb) in zip(res[0], res[1]):
        assert torch.equal(la, lb)
        for a, b in zip(ga, gb):
            assert torch.equal(a, b)
    st = (ctypes.c_int64 * 3)()
    vln._lib.load().vln_graph_stats(st)
    assert st[1] > 0                                  # steps were captured (replays need address-stable callers)


def test_rollout_arena_replays_step_graphs_with_identical_results(vln):
    """ops.RolloutArena: per-iteration buffers come back at the same addresses, so from the third iteration on every
    decoder step replays its hipGraph; losses and gradients equal the torch.empty / plain-launch configuration bit for
    bit (dropout on), and the arena must not be left active after an iteration."""
    import ctypes
    dev_ = torch.device(DEV)
    tape = vln.synthetic.tape_to(vln.synthetic.make_tape(16, 24, 3, 6, seed=6), dev_, store_dtype=torch.bfloat16)
    lib = vln._lib.load()
    st0, st1 = (ctypes.c_int64 * 3)(), (ctypes.c_int64 * 3)()
    res = []
    for arena in (True, False):
        torch.manual_seed(17)
        ag = vln.trainers.EnvDropILIteration(dev_, torch.bfloat16, 1, arena=arena)
        ag.enc._calls = 0; ag.dec._step_counter = 0
        ag.enc.deterministic_embedding_grad = True    # no float atomics anywhere: every gradient must match bit for bit
        tape["store"]._calls = 0                      # the store's feature-dropout stream restarts too
        ag.opt.lr = 0.0
        if arena:
            lib.vln_graph_stats(st0)
        out = []
        for _ in range(5):
            loss = ag.iteration(tape)
            torch.cuda.synchronize()
            # every gradient of both modules: all reductions on the path (incl. the embedding's) have a fixed order
            out.append((loss.detach().clone(), [p.grad.detach().clone() for p in list(ag.dec.parameters()) + list(ag.enc.parameters())]))
        if arena:
            lib.vln_graph_stats(st1)
            assert ag.arena.misses == 0
            assert ag.dec.plan_hits >= 3 * 3          # iterations 3..5 reuse the recorded step plans (host fast path)
        assert vln.ops._arena is None
        res.append(out)
    assert st1[0] - st0[0] >= 3 * 6                  # iterations 3..5: 3 steps x (fwd + bwd) replays each
    for (la, ga), (lb, gb) in zip(res[0], res[1]):
        assert torch.equal(la, lb)
        for a, b in zip(ga, gb):
            assert torch.equal(a, b)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_per_step_graphs_without_an_arena_equal_plain_launches(vln, dtype):
    """Round 6: the reference's UNCHANGED caller (feature tensors handed in every step, per-step masked CE, torch's own allocator, no
    arena, no iteration graph) gets its decoder steps replayed as hipGraphs too -- `EnvDropDecoder.step_graphs` is on by default; the
    step's dropout offset then lives in a device word and the C call memoises its launch chain by argument block, which repeats once
    the caching allocator has settled.  Same losses and gradients as plain launches, bit for bit, over six iterations (dropout on),
    and the later iterations really replay."""
    import ctypes
    dev_ = torch.device(DEV)
    cpu_tape = vln.synthetic.make_tape(16, 24, 3, 6, seed=16)
    tape = vln.synthetic.tape_to(cpu_tape, dev_)                   # explicit img / cand tensors per step: the reference's marshalling
    lib = vln._lib.load()
    st0, st1 = (ctypes.c_int64 * 3)(), (ctypes.c_int64 * 3)()
    res = []
    for graphs in (False, True):
        torch.manual_seed(23)
        ag = vln.trainers.EnvDropILIteration(dev_, dtype, 1, arena=False, rollout_ce=False)
        assert ag.dec.step_graphs is True                          # the module's default, arena or not
        ag.dec.step_graphs = graphs
        ag.enc._calls = 0; ag.dec._step_counter = 0
        ag.enc.deterministic_embedding_grad = True
        ag.opt.lr = 0.0
        if graphs:
            lib.vln_graph_stats(st0)
        out = []
        for _ in range(14):
            loss = ag.iteration(tape)
            torch.cuda.synchronize()
            # (host copies: a device clone kept per iteration would move every later allocation and no argument block could repeat)
            out.append((loss.detach().cpu(), [p.grad.detach().cpu() for p in list(ag.dec.parameters()) + list(ag.enc.parameters())]))
            del loss
        if graphs:
            lib.vln_graph_stats(st1)
        res.append(out)
    print(f"per-step graphs without an arena: {st1[0] - st0[0]} replays, {st1[1] - st0[1]} captures over 14 iterations of 3 steps")
    assert st1[0] - st0[0] >= 3                                    # the allocator settles: later iterations replay steps
    for (la, ga), (lb, gb) in zip(res[0], res[1]):
        assert torch.equal(la, lb)
        for a, b in zip(ga, gb):
            assert torch.equal(a, b)
    assert vln.EnvDropDecoder(64, 0.5, 0.3, 16, 32, 128).step_graphs is True      # the module's own default


def test_a_rollout_on_the_projected_context_survives_the_split_attention_being_switched_off(vln):
    """the projected-context (kctx) form is chosen once per rollout; if the four-workgroup attention is switched off in the
    middle of it (what a timeout report does: vln_persistent_check -> g_split_attn_enabled = 0), the remaining forward steps and the
    whole backward still run on the kernels the rollout started with -- they have no per-step query to fall back to -- instead of
    failing with VLN_ERR_ARG; the NEXT rollout takes the per-step query form.  Loss and gradients agree with the undisturbed run."""
    dev_ = torch.device(DEV)
    lib = vln._lib.load()
    tape = vln.synthetic.tape_to(vln.synthetic.make_tape(16, 24, 4, 6, seed=26), dev_, store_dtype=torch.bfloat16)
    res = []
    for disturb in (False, True):
        torch.manual_seed(29)
        ag = vln.trainers.EnvDropILIteration(dev_, torch.bfloat16, 1, arena=True)
        ag.enc._calls = 0; ag.dec._step_counter = 0
        ag.enc.deterministic_embedding_grad = True
        tape["store"]._calls = 0
        ag.opt.lr = 0.0
        assert ag.dec.project_context
        fwd, calls = ag.dec.forward, [0]

        def spy(*a, **k):
            calls[0] += 1
            if disturb and calls[0] == 2:              # after the rollout's first step went out in K mode
                lib.vln_set_split_attention(0)
            return fwd(*a, **k)
        ag.dec.forward = spy
        try:
            loss = ag.iteration(tape)
            torch.cuda.synchronize()
            assert ag.dec.last_projected
        finally:
            lib.vln_set_split_attention(1)
        res.append((loss.detach().clone(), [p.grad.detach().clone() for p in list(ag.dec.parameters()) + list(ag.enc.parameters())]))
    # (the panorama attention DOES have a one-workgroup form and takes it from the switch on: same numbers in another summation order)
    check(res[1][0], res[0][0], 1e-4, "loss with the split attention switched off mid-rollout")
    for i, (a, b) in enumerate(zip(res[0][1], res[1][1])):
        check(b, a, 2e-3, f"gradient {i} with the split attention switched off mid-rollout")


def test_missing_library_fails_loudly(vln, monkeypatch):
    monkeypatch.setattr(vln._lib, "_lib", None)
    monkeypatch.setattr(vln._lib, "LIB_PATH", "/nonexistent/libvln_hip.so")
    with pytest.raises(vln.VlnError):
        vln._lib.load()


def test_side_stream_gather_and_rollout_ce_are_transparent(vln):
    """trainers.EnvDropILIteration's two scheduling choices -- the per-step feature gather on a side stream and the IL loss of the whole
    rollout in one launch (losses.RolloutCE) -- change WHEN work is issued, not what is computed: gradients of every
    parameter equal the in-line / per-step configuration bit for bit over four arena iterations (dropout on); the loss
    value differs only by the summation order of the per-step terms."""
    dev_ = torch.device(DEV)
    tape = vln.synthetic.tape_to(vln.synthetic.make_tape(16, 24, 4, 6, seed=8), dev_, store_dtype=torch.bfloat16)
    res = []
    for side, rce in ((True, True), (False, False), (True, False), (False, True)):
        torch.manual_seed(19)
        ag = vln.trainers.EnvDropILIteration(dev_, torch.bfloat16, 1, arena=True, rollout_ce=rce, side_gather=side, fused_gather=False)
        ag.dec.batch_logit_backward = False           # (the batched logit branch sums in another order: its own test below)
        ag.dec.defer_logits = False
        ag.enc._calls = 0; ag.dec._step_counter = 0
        ag.enc.deterministic_embedding_grad = True
        tape["store"]._calls = 0
        ag.opt.lr = 0.0
        out = []
        for _ in range(4):
            loss = ag.iteration(tape)
            torch.cuda.synchronize()
            out.append((loss.detach().clone(), [p.grad.detach().clone() for p in list(ag.dec.parameters()) + list(ag.enc.parameters())]))
        res.append(out)
    for other in res[1:]:
        for (la, ga), (lb, gb) in zip(res[0], other):
            assert abs(la.item() - lb.item()) <= 1e-6 * abs(lb.item())
            for a, b in zip(ga, gb):
                assert torch.equal(a, b)


def test_host_feature_staging_matches_resident_tensors(vln):
    """bench.py --features host / host-bf16: per-step features in pinned host memory, copied on a copy stream into per-step
    device buffers; dropout off -> the same loss and gradients as the device-resident tensors (fp32 host: bit for bit; bf16
    host: the features are rounded once at load time, so to bf16 tolerance), over three arena iterations (buffer reuse)."""
    dev_ = torch.device(DEV)
    cpu_tape = vln.synthetic.make_tape(16, 24, 3, 6, seed=9)
    res = {}
    for mode, hd in (("tensor", None), ("host", torch.float32), ("host-bf16", torch.bfloat16)):
        tape = vln.synthetic.tape_to(cpu_tape, dev_, host_dtype=hd)
        torch.manual_seed(23)
        ag = vln.trainers.EnvDropILIteration(dev_, torch.bfloat16, 1, arena=True)
        ag.enc.eval(); ag.dec.eval(); ag.opt.lr = 0.0
        ag.enc.deterministic_embedding_grad = True
        for _ in range(3):
            loss = ag.iteration(tape)
            torch.cuda.synchronize()
        res[mode] = (loss.detach().clone(), [p.grad.detach().clone() for p in ag.dec.parameters()])
    assert torch.equal(res["tensor"][0], res["host"][0])
    for a, b in zip(res["tensor"][1], res["host"][1]):
        assert torch.equal(a, b)
    check(res["host-bf16"][0], res["tensor"][0], 1e-2, "bf16 host features: loss")


def test_host_feature_copies_under_the_previous_backward(vln):
    """Host-resident features with the arena: an iteration's H2D copies wait for the end of the iteration TWO back (the last
    reader of its buffer generation) and so run under the previous iteration's backward.  Three different episode batches
    rotating, six iterations back to back with no host sync in between, dropout ON: every iteration's loss and the final
    gradients equal the forward-only overlap (copies wait for the previous iteration's end) bit for bit."""
    dev_ = torch.device(DEV)
    cpu_tapes = [vln.synthetic.make_tape(16, 24, 3, 6, seed=90 + k) for k in range(3)]
    res = []
    for prefetch in (True, False):
        tapes = [vln.synthetic.tape_to(t, dev_, host_dtype=torch.float32) for t in cpu_tapes]
        torch.manual_seed(29)
        ag = vln.trainers.EnvDropILIteration(dev_, torch.bfloat16, 1, arena=True)
        ag.prefetch_under_backward = prefetch
        ag.enc._calls = 0; ag.dec._step_counter = 0
        ag.enc.deterministic_embedding_grad = True
        ag.opt.lr = 0.0
        losses = [ag.iteration(tapes[k % 3]).detach().clone() for k in range(6)]
        torch.cuda.synchronize()
        res.append((losses, [p.grad.detach().clone() for p in list(ag.dec.parameters()) + list(ag.enc.parameters())]))
    for a, b in zip(res[0][0], res[1][0]):
        assert torch.isfinite(a).all() and torch.equal(a, b)
    for a, b in zip(res[0][1], res[1][1]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.usefixtures("split_wgrads")
def test_batched_logit_branch_backward_equals_per_step(vln, dtype):
    """EnvDropDecoder.logit_branch_backward (losses.RolloutCE hands over every step's d logits at once: one multi-step
    weighted sum into the stash + one GEMM over (steps x batch) rows) against the per-step branch inside each step's
    backward, and `defer_logits` (the forward's candidate logits for the whole rollout at once) against the per-step logits:
    loss and every gradient to summation-order rounding; over four arena iterations with dropout
    on (plans and graph replays included), steps with different candidate counts."""
    dev_ = torch.device(DEV)
    tape = vln.synthetic.tape_to(vln.synthetic.make_tape(16, 24, 4, 6, seed=10), dev_, store_dtype=dtype)
    res = []
    for batched in (True, False):
        torch.manual_seed(29)
        ag = vln.trainers.EnvDropILIteration(dev_, dtype, 1, arena=True)
        ag.dec.batch_logit_backward = batched
        ag.dec.defer_logits = batched                 # ... and the forward's logits for the whole rollout at once
        ag.enc._calls = 0; ag.dec._step_counter = 0
        ag.enc.deterministic_embedding_grad = True
        tape["store"]._calls = 0
        ag.opt.lr = 0.0
        out = []
        for _ in range(4):
            loss = ag.iteration(tape)
            torch.cuda.synchronize()
            out.append((loss.detach().clone(), {n: p.grad.detach().clone() for n, p in list(ag.dec.named_parameters()) + list(ag.enc.named_parameters())}))
        res.append(out)
        if batched:
            assert ag.dec.plan_hits > 0
    for (la, ga), (lb, gb) in zip(res[0], res[1]):
        assert abs(la.item() - lb.item()) <= 1e-5 * abs(lb.item())
        scale = max(v.abs().max().item() for v in gb.values())
        for n in ga:
            err = (ga[n].double() - gb[n].double()).abs().max().item()
            assert err <= 2e-5 * max(gb[n].abs().max().item(), 1e-3 * scale), (n, err)


def test_deferred_logits_are_filled_by_the_rollout_loss(vln):
    """EnvDropDecoder.defer_logits: the tensors forward() returned hold the per-step logits once losses.RolloutCE has
    evaluated -- equal to the eagerly computed ones to rounding, the STOP / padded slots exactly 0."""
    B, L, V, Cn, H, F = 12, 9, 36, 5, 64, 256 + 128
    g = torch.Generator().manual_seed(41)
    ctx = torch.randn(B, L, H, generator=g).to(DEV)
    h = torch.randn(B, H, generator=g).to(DEV); c = torch.randn(B, H, generator=g).to(DEV)
    a = torch.randn(B, 128, generator=g).to(DEV)
    steps = []
    for t in range(3):
        img = torch.randn(B, V, F, generator=g).abs()
        cand = torch.randn(B, Cn - t, F, generator=g).abs()
        cand[:, -1] = 0
        steps.append((img.to(DEV), cand.to(DEV), torch.randint(0, Cn - t, (B,), generator=g).to(DEV)))
    outs = []
    for defer in (True, False):
        torch.manual_seed(3)
        dec = vln.EnvDropDecoder(H, 0.5, 0.3, 16, 128, F).to(DEV).eval()
        dec.defer_logits = defer
        hh, cc, ht = h.clone().requires_grad_(True), c.clone(), h.clone()
        ce = vln.losses.RolloutCE()
        logits = []
        for img, cand, tgt in steps:
            lg, (hh, cc), ht = dec(a, img.clone(), cand.clone(), ht, hh, cc, ctx)
            ce.add(lg, tgt)
            logits.append(lg)
        loss = ce.sum()
        loss.backward()
        outs.append((loss.detach(), [l.detach().clone() for l in logits]))
    check(outs[0][0], outs[1][0], 1e-5, "loss")
    for x, y in zip(outs[0][1], outs[1][1]):
        check(x, y, 1e-5, "deferred logits")
        assert x[:, -1].abs().max().item() == 0.0
    # a per-step consumer of deferred logits is told so instead of reading uninitialised memory
    dec = vln.EnvDropDecoder(H, 0.5, 0.3, 16, 128, F).to(DEV).eval()
    dec.defer_logits = True
    img, cand, tgt = steps[0]
    lg, _, _ = dec(a, img.clone(), cand.clone(), h.clone().requires_grad_(True), h, c, ctx)
    for fn in (lambda: vln.losses.masked_cross_entropy(lg, tgt, None, "sum"), lambda: vln.losses.sample_action(lg),
               lambda: vln.losses.action_stats(lg, tgt)):
        with pytest.raises(vln.VlnError):
            fn()


@pytest.mark.usefixtures("split_wgrads")
def test_per_sample_rollout_loss_through_the_decoder(vln):
    """SELF-PACE form (curriculum.py:296): `dot(weight, RolloutCE.per_sample())` through EnvDropDecoder steps -- the per-episode
    upstream gradients take the materialised-d-logits route of the rollout-wide logit branch; deferred logits + batched branch
    against the per-step configuration: loss vector and every gradient to rounding."""
    B, L, V, Cn, H, F = 10, 9, 36, 6, 64, 256 + 128
    g = torch.Generator().manual_seed(43)
    ctx0 = torch.randn(B, L, H, generator=g).to(DEV)
    h = torch.randn(B, H, generator=g).to(DEV); c = torch.randn(B, H, generator=g).to(DEV)
    a = torch.randn(B, 128, generator=g).to(DEV)
    w = torch.rand(B, generator=g).to(DEV)
    steps = []
    for t in range(3):
        n = torch.randint(2, Cn + 1, (B,), generator=g)
        cand = torch.randn(B, Cn, F, generator=g).abs() * (torch.arange(Cn)[None, :] < n[:, None])[..., None]
        tgt = (torch.rand(B, generator=g) * n.float()).long()
        tgt[torch.rand(B, generator=g) < 0.2] = -1
        steps.append((torch.randn(B, V, F, generator=g).abs().to(DEV), cand.to(DEV), tgt.to(DEV), (torch.arange(Cn)[None, :] >= n[:, None]).to(DEV)))
    res = []
    for fast in (True, False):
        torch.manual_seed(4)
        dec = vln.EnvDropDecoder(H, 0.5, 0.3, 16, 128, F).to(DEV).eval()
        dec.defer_logits = dec.batch_logit_backward = fast
        ctx = ctx0.clone().requires_grad_(True)
        hh, cc, ht = h.clone().requires_grad_(True), c.clone(), h.clone()
        ce = vln.losses.RolloutCE()
        for img, cand, tgt, cm in steps:
            lg, (hh, cc), ht = dec(a, img.clone(), cand.clone(), ht, hh, cc, ctx)
            ce.add(lg, tgt, cm)
        vec = ce.per_sample(scale=0.5)
        torch.dot(w, vec).backward()
        res.append((vec.detach().clone(), {n: p.grad.detach().clone() for n, p in dec.named_parameters()}, ctx.grad.clone()))
    check(res[0][0], res[1][0], 1e-5, "per-episode losses")
    check(res[0][2], res[1][2], 1e-4, "d ctx")
    scale = max(v.abs().max().item() for v in res[1][1].values())
    for n in res[0][1]:
        err = (res[0][1][n].double() - res[1][1][n].double()).abs().max().item()
        assert err <= 2e-5 * max(res[1][1][n].abs().max().item(), 1e-3 * scale), (n, err)


@pytest.mark.parametrize("per_sample", [False, True])
@pytest.mark.usefixtures("split_wgrads")
def test_rollout_ce_and_sampled_log_probs_share_the_logits(vln, per_sample):
    """Two differentiable consumers of the SAME logits (the ML loss and the sampled actions' log-probs / entropies,
    envdrop.py:173-195): with the rollout-wide logit branch (batch_logit_backward, the default) the second consumer's gradient
    used to be dropped silently (round-1 advisor finding).  Every gradient must equal the per-step configuration."""
    B, L, V, Cn, H, F = 12, 9, 36, 5, 64, 256 + 128
    g = torch.Generator().manual_seed(51)
    ctx0 = torch.randn(B, L, H, generator=g).to(DEV)
    h = torch.randn(B, H, generator=g).to(DEV); c = torch.randn(B, H, generator=g).to(DEV)
    a = torch.randn(B, 128, generator=g).to(DEV)
    steps = []
    for t in range(3):
        cand = torch.randn(B, Cn, F, generator=g).abs(); cand[:, -1] = 0
        steps.append((torch.randn(B, V, F, generator=g).abs().to(DEV), cand.to(DEV), torch.randint(0, Cn, (B,), generator=g).to(DEV),
                      torch.randint(0, Cn, (B,), generator=g).to(DEV)))
    w = torch.rand(B, generator=g).to(DEV)
    res = []
    for batched in (True, False):
        torch.manual_seed(6)
        dec = vln.EnvDropDecoder(H, 0.5, 0.3, 16, 128, F).to(DEV).eval()
        dec.batch_logit_backward = batched
        ctx = ctx0.clone().requires_grad_(True)
        hh, cc, ht = h.clone().requires_grad_(True), c.clone(), h.clone()
        ce = vln.losses.RolloutCE()
        rl = 0.0
        for img, cand, tgt, act in steps:
            lg, (hh, cc), ht = dec(a, img.clone(), cand.clone(), ht, hh, cc, ctx)
            ce.add(lg, tgt)
            _, logp, ent = vln.losses.sample_action(lg, None, action=act)
            rl = rl + (-(logp * 0.7).sum() - 0.01 * ent.sum())
        ml = torch.dot(w, ce.per_sample(scale=0.2)) if per_sample else ce.sum(scale=0.2)
        (ml + rl).backward()
        res.append(({n: p.grad.detach().clone() for n, p in dec.named_parameters()}, ctx.grad.clone()))
    check(res[0][1], res[1][1], 1e-5, "d ctx")
    scale = max(v.abs().max().item() for v in res[1][0].values())
    for n in res[0][0]:
        check(res[0][0][n], res[1][0][n], 2e-5, f"grad[{n}]", floor=1e-3 * scale)


@pytest.mark.parametrize("given_actions", [False, True])
@pytest.mark.usefixtures("split_wgrads")
def test_rollout_sampler_equals_per_step_sample_action(vln, given_actions):
    """losses.RolloutSampler: the sampled branch of every step (envdrop.py:186-195) with ONE backward node for the whole
    rollout (all d logits in one launch, then the decoder's rollout-wide logit branch) against `sample_action` per step with
    the same Philox offsets: identical draws, log-probs and entropies bit for bit; A2C loss (losses.a2c_loss) and every
    gradient to rounding (the batched branch contracts over steps x batch rows)."""
    B, L, V, Cn, H, F, T = 12, 9, 36, 6, 64, 256 + 128, 5
    g = torch.Generator().manual_seed(61)
    ctx0 = torch.randn(B, L, H, generator=g).to(DEV)
    h = torch.randn(B, H, generator=g).to(DEV); c = torch.randn(B, H, generator=g).to(DEV)
    a = torch.randn(B, 128, generator=g).to(DEV)
    steps = []
    for t in range(T):
        n = torch.randint(2, Cn + 1, (B,), generator=g)
        cmask = torch.arange(Cn)[None, :] >= n[:, None]
        cand = torch.randn(B, Cn, F, generator=g).abs() * (~cmask)[..., None]
        act = (torch.rand(B, generator=g) * n.float()).long()
        steps.append((torch.randn(B, V, F, generator=g).abs().to(DEV), cand.to(DEV), cmask.to(DEV), act.to(DEV)))
    rewards = [torch.randn(B, generator=g).sign().to(DEV) for _ in range(T)]
    lens = torch.randint(2, T + 1, (B,), generator=g); lens[0] = T
    masks = [(t < lens).to(DEV) for t in range(T)]
    ended = (lens < T).to(DEV)
    res = []
    for rollout_wide in (True, False):
        torch.manual_seed(6)
        dec = vln.EnvDropDecoder(H, 0.5, 0.3, 16, 128, F).to(DEV).eval()
        cri = vln.Critic(H, 0.5).to(DEV).eval()
        ctx = ctx0.clone().requires_grad_(True)
        hh, cc, ht = h.clone().requires_grad_(True), c.clone(), h.clone()
        sampler = vln.losses.RolloutSampler(seed=77)
        logps, ents, acts, hidden = [], [], [], []
        for t, (img, cand, cmask, act) in enumerate(steps):
            lg, (hh, cc), ht = dec(a, img.clone(), cand.clone(), ht, hh, cc, ctx)
            hidden.append(hh)
            given = act if given_actions else None
            if rollout_wide:
                acts.append(sampler.step(lg, cmask, action=given, offset=100 + t))
            else:
                a_t, lp_t, en_t = vln.losses.sample_action(lg, cmask, action=given, seed=77, offset=100 + t)
                acts.append(a_t); logps.append(lp_t); ents.append(en_t)
        if rollout_wide:
            logps, ents = sampler.stats()
            assert logps.shape == (T, B) and ents.shape == (T, B)
        vals = [cri(x) for x in hidden]
        with torch.no_grad():
            last_v = cri(hh).detach()
        loss, total = vln.losses.a2c_loss(logps, ents, vals, rewards, masks, last_v, ended, 0.9, "total")
        loss.backward()
        lp_all = logps.detach().clone() if rollout_wide else torch.stack([x.detach() for x in logps])
        en_all = ents.detach().clone() if rollout_wide else torch.stack([x.detach() for x in ents])
        res.append((torch.stack(acts).clone(), lp_all, en_all, loss.detach().clone(),
                    {n: p.grad.detach().clone() for n, p in dec.named_parameters()}, ctx.grad.clone()))
    assert torch.equal(res[0][0], res[1][0])                                # the same draws
    if not given_actions:
        for t, (_, _, cmask, _) in enumerate(steps):
            assert not cmask[torch.arange(B), res[0][0][t]].any()             # never a masked slot
    assert torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])
    check(res[0][3], res[1][3], 1e-6, "a2c loss")
    check(res[0][5], res[1][5], 1e-5, "d ctx")
    scale = max(v.abs().max().item() for v in res[1][4].values())
    for n in res[0][4]:
        check(res[0][4][n], res[1][4][n], 2e-5, f"grad[{n}]", floor=1e-3 * scale)


@pytest.mark.parametrize("ce_first", [True, False])
@pytest.mark.usefixtures("split_wgrads")
def test_rollout_ce_and_rollout_sampler_share_the_logits(vln, ce_first):
    """BOTH rollout-wide consumers on the same logits (envdrop.py:173-195 with train_ml and train_rl in one rollout): the
    decoder's rollout-wide logit branch can run only once per step, so whichever backward comes second must send its d logits
    through autograd to the steps (added there).  Either order == the per-step configuration, every gradient."""
    B, L, V, Cn, H, F, T = 12, 9, 36, 5, 64, 256 + 128, 3
    g = torch.Generator().manual_seed(71)
    ctx0 = torch.randn(B, L, H, generator=g).to(DEV)
    h = torch.randn(B, H, generator=g).to(DEV); c = torch.randn(B, H, generator=g).to(DEV)
    a = torch.randn(B, 128, generator=g).to(DEV)
    steps = []
    for t in range(T):
        cand = torch.randn(B, Cn, F, generator=g).abs(); cand[:, -1] = 0
        steps.append((torch.randn(B, V, F, generator=g).abs().to(DEV), cand.to(DEV), torch.randint(0, Cn, (B,), generator=g).to(DEV),
                      torch.randint(0, Cn, (B,), generator=g).to(DEV)))
    wl = torch.rand(T, B, generator=g).to(DEV)
    res = []
    for rollout_wide in (True, False):
        torch.manual_seed(6)
        dec = vln.EnvDropDecoder(H, 0.5, 0.3, 16, 128, F).to(DEV).eval()
        dec.batch_logit_backward = rollout_wide
        ctx = ctx0.clone().requires_grad_(True)
        hh, cc, ht = h.clone().requires_grad_(True), c.clone(), h.clone()
        ce = vln.losses.RolloutCE()
        sampler = vln.losses.RolloutSampler(seed=9)
        lps, ens = [], []
        for t, (img, cand, tgt, act) in enumerate(steps):
            lg, (hh, cc), ht = dec(a, img.clone(), cand.clone(), ht, hh, cc, ctx)
            ce.add(lg, tgt)
            if rollout_wide:
                sampler.step(lg, None, action=act, offset=50 + t)
            else:
                _, lp_t, en_t = vln.losses.sample_action(lg, None, action=act, seed=9, offset=50 + t)
                lps.append(lp_t); ens.append(en_t)
        if rollout_wide:
            lp, en = sampler.stats()
        else:
            lp, en = torch.stack(lps), torch.stack(ens)
        ml = ce.sum(scale=0.2)
        rl = -(lp * wl).sum() - 0.01 * en.sum()
        # autograd runs the node created LAST first: the order of the two sums decides which consumer takes the branch
        loss = (rl + ml) if ce_first else (ml + rl)
        loss.backward()
        res.append(({n: p.grad.detach().clone() for n, p in dec.named_parameters()}, ctx.grad.clone()))
    check(res[0][1], res[1][1], 1e-5, "d ctx")
    scale = max(v.abs().max().item() for v in res[1][0].values())
    for n in res[0][0]:
        check(res[0][0][n], res[1][0][n], 2e-5, f"grad[{n}]", floor=1e-3 * scale)


def test_long_rollouts_replay_their_step_graphs(vln):
    """T = 20 decoder steps per iteration (the reference's sampled rollouts run up to MAX_EPISODE_LEN = 35): the graph cache
    used to switch itself off for good after 24 misses in a row -- i.e. inside the first two (all-miss by construction)
    arena generations -- before the first possible hit.  Iterations 3.. must replay every step, forward and backward."""
    import ctypes
    dev_ = torch.device(DEV)
    T = 20
    tape = vln.synthetic.tape_to(vln.synthetic.make_tape(8, 12, T, 5, seed=12), dev_, store_dtype=torch.bfloat16)
    lib = vln._lib.load()
    ag = vln.trainers.EnvDropILIteration(dev_, torch.bfloat16, 1, arena=True)
    st = [(ctypes.c_int64 * 3)() for _ in range(3)]
    lib.vln_graph_stats(st[0])
    for _ in range(2):
        ag.iteration(tape)
    torch.cuda.synchronize()
    lib.vln_graph_stats(st[1])
    for _ in range(3):
        ag.iteration(tape)
    torch.cuda.synchronize()
    lib.vln_graph_stats(st[2])
    assert st[2][0] - st[1][0] >= 3 * 2 * T, (list(st[1]), list(st[2]))      # replays: 3 iterations x (fwd + bwd) x T steps
    assert st[2][1] == st[1][1]                                                # no further captures
    assert st[2][2] == st[0][2]                                                # capturing was never paused


def test_arena_refuses_tensors_whose_memory_was_recycled(vln):
    """ops.RolloutArena lifetime (a tensor of iteration i shares memory with iteration i + 2) is enforced, not only documented:
    a module that is handed a stamped tensor from 2+ iterations ago raises; within the window it works."""
    B, L, V, Cn, H, F = 4, 6, 36, 3, 64, 256 + 128
    g = torch.Generator().manual_seed(61)
    dec = vln.EnvDropDecoder(H, 0.5, 0.3, 16, 128, F).to(DEV).eval()
    cri = vln.Critic(H, 0.5).to(DEV).eval()
    ctx = torch.randn(B, L, H, generator=g).to(DEV); h = torch.randn(B, H, generator=g).to(DEV); c = torch.randn(B, H, generator=g).to(DEV)
    a = torch.randn(B, 128, generator=g).to(DEV)
    img = torch.randn(B, V, F, generator=g).abs().to(DEV); cand = torch.randn(B, Cn, F, generator=g).abs().to(DEV)
    arena = vln.ops.RolloutArena()
    vln.ops.set_arena(arena)
    try:
        kept = []
        for it in range(3):
            arena.begin()
            with torch.no_grad():
                lg, (h1, c1), ht = dec(a, img.clone(), cand.clone(), h, h, c, ctx)
            kept.append(h1)
            cri(kept[-1])                                  # same iteration: fine
            if it >= 1:
                cri(kept[-2])                              # previous iteration: still its own memory
        with pytest.raises(vln.VlnError):
            cri(kept[0])                                   # two iterations old: recycled
        with pytest.raises(vln.VlnError):
            dec(a, img.clone(), cand.clone(), kept[0], h, c, ctx)
    finally:
        vln.ops.set_arena(None)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.usefixtures("split_wgrads")
def test_step_gathers_its_own_features_like_caller_given_tensors(vln, dtype):
    """EnvDropDecoder.forward(gather=(store, indices)): the step reads its rows from the resident table inside its first launch
    and applies the feature dropout with its own Philox sites -- the very masks it uses on caller-given tensors.  Against the
    tensor path (explicit img / cand tensors, dropped in place) on the same episode data: loss and every gradient bit for
    bit, dropout ON, over four arena iterations (plans + graph replays)."""
    dev_ = torch.device(DEV)
    cpu_tape = vln.synthetic.make_tape(16, 24, 4, 6, seed=31)
    cpu_tape["table"] = cpu_tape["table"].bfloat16().float()          # values both table dtypes hold exactly
    for s_ in cpu_tape["steps"]:
        s_.update(vln.synthetic.materialize_step(s_, cpu_tape["table"]))
    res = []
    for gathered in (True, False):
        tape = vln.synthetic.tape_to(cpu_tape, dev_, store_dtype=dtype) if gathered else vln.synthetic.tape_to(cpu_tape, dev_)
        torch.manual_seed(37)
        ag = vln.trainers.EnvDropILIteration(dev_, dtype, 1, arena=True, fused_gather=gathered)
        ag.enc._calls = 0; ag.dec._step_counter = 0
        ag.enc.deterministic_embedding_grad = True
        ag.opt.lr = 0.0
        out = []
        for _ in range(4):
            loss = ag.iteration(tape)
            torch.cuda.synchronize()
            out.append((loss.detach().clone(), [p.grad.detach().clone() for p in list(ag.dec.parameters()) + list(ag.enc.parameters())]))
        if gathered:
            assert ag.dec.plan_hits >= 2 * 4
        res.append(out)
    # bf16: bit for bit.  fp32: the candidates' angle features come from the device's sinf / cosf in the gather and from
    # libm in the materialised tensors (last-bit differences, test_hip_staging.py): to rounding.
    for (la, ga), (lb, gb) in zip(res[0], res[1]):
        if dtype == torch.bfloat16:
            assert torch.equal(la, lb)
        else:
            check(la, lb, 1e-6, "loss")
        gmax = max(float(b.abs().max()) for b in gb)
        for i, (a, b) in enumerate(zip(ga, gb)):
            if dtype == torch.bfloat16:
                assert torch.equal(a, b)
            else:
                check(a, b, 1e-5, f"grad {i}", floor=1e-2 * gmax)


@pytest.mark.usefixtures("split_wgrads")
def test_envdrop_full_size_bf16_split_weight_gradients(vln):
    """The fp32-grade form of the bf16 mode's weight gradients (ops.set_wgrad_precision("split")): every gradient meets the
    same-weights bound of the outputs (1e-4)."""
    assert same_bf16_grad_tol() == SAME_BF16
    _full_size_envdrop(vln, torch.bfloat16)


@pytest.mark.parametrize("dtype,given_actions,C", [(torch.float32, False, 7), (torch.float32, True, 7), (torch.bfloat16, False, 7),
                                                   (torch.bfloat16, True, 7)])
def test_in_step_sampler_and_chained_backward_equal_the_separate_launches(vln, dtype, given_actions, C):
    """Round 5, sampled rollouts (envdrop.py:173,186-206): `forward(sampler=...)` runs mask + softmax + draw + log-prob + entropy
    inside the step's logits launch and stores the action into host-mapped pinned words itself (vln_envdrop_step.s_*), and
    `chain_backward` lets step t's act-embedding / h_tilde_prev backward stage ride in step t - 1's first backward launch
    (chain == 2).  Same arithmetic in the same order as `RolloutSampler.step` + a D2H copy and the unchained backward: logits,
    actions (device AND host words), log-probs, entropies, the loss and every gradient are equal bit for bit.  (More than 64
    candidates per row -- R2R has at most ~15 -- are refused by `forward(sampler=...)` before any launch: ADVICE r5.)"""
    import ctypes as C_
    dev = torch.device(DEV)
    B, L, V, H, IMG, ANG, AE, T = 16, 20, 36, 64, 96, 32, 16, 5
    F = IMG + ANG

    def rollout(in_step):
        torch.manual_seed(11)
        dec = vln.EnvDropDecoder(H, 0.5, 0.3, AE, ANG, F, compute_dtype=dtype).to(dev).train()
        dec.chain_backward = in_step
        g = torch.Generator().manual_seed(12)
        ctx = (torch.randn(B, L, H, generator=g) * 0.5).to(dev).requires_grad_(True)
        ht = torch.tanh(torch.randn(B, H, generator=g)).to(dev).requires_grad_(True); c = (torch.randn(B, H, generator=g) * 0.5).to(dev)
        mask = (torch.arange(L)[None, :] >= torch.randint(4, L + 1, (B, 1), generator=g)).to(dev)
        a_host = torch.full((T, B), -1, dtype=torch.int64).pin_memory()
        d = C_.c_void_p()
        vln._lib.check(vln._lib.load().vln_host_device_pointer(a_host.data_ptr(), C_.byref(d)), "vln_host_device_pointer")
        sampler = vln.losses.RolloutSampler(seed=77)
        h, hidden, logits, acts = ht, [], [], []
        for t in range(T):
            img = (torch.randn(B, V, F, generator=g).abs() * 0.5).to(dev); cand = (torch.randn(B, C, F, generator=g).abs() * 0.5).to(dev)
            ncand = torch.randint(2, C + 1, (B,), generator=g)
            cmask = (torch.arange(C)[None, :] >= ncand[:, None]).to(dev)
            given = (torch.rand(B, generator=g) * ncand.float()).long().to(dev) if given_actions else None
            a_in = torch.sin(torch.randn(B, ANG, generator=g)).to(dev)
            if in_step:
                logit, (h, c), ht = dec(a_in, img, cand, ht, h, c, ctx, mask, sampler=(sampler, cmask, given, int(d.value) + 8 * B * t))
                a = sampler.keep[-1][1]
            else:
                logit, (h, c), ht = dec(a_in, img, cand, ht, h, c, ctx, mask)
                a = sampler.step(logit, cmask, action=given, offset=None)
                a_host[t].copy_(a, non_blocking=True)
            hidden.append(h); logits.append(logit.detach().clone()); acts.append(a.clone())
        logp, ent = sampler.stats()
        w = torch.randn(T, B, generator=g).to(dev)
        loss = (logp * w).sum() - 0.01 * ent.sum() + sum(x.sum() for x in hidden) * 0.01 + ht.sum() * 0.1
        loss.backward()
        torch.cuda.synchronize()
        return dict(logits=logits, acts=acts, host=a_host.clone(), logp=logp.detach().clone(), ent=ent.detach().clone(), loss=loss.detach().clone(),
                    grads=[ctx.grad.clone()] + [p.grad.clone() for p in dec.parameters()])

    wide = vln.EnvDropDecoder(H, 0.5, 0.3, AE, ANG, F, compute_dtype=dtype).to(dev).train()
    with pytest.raises(ValueError, match="at most 64 candidates"):
        wide(torch.zeros(B, ANG, device=dev), torch.zeros(B, V, F, device=dev), torch.zeros(B, 70, F, device=dev), torch.zeros(B, H, device=dev),
             torch.zeros(B, H, device=dev), torch.zeros(B, H, device=dev), torch.zeros(B, L, H, device=dev), None,
             sampler=(vln.losses.RolloutSampler(seed=1), torch.zeros(B, 70, dtype=torch.bool, device=dev), None, 0))
    # the draws take their Philox offsets from a global call counter: both rollouts start from the same value
    start = vln.losses._sample_calls[0]
    ref = rollout(False)
    vln.losses._sample_calls[0] = start
    got = rollout(True)
    for t in range(T):
        assert torch.equal(ref["logits"][t], got["logits"][t]), f"step {t}: logits"
        assert torch.equal(ref["acts"][t], got["acts"][t]), f"step {t}: actions"
    assert torch.equal(ref["host"], got["host"]) and int(got["host"].min()) >= 0            # every word was written by the step's own launch
    for k in ("logp", "ent", "loss"):
        assert torch.equal(ref[k], got[k]), k
    for i, (a, b) in enumerate(zip(ref["grads"], got["grads"])):
        assert torch.equal(a, b), f"gradient {i} differs between the in-step sampler / chained backward and the separate launches"
