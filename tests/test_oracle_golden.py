"""CPU: pin the oracle (oracle/torch_port.py) against the golden vectors captured
from the imported reference (oracle/make_goldens.py).  Tolerance: float32
outputs 2e-5 abs+rel, gradients 1e-4 (different summation order only)."""
import pytest
import torch

from oracle import torch_port as O
from conftest import load_golden


def close(a, b, tol=2e-5, what=""):
    a, b = a.double(), b.double()
    err = (a - b).abs().max().item() if a.numel() else 0.0
    scale = max(1.0, b.abs().max().item()) if b.numel() else 1.0
    assert err <= tol * scale, f"{what}: max err {err:.3e} (scale {scale:.2f})"


def leafify(d):
    return {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in d.items()}


def check_grads(loss, P, gold_grads, extra=None, tol=1e-4):
    names = [k for k in gold_grads if k in P]
    xs = [P[k] for k in names]
    extra = extra or {}
    en = list(extra)
    gs = torch.autograd.grad(loss, xs + [extra[k] for k in en], allow_unused=True)
    for n, g in zip(names + en, gs):
        g = torch.zeros_like(gold_grads[n]) if g is None else g
        close(g, gold_grads[n], tol, f"grad[{n}]")


@pytest.mark.parametrize("name", ["encoder_envdrop", "encoder_follower", "encoder_monitor"])
def test_encoder(name):
    G = load_golden(name)
    cfg, I = G["cfg"], G["inp"]
    P = leafify(G["param"])
    ctx, h, c = O.encoder_forward(P, I["tokens"], I["lengths"].tolist(), num_layers=int(cfg["layers"]),
                                  bidirectional=bool(cfg["bidir"]))
    close(ctx, G["out"]["ctx"], what="ctx"); close(h, G["out"]["h"], what="h"); close(c, G["out"]["c"], what="c")
    # padded positions are exactly zero
    for i, n in enumerate(I["lengths"].tolist()):
        assert ctx[i, n:].abs().max().item() == 0.0 if n < ctx.shape[1] else True
    loss = (ctx * I["r1"]).sum() + (h * I["r2"]).sum() + (c * I["r3"]).sum()
    check_grads(loss, P, G["grad"])


@pytest.mark.parametrize("tag", ["full", "ctxonly", "visual"])
def test_softdot(tag):
    G = load_golden("softdot_" + tag)
    I, P = leafify(G["inp"]), leafify(G["param"])
    mask = None if tag == "visual" else I["mask"]
    out, attn = O.softdot_attention(I["h"], I["ctx"], mask, P["linear_in.weight"], P.get("linear_out.weight"))
    close(out, G["out"]["out"], what="out"); close(attn, G["out"]["attn"], what="attn")
    loss = (out * I["r"]).sum() + (attn * I["ra"]).sum()
    check_grads(loss, P, G["grad"], {"h": I["h"], "ctx": I["ctx"]})


@pytest.mark.parametrize("tag", ["follower", "monitor"])
def test_visualdot(tag):
    G = load_golden("visualdot_" + tag)
    I, P = leafify(G["inp"]), leafify(G["param"])
    mask = I["mask"] if tag == "monitor" else None
    out, attn = O.visual_softdot_attention(I["h"], I["v"], mask, P["linear_in_h.weight"], P["linear_in_h.bias"],
                                           P.get("linear_in_v.weight"), P.get("linear_in_v.bias"))
    close(out, G["out"]["out"], what="out"); close(attn, G["out"]["attn"], what="attn")
    loss = (out * I["r"]).sum() + (attn * I["ra"]).sum()
    check_grads(loss, P, G["grad"], {"h": I["h"], "v": I["v"]})


@pytest.mark.parametrize("name", ["envdrop_step", "envdrop_chain3"])
def test_envdrop(name):
    G = load_golden(name)
    I, P = leafify(G["inp"]), leafify(G["param"])
    T = int(G["cfg"]["steps"])
    h_tilde, c = I["h_tilde0"], I["c0"]
    loss = 0.
    for t in range(T):
        logit, (h1, c), h_tilde, _ = O.envdrop_step(P, I[f"a{t}"], I[f"img{t}"], I[f"cand{t}"], h_tilde, c,
                                                    I["ctx"], I["ctx_mask"])
        close(logit, G["out"][f"logit{t}"], what=f"logit{t}")
        close(h1, G["out"][f"h1_{t}"], what="h1"); close(c, G["out"][f"c1_{t}"], what="c1")
        close(h_tilde, G["out"][f"h_tilde{t}"], what="h_tilde")
        loss = loss + (logit * I[f"rl{t}"]).sum() + (h1 * I[f"rh{t}"]).sum() * 0.1
    loss = loss + (h_tilde * I["rf"]).sum() + (c * I["rc"]).sum()
    close(loss, G["out"]["loss"], 1e-5, "loss")
    check_grads(loss, P, G["grad"], {"ctx": I["ctx"], "h_tilde0": I["h_tilde0"], "c0": I["c0"]})


def test_envdrop_stop_logit_is_zero():
    """STOP slot is an all-zero feature row -> its logit is exactly 0."""
    G = load_golden("envdrop_step")
    I, P = G["inp"], G["param"]
    logit, *_ = O.envdrop_step(P, I["a0"], I["img0"], I["cand0"], I["h_tilde0"], I["c0"], I["ctx"], I["ctx_mask"])
    for i, n in enumerate((5, 4, 3, 2)):
        assert logit[i, n - 1].item() == 0.0


@pytest.mark.parametrize("name", ["follower_step", "follower_chain3"])
def test_follower(name):
    G = load_golden(name)
    I, P = leafify(G["inp"]), leafify(G["param"])
    T = int(G["cfg"]["steps"])
    h, c = I["h0"], I["c0"]
    loss = 0.
    for t in range(T):
        logit, (h, c), (ac, av) = O.follower_step(P, I[f"img{t}"], I[f"a_prev{t}"].detach(), I[f"cand{t}"], h, c,
                                                  I["ctx"], I["ctx_mask"])
        close(logit, G["out"][f"logit{t}"], what=f"logit{t}")
        close(h, G["out"][f"h1_{t}"], what="h1"); close(c, G["out"][f"c1_{t}"], what="c1")
        close(ac, G["out"][f"alpha_c{t}"], what="alpha_c"); close(av, G["out"][f"alpha_v{t}"], what="alpha_v")
        loss = loss + (logit * I[f"rl{t}"]).sum()
    loss = loss + (h * I["rf"]).sum() + (c * I["rc"]).sum()
    check_grads(loss, P, G["grad"], {"ctx": I["ctx"], "h0": I["h0"], "c0": I["c0"]})


@pytest.mark.parametrize("name", ["monitor_step_train", "monitor_step_eval"])
def test_monitor(name):
    G = load_golden(name)
    I, P = leafify(G["inp"]), leafify(G["param"])
    training = bool(G["cfg"]["training"])
    (logit, prog), (h1, c1), (ca, va), stats = O.monitor_step(
        P, I["a_prev"].detach(), I["cand"].detach(), I["h0"], I["c0"], I["ctx"], I["ctx_mask"], I["cand_mask"],
        training=training)
    for k, v in (("logit", logit), ("prog", prog), ("h1", h1), ("c1", c1), ("ctx_attn", ca), ("cand_attn", va)):
        close(v, G["out"][k], what=k)
    loss = (logit * I["rl"]).sum() + (prog * I["rp"]).sum() + (h1 * I["rh"]).sum() + (c1 * I["rc"]).sum()
    check_grads(loss, P, G["grad"], {"ctx": I["ctx"], "h0": I["h0"], "c0": I["c0"]}, tol=2e-4)
    if training:  # two running-stat updates per step (SURVEY §7 hard parts)
        A = G["param_after"]
        close(stats["rm0"], A["proj_navigable_mlp.mlp.0.running_mean"], what="rm0")
        close(stats["rv0"], A["proj_navigable_mlp.mlp.0.running_var"], what="rv0")
        close(stats["rm1"], A["proj_navigable_mlp.mlp.2.running_mean"], what="rm1")
        close(stats["rv1"], A["proj_navigable_mlp.mlp.2.running_var"], what="rv1")
        assert int(A["proj_navigable_mlp.mlp.0.num_batches_tracked"]) == 2


def test_critic():
    G = load_golden("critic")
    I, P = leafify(G["inp"]), leafify(G["param"])
    v = O.critic(P, I["state"])
    close(v, G["out"]["value"], what="value")
    check_grads((v * I["r"]).sum(), P, G["grad"], {"state": I["state"]})


def test_losses():
    G = load_golden("losses")
    I = leafify(G["inp"])
    B = I["logits"].shape[0]
    w = torch.arange(1, B + 1).float()
    for red in ("none", "sum", "mean"):
        ce = O.masked_cross_entropy(I["logits"], I["target"], I["cand_mask"], red)
        close(ce, G["out"][f"ce_{red}"], what=f"ce_{red}")
        (g,) = torch.autograd.grad((ce * (w if red == "none" else 1.0)).sum(), I["logits"])
        close(g, G["grad"][f"ce_{red}"], 1e-5, f"dce_{red}")
    lg = I["logits"].masked_fill(I["cand_mask"], -float("inf"))
    lp, ent = O.categorical_logprob_entropy(lg, I["action"])
    close(lp, G["out"]["log_prob"], what="log_prob"); close(ent, G["out"]["entropy"], what="entropy")
    (g,) = torch.autograd.grad((lp * w).sum() + (ent * 0.5).sum(), I["logits"])
    close(g, G["grad"]["cat"], 1e-5, "dcat")


def test_angle_features_and_masks():
    G = load_golden("angle_feats")
    table = O.loc_embedding_table(128)
    close(table, G["out"]["table"], 1e-6, "static location embeddings")
    for h, e, ref in zip(G["inp"]["headings"].tolist(), G["inp"]["elevations"].tolist(), G["out"]["samples"]):
        close(O.angle_feat(h, e), ref, 1e-6, "make_angle_feat")
    assert torch.equal(O.length2mask(G["inp"]["lengths"].tolist()), G["out"]["mask"])


def test_length2mask():
    m = O.length2mask([3, 1, 2])
    assert m.tolist() == [[False, False, False], [False, True, True], [False, False, True]]


@pytest.mark.parametrize("kind", ["bi", "uni"])
def test_speaker_encoder(kind):
    G = load_golden("speaker_encoder_" + kind)
    I = G["inp"]
    P = leafify(G["param"])
    ctx = O.speaker_encoder(P, I["act"], I["feat"], bool(int(G["cfg"]["bidir"])))
    close(ctx, G["out"]["ctx"], what="ctx")
    check_grads((ctx * I["r"]).sum(), P, G["grad"])


def test_speaker_decoder():
    G = load_golden("speaker_decoder")
    I = G["inp"]
    P = leafify(G["param"])
    ctx = I["ctx"].clone().requires_grad_(True)
    B, H = ctx.shape[0], ctx.shape[2]
    z = torch.zeros(1, B, H)
    logit, h1, c1 = O.speaker_decoder(P, I["words"], ctx, I["mask"], z, z)
    for a, k in ((logit, "logit"), (h1, "h1"), (c1, "c1")):
        close(a, G["out"][k], what=k)
    loss = (logit * I["r"]).sum() + (h1 * 0.3).sum() + (c1 * 0.2).sum()
    check_grads(loss, P, G["grad"], extra={"ctx": ctx})
    with torch.no_grad():                       # one word from a carried state (word-by-word inference)
        l2, h2, c2 = O.speaker_decoder(P, I["words"][:, :1], ctx, I["mask"], I["hs"], I["cs"])
    for a, k in ((l2, "step_logit"), (h2, "step_h"), (c2, "step_c")):
        close(a, G["out"][k], what=k)


def test_speaker_loop():
    """N3 loop (agent/speaker.py:235-376, envdrop.py:105-121): the CPU restatement of the speaker modules driven by the
    restated loop reproduces what the reference's own modules produced under the same loop (tests/golden/speaker_loop.npz)."""
    from oracle import rollout as R
    G = load_golden("speaker_loop")
    I, out, cfg = G["inp"], G["out"], G["cfg"]
    H, ANG, MAXD = int(cfg["H"]), int(cfg["ANG"]), int(cfg["MAXD"])
    spk = R.SpeakerOracle(G["enc"], G["dec"], True)
    lengths = I["lengths"].tolist()
    loss, per_word, predict = R.speaker_teacher_forcing(spk.encode, spk.decode, I["can"], I["img"], lengths, I["insts"], H)
    close(loss, out["loss"], what="loss")
    close(per_word, out["per_word"], what="per_word")
    assert torch.equal(predict, out["predict"])
    assert (per_word[I["insts"][:, 1:] == 0] == 0).all()              # pad targets carry no loss
    loss.backward()
    g = spk.named_grads()
    for n, ref in G["grad_enc"].items():
        close(g["encoder." + n], ref, 1e-4, "grad encoder." + n)
    for n, ref in G["grad_dec"].items():
        close(g.get("decoder." + n, torch.zeros_like(ref)), ref, 1e-4, "grad decoder." + n)
    with torch.no_grad():
        words, step_logits = R.speaker_infer_batch(spk.encode, spk.decode, I["can"], I["img"], lengths, H, MAXD,
                                                   featdropmask=I["noise"].double(), angle=ANG)
        plain, _ = R.speaker_infer_batch(spk.encode, spk.decode, I["can"], I["img"], lengths, H, MAXD, angle=ANG)
    assert (words == out["words"].numpy()).all() and (plain == out["words_plain"].numpy()).all()
    fin = torch.isfinite(out["step_logits"])
    close(step_logits[fin], out["step_logits"][fin], 1e-4, "step logits")
    assert (step_logits[:, :, 1] == -float("inf")).all()                # <UNK> is never produced
    bt = R.back_translate_instructions(words)
    assert (bt == out["instr_encoding"].numpy()).all()
    assert (bt[:, 0] == 3).all() and all(2 in row for row in bt)        # <BOS> first, every sentence closed by <EOS>
