"""Generate tests/golden/*.npz by IMPORTING the reference (build container only).

Test infrastructure.  Runs only where `/root/reference` exists; the reference
sources never travel -- only the captured input/output vectors do.

Recipe: `src/model/__init__.py` drags in boto3, so `units.py`
and `policy.py` are loaded as a synthetic package `refmodel` straight from
their files.  All captures are float32, dropout OFF (module.eval()), except
the Self-Monitor train-mode BatchNorm capture where the two Dropout layers are
set to p=0 so batch statistics are exercised deterministically.

    python oracle/make_goldens.py            # rewrites tests/golden/*.npz
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/tasks/R2R-judy/src/model"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def load_reference():
    pkg = types.ModuleType("refmodel")
    pkg.__path__ = [REF]
    sys.modules["refmodel"] = pkg
    mods = {}
    for name in ("units", "policy"):
        spec = importlib.util.spec_from_file_location(f"refmodel.{name}", os.path.join(REF, f"{name}.py"))
        m = importlib.util.module_from_spec(spec)
        sys.modules[f"refmodel.{name}"] = m
        spec.loader.exec_module(m)
        setattr(pkg, name, m)
        mods[name] = m
    return mods["units"], mods["policy"]


def npify(d):
    out = {}
    for k, v in d.items():
        if isinstance(v, torch.Tensor):
            out[k] = v.detach().cpu().numpy()
        else:
            out[k] = np.asarray(v)
    return out


def save(name, **groups):
    flat = {}
    for g, d in groups.items():
        for k, v in npify(d).items():
            flat[f"{g}/{k}"] = v
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **flat)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} kB, {len(flat)} arrays")


def grads_of(module, loss, extra=()):
    params = [p for p in module.parameters() if p.requires_grad]
    names = [n for n, p in module.named_parameters() if p.requires_grad]
    gs = torch.autograd.grad(loss, params + list(extra), allow_unused=True)
    out = {n: (g if g is not None else torch.zeros_like(p)) for n, g, p in zip(names, gs, params)}
    return out, gs[len(params):]


def feats(g, B, S, img, angle):
    """Non-negative image part (post-ReLU ResNet pool5) + angle tail in [-1,1]."""
    x = torch.randn(B, S, img + angle, generator=g)
    x[..., :img] = x[..., :img].abs() * 0.5
    x[..., img:] = torch.sin(x[..., img:] * 3)
    return x


def gen_encoder(U, name, g, *, vocab, E, H, bidir, layers, B=4, L=20, lengths=(20, 13, 7, 3)):
    torch.manual_seed(int(torch.randint(0, 10000, (1,), generator=g)))
    enc = U.EncoderLSTM(vocab, E, H, padding_idx=0, drop_ratio=0.5, bidirectional=bidir, num_layers=layers).eval()
    tokens = torch.zeros(B, L, dtype=torch.long)
    for i, n in enumerate(lengths):
        tokens[i, :n] = torch.randint(4, vocab, (n,), generator=g)
    lens = torch.tensor(lengths)
    ctx, h, c = enc(tokens, lens)
    r1, r2, r3 = torch.randn(ctx.shape, generator=g), torch.randn(h.shape, generator=g), torch.randn(c.shape, generator=g)
    loss = (ctx * r1).sum() + (h * r2).sum() + (c * r3).sum()
    gp, _ = grads_of(enc, loss)
    save(name, cfg=dict(vocab=vocab, E=E, H=H, bidir=int(bidir), layers=layers),
         inp=dict(tokens=tokens, lengths=lens, r1=r1, r2=r2, r3=r3),
         param=dict(enc.state_dict()), out=dict(ctx=ctx, h=h, c=c, loss=loss), grad=gp)


def gen_attention(U, g):
    B, S, Q, D = 4, 9, 32, 48
    # (1) text attention, full (cat + linear_out + tanh), masked
    for tag, ctx_only, cdim in (("full", False, None), ("ctxonly", True, None), ("visual", True, D)):
        torch.manual_seed(7)
        att = U.SoftDotAttention(Q, context_only=ctx_only, context_dim=cdim).eval()
        d = Q if cdim is None else cdim
        h = torch.randn(B, Q, generator=g, requires_grad=True)
        ctx = torch.randn(B, S, d, generator=g, requires_grad=True)
        mask = torch.zeros(B, S, dtype=torch.bool)
        if tag != "visual":
            for i, n in enumerate((9, 6, 4, 1)):
                mask[i, n:] = True
        out, attn = att(h, ctx, mask if tag != "visual" else None)
        r = torch.randn(out.shape, generator=g)
        ra = torch.randn(attn.shape, generator=g)
        loss = (out * r).sum() + (attn * ra).sum()
        gp, (gh, gc) = grads_of(att, loss, (h, ctx))
        save(f"softdot_{tag}", inp=dict(h=h, ctx=ctx, mask=mask, r=r, ra=ra), param=dict(att.state_dict()),
             out=dict(out=out, attn=attn, loss=loss), grad=dict(gp, h=gh, ctx=gc))
    # (2) VisualSoftDot: follower (v-projection) and monitor (no projection, masked)
    for tag, vdim, dot in (("follower", 40, 16), ("monitor", None, 24)):
        torch.manual_seed(11)
        att = U.VisualSoftDotAttention(Q, vdim, dot).eval()
        vd = vdim if vdim is not None else dot
        h = torch.randn(B, Q, generator=g, requires_grad=True)
        v = torch.randn(B, S, vd, generator=g, requires_grad=True)
        mask = None
        if tag == "monitor":
            mask = torch.zeros(B, S, dtype=torch.bool)
            for i, n in enumerate((9, 5, 3, 2)):
                mask[i, n:] = True
        out, attn = att(h, v, mask)
        r = torch.randn(out.shape, generator=g)
        ra = torch.randn(attn.shape, generator=g)
        loss = (out * r).sum() + (attn * ra).sum()
        gp, (gh, gv) = grads_of(att, loss, (h, v))
        save(f"visualdot_{tag}", inp=dict(h=h, v=v, mask=mask if mask is not None else torch.zeros(0), r=r, ra=ra),
             param=dict(att.state_dict()), out=dict(out=out, attn=attn, loss=loss), grad=dict(gp, h=gh, v=gv))


def gen_envdrop(P, g, steps, name):
    B, L, V, C, H, IMG, ANG, AE = 4, 11, 36, 5, 64, 96, 32, 16
    F = IMG + ANG
    torch.manual_seed(21)
    dec = P.EnvDropDecoder(H, 0.5, 0.3, action_embed_size=AE, angle_feat_size=ANG, feature_size=F).eval()
    ctx = torch.randn(B, L, H, generator=g, requires_grad=True)
    ctx_mask = torch.zeros(B, L, dtype=torch.bool)
    for i, n in enumerate((11, 8, 5, 2)):
        ctx_mask[i, n:] = True
    h_tilde = torch.randn(B, H, generator=g, requires_grad=True)
    c = torch.randn(B, H, generator=g, requires_grad=True)
    h_t = torch.randn(B, H, generator=g)
    ht0, c0 = h_tilde, c
    inp, out = dict(ctx=ctx, ctx_mask=ctx_mask, h_tilde0=h_tilde, c0=c), {}
    loss = 0.
    cand_len = (5, 4, 3, 2)
    for t in range(steps):
        a = torch.sin(torch.randn(B, ANG, generator=g) * 3)
        img = feats(g, B, V, IMG, ANG)
        cand = feats(g, B, C, IMG, ANG)
        for i, n in enumerate(cand_len):
            cand[i, n - 1:] = 0                                # STOP slot + padding are zero rows
        logit, (h_t, c), h_tilde = dec(a, img.clone(), cand.clone(), h_tilde, h_t, c, ctx, ctx_mask, False)
        rl = torch.randn(logit.shape, generator=g)
        rh = torch.randn(B, H, generator=g)
        loss = loss + (logit * rl).sum() + (h_t * rh).sum() * 0.1
        inp.update({f"a{t}": a, f"img{t}": img, f"cand{t}": cand, f"rl{t}": rl, f"rh{t}": rh})
        out.update({f"logit{t}": logit, f"h1_{t}": h_t, f"c1_{t}": c, f"h_tilde{t}": h_tilde})
    rf = torch.randn(B, H, generator=g)
    rc = torch.randn(B, H, generator=g)
    loss = loss + (h_tilde * rf).sum() + (c * rc).sum()
    inp.update(rf=rf, rc=rc)
    out["loss"] = loss
    gp, (gctx, ght, gc0) = grads_of(dec, loss, (ctx, ht0, c0))
    save(name, cfg=dict(steps=steps, H=H, IMG=IMG, ANG=ANG, AE=AE), inp=inp, param=dict(dec.state_dict()),
         out=out, grad=dict(gp, ctx=gctx, h_tilde0=ght, c0=gc0))


def gen_follower(P, g, steps, name):
    B, L, V, C, H, IMG, ANG = 4, 11, 36, 5, 32, 64, 16
    F = IMG + ANG
    torch.manual_seed(22)
    dec = P.AttnDecoderLSTM(H, 0.5, action_embed_size=F, feature_size=F).eval()
    ctx = torch.randn(B, L, H, generator=g, requires_grad=True)
    ctx_mask = torch.zeros(B, L, dtype=torch.bool)
    for i, n in enumerate((11, 8, 5, 2)):
        ctx_mask[i, n:] = True
    h = torch.randn(B, H, generator=g, requires_grad=True)
    c = torch.randn(B, H, generator=g, requires_grad=True)
    h0, c0 = h, c
    inp, out = dict(ctx=ctx, ctx_mask=ctx_mask, h0=h, c0=c), {}
    loss = 0.
    a_prev = torch.zeros(B, F)
    for t in range(steps):
        img = feats(g, B, V, IMG, ANG)
        cand = feats(g, B, C, IMG, ANG)
        for i, n in enumerate((5, 4, 3, 2)):
            cand[i, n - 1:] = 0
        logit, (h, c), (ac, av) = dec(img, a_prev, cand, h, c, ctx, ctx_mask)
        rl = torch.randn(logit.shape, generator=g)
        loss = loss + (logit * rl).sum()
        inp.update({f"img{t}": img, f"cand{t}": cand, f"a_prev{t}": a_prev, f"rl{t}": rl})
        out.update({f"logit{t}": logit, f"h1_{t}": h, f"c1_{t}": c, f"alpha_c{t}": ac, f"alpha_v{t}": av})
        a_prev = cand[:, t % 2].detach()
    rf = torch.randn(B, H, generator=g)
    rc = torch.randn(B, H, generator=g)
    loss = loss + (h * rf).sum() + (c * rc).sum()
    inp.update(rf=rf, rc=rc)
    out["loss"] = loss
    gp, (gctx, gh0, gc0) = grads_of(dec, loss, (ctx, h0, c0))
    save(name, cfg=dict(steps=steps, H=H, IMG=IMG, ANG=ANG), inp=inp, param=dict(dec.state_dict()),
         out=out, grad=dict(gp, ctx=gctx, h0=gh0, c0=gc0))


def gen_monitor(P, g, training, name):
    B, L, C, H, IMG, ANG, M = 4, 12, 5, 32, 64, 16, 48
    F = IMG + ANG
    torch.manual_seed(23)
    dec = P.MonitorDecoder(H, 0.5, L, mlp_dims=[M], action_embed_size=F, feature_size=F)
    # non-trivial BN affine + running stats so eval mode is exercised
    with torch.no_grad():
        for k in ("0", "2"):
            bn = dec.proj_navigable_mlp.mlp[int(k)]
            bn.weight.uniform_(0.5, 1.5, generator=g); bn.bias.uniform_(-0.3, 0.3, generator=g)
            bn.running_mean.uniform_(-0.2, 0.2, generator=g); bn.running_var.uniform_(0.5, 1.5, generator=g)
    if training:
        dec.train()
        for m in dec.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
    else:
        dec.eval()
    sd_before = {k: v.clone() for k, v in dec.state_dict().items()}
    ctx = torch.randn(B, L, H, generator=g, requires_grad=True)
    ctx_mask = torch.zeros(B, L, dtype=torch.bool)
    for i, n in enumerate((12, 8, 5, 2)):
        ctx_mask[i, n:] = True
    h = torch.randn(B, H, generator=g, requires_grad=True)
    c = torch.randn(B, H, generator=g, requires_grad=True)
    a_prev = feats(g, B, 1, IMG, ANG)[:, 0]
    cand = feats(g, B, C, IMG, ANG)
    cand_len = (5, 4, 3, 2)
    for i, n in enumerate(cand_len):
        cand[i, n - 1:] = 0
    cmask = torch.zeros(B, C, dtype=torch.bool)
    for i, n in enumerate(cand_len):
        cmask[i, n:] = True
    (logit, prog), (h1, c1), (ctx_attn, cand_attn) = dec(None, a_prev, cand, h, c, ctx, ctx_mask, cmask)
    rl, rp = torch.randn(logit.shape, generator=g), torch.randn(prog.shape, generator=g)
    rh, rc = torch.randn(B, H, generator=g), torch.randn(B, H, generator=g)
    loss = (logit * rl).sum() + (prog * rp).sum() + (h1 * rh).sum() + (c1 * rc).sum()
    gp, (gctx, gh0, gc0) = grads_of(dec, loss, (ctx, h, c))
    save(name, cfg=dict(H=H, IMG=IMG, ANG=ANG, M=M, L=L, training=int(training)),
         inp=dict(ctx=ctx, ctx_mask=ctx_mask, h0=h, c0=c, a_prev=a_prev, cand=cand, cand_mask=cmask,
                  rl=rl, rp=rp, rh=rh, rc=rc),
         param=sd_before, param_after=dict(dec.state_dict()),
         out=dict(logit=logit, prog=prog, h1=h1, c1=c1, ctx_attn=ctx_attn, cand_attn=cand_attn, loss=loss),
         grad=dict(gp, ctx=gctx, h0=gh0, c0=gc0))


def gen_critic(P, g):
    torch.manual_seed(24)
    cr = P.Critic(64, 0.5).eval()
    s = torch.randn(5, 64, generator=g, requires_grad=True)
    v = cr(s)
    r = torch.randn(5, generator=g)
    loss = (v * r).sum()
    gp, (gs,) = grads_of(cr, loss, (s,))
    save("critic", inp=dict(state=s, r=r), param=dict(cr.state_dict()), out=dict(value=v, loss=loss), grad=dict(gp, state=gs))


def gen_losses(g):
    """The torch ops the agents call inline for A9 (follower.py:62,123-128;
    envdrop.py:70,173-195): CrossEntropyLoss(ignore_index=-1) on -inf-masked
    logits, Categorical log_prob/entropy.  Captured from torch itself -- the
    arithmetic the reference delegates to."""
    B, C = 6, 5
    logits = torch.randn(B, C, generator=g, requires_grad=True)
    lens = (5, 4, 3, 2, 5, 1)
    cmask = torch.zeros(B, C, dtype=torch.bool)
    for i, n in enumerate(lens):
        cmask[i, n:] = True
    target = torch.tensor([1, 3, -1, 0, 4, 0])
    out, grad = {}, {}
    for red in ("none", "sum", "mean"):
        lg = logits.masked_fill(cmask, -float("inf"))
        ce = torch.nn.CrossEntropyLoss(ignore_index=-1, reduction=red)(lg, target)
        w = torch.arange(1, B + 1).float() if red == "none" else torch.tensor(1.0)
        (gl,) = torch.autograd.grad((ce * w).sum(), logits)
        out[f"ce_{red}"] = ce
        grad[f"ce_{red}"] = gl
    lg = logits.masked_fill(cmask, -float("inf"))
    dist = torch.distributions.Categorical(torch.softmax(lg, 1))
    act = torch.tensor([0, 2, 1, 1, 4, 0])
    lp, ent = dist.log_prob(act), dist.entropy()
    (gl,) = torch.autograd.grad((lp * torch.arange(1, B + 1).float()).sum() + (ent * 0.5).sum(), logits)
    out.update(log_prob=lp, entropy=ent)
    grad["cat"] = gl
    save("losses", inp=dict(logits=logits, cand_mask=cmask, target=target, action=act), out=out, grad=grad)


def gen_angle_tables():
    """utils/misc.py:285-317 (make_angle_feat, the 36 static panorama location embeddings) and misc.py:481-486
    (length2mask).  misc.py imports MatterSim / prettytable at module level: stubbed, never called."""
    for m in ("MatterSim", "prettytable"):
        sys.modules.setdefault(m, types.ModuleType(m))
    sys.modules["prettytable"].PrettyTable = object
    spec = importlib.util.spec_from_file_location("refmisc", "/root/reference/tasks/R2R-judy/src/utils/misc.py")
    misc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(misc)
    table = np.stack(misc._static_loc_embeddings)                       # [36 viewIndex, 36 views, 128]
    hs = np.linspace(-3.0, 3.0, 7).astype(np.float32)
    es = np.linspace(-0.5, 0.5, 7).astype(np.float32)
    samples = np.stack([misc.ImageFeatures.make_angle_feat(float(h), float(e)) for h, e in zip(hs, es)])
    lens = [3, 1, 5, 2]
    save("angle_feats", out=dict(table=table, samples=samples, mask=misc.length2mask(lens)),
         inp=dict(headings=hs, elevations=es, lengths=np.array(lens)))


def _import_reference_agents():
    """`import src.agent` needs six absent third-party modules: stub them in sys.modules (never
    called on this path) and put tasks/R2R-judy on sys.path.  Reference files are untouched."""
    class _CN(dict):                       # attribute-style dict: enough for utils/config.py to build its defaults
        __getattr__ = dict.__getitem__
        __setattr__ = dict.__setitem__

        def clone(self):
            return self

    stubs = {"MatterSim": {}, "prettytable": {"PrettyTable": object}, "yacs": {}, "yacs.config": {"CfgNode": _CN},
             "boto3": {}, "botocore": {}, "botocore.exceptions": {"ClientError": Exception},
             "tensorboardX": {"SummaryWriter": object}}
    for name, attrs in stubs.items():
        m = sys.modules.get(name) or types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
    root = "/root/reference/tasks/R2R-judy"
    if root not in sys.path:
        sys.path.insert(0, root)
    import src.agent as A
    return A


class _Tok:
    word_to_index = {"<PAD>": 0, "<UNK>": 1, "<EOS>": 2, "<BOS>": 3}

    def __init__(self, n):
        self.n = n

    def vocab_size(self):
        return self.n


def gen_agent_tapes():
    """Rollout-level goldens (SURVEY §8c last row): the reference's EnvDropAgent.rollout() driven by oracle/fake_env.py,
    teacher-forced (IL) and sampled (IL + A2C), dropout off.  `torch.Tensor.cpu` is patched to return a clone while
    capturing: on CPU the reference's `a_t.detach().cpu().numpy()` aliases the CE target and silently drops the
    STOP rows' gradient; on a GPU `.cpu()` copies, which is the intended semantics."""
    A = _import_reference_agents()
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from oracle.fake_env import FakeR2REnv
    orig_cpu = torch.Tensor.cpu
    torch.Tensor.cpu = lambda self, *a, **k: orig_cpu(self, *a, **k).clone()
    try:
        cfg = types.SimpleNamespace(ACT_EMB_SIZE=8, WORD_EMB_SIZE=16, HIDDEN_SIZE=32, DROP_RATE=0.5, FEAT_DROP_RATE=0.3,
                                    ENC_BIDIRECTION=True, ENC_LAYERS=1, ML_WEIGHT=0.2, GAMMA=0.9, RL_NORMALIZE="total")
        for mode in ("teacher", "sample"):
            torch.manual_seed(2020)
            env = FakeR2REnv(batch_size=4, max_len=8, vocab=40, seed=7)
            agent = A.EnvDropAgent(cfg, 8, "/tmp", torch.device("cpu"), env, _Tok(40), episode_len=6)
            agent.eval()                                               # dropout off; losses are still built
            sd = {"enc." + k: v.clone() for k, v in agent.encoder.state_dict().items()}
            sd.update({"dec." + k: v.clone() for k, v in agent.decoder.state_dict().items()})
            sd.update({"cri." + k: v.clone() for k, v in agent.critic.state_dict().items()})
            torch.manual_seed(99)
            agent.rollout(train_ml=True, train_rl=(mode == "sample"), train_cl=False, reset=True, feedback=mode)
            loss = agent.loss["ml_loss"] + agent.loss["rl_loss"]
            params = dict([("enc." + n, p) for n, p in agent.encoder.named_parameters()] +
                          [("dec." + n, p) for n, p in agent.decoder.named_parameters()] +
                          [("cri." + n, p) for n, p in agent.critic.named_parameters()])
            gs = torch.autograd.grad(loss, list(params.values()), allow_unused=True)
            grads = {n: (g if g is not None else torch.zeros_like(p)) for (n, p), g in zip(params.items(), gs)}
            small = {n: g for n, g in grads.items() if g.numel() <= 4096}
            norms = {n: g.norm() for n, g in grads.items()}
            out = dict(ml_loss=torch.as_tensor(float(agent.loss["ml_loss"])), rl_loss=torch.as_tensor(float(agent.loss["rl_loss"])),
                       actions=np.stack(env.actions_log))
            if mode == "sample":
                out["total"] = np.array(agent.logs["total"][-1])
            save(f"agent_envdrop_{mode}", cfg={k: (int(v) if isinstance(v, bool) else v) for k, v in vars(cfg).items()
                                                if not isinstance(v, str)},
                 param=sd, out=out, grad=small, gradnorm=norms)
    finally:
        torch.Tensor.cpu = orig_cpu


def gen_agent_tapes_more():
    """Rollout-level goldens for the other two agents and for the evaluation path (SURVEY §8c last row, §8f N4):
    FollowerAgent / SelfMonitorAgent teacher-forced (training loss + grads) and all three agents with
    feedback="argmax" in eval mode (what `BaseAgent.test` runs, base.py:63-82): actions, trajectories, loss."""
    A = _import_reference_agents()
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from oracle.fake_env import FakeR2REnv
    orig_cpu = torch.Tensor.cpu
    torch.Tensor.cpu = lambda self, *a, **k: orig_cpu(self, *a, **k).clone()
    # 64-wide visual features instead of 2048 keep the captured parameters small (the agents take the width from
    # BasicR2RAgent's constructor default, base.py:95-104; changed on the imported class object only)
    init = A.BasicR2RAgent.__init__
    orig_defaults = init.__defaults__
    init.__defaults__ = (64,) + tuple(orig_defaults[1:])
    try:
        base = dict(ACT_EMB_SIZE=8, WORD_EMB_SIZE=16, HIDDEN_SIZE=32, DROP_RATE=0.5, FEAT_DROP_RATE=0.3, ML_WEIGHT=0.2,
                    GAMMA=0.9, RL_NORMALIZE="total")
        kinds = {
            "follower": (dict(base, ENC_BIDIRECTION=True, ENC_LAYERS=2),
                         lambda cfg, env: A.FollowerAgent(cfg, "/tmp", torch.device("cpu"), env, _Tok(40), episode_len=6)),
            "monitor": (dict(base, ENC_BIDIRECTION=False, ENC_LAYERS=1, MLP_HIDDEN=[24]),
                        lambda cfg, env: A.SelfMonitorAgent(cfg, 8, "/tmp", torch.device("cpu"), env, _Tok(40), episode_len=6)),
            "envdrop": (dict(base, ENC_BIDIRECTION=True, ENC_LAYERS=1),
                        lambda cfg, env: A.EnvDropAgent(cfg, 8, "/tmp", torch.device("cpu"), env, _Tok(40), episode_len=6)),
        }
        for kind, (cfgd, make) in kinds.items():
            for mode in ("teacher", "argmax"):
                if kind == "envdrop" and mode == "teacher":
                    continue                                        # already captured by gen_agent_tapes
                cfg = types.SimpleNamespace(**cfgd)
                torch.manual_seed(2021)
                env = FakeR2REnv(batch_size=5, max_len=8, vocab=40, seed=11 if mode == "teacher" else 13, img=64)
                agent = make(cfg, env)
                assert agent.feature_size == 64 + 128
                if hasattr(agent, "reset_loss"):
                    agent.reset_loss()
                agent.eval()                                        # dropout off, BatchNorm on running stats
                mods = [("enc.", agent.encoder), ("dec.", agent.decoder)]
                sd = {pre + k: v.clone() for pre, m in mods for k, v in m.state_dict().items()}
                torch.manual_seed(99)
                if kind == "envdrop":
                    traj = agent.rollout(train_ml=True, train_rl=False, train_cl=False, reset=True, feedback=mode)
                    ml = agent.loss["ml_loss"]
                elif kind == "monitor":
                    traj = agent.rollout(train_ml=True, train_cl=False, reset=True, lamb=0.5, feedback=mode)
                    ml = agent.ml_loss
                else:
                    traj = agent.rollout(train_ml=True, train_rl=False, train_cl=False, reset=True, feedback=mode)
                    ml = agent.ml_loss
                params = dict((pre + n, p) for pre, m in mods for n, p in m.named_parameters())
                gs = torch.autograd.grad(ml, list(params.values()), allow_unused=True)
                grads = {n: (g if g is not None else torch.zeros_like(p)) for (n, p), g in zip(params.items(), gs)}
                small = {n: g for n, g in grads.items() if g.numel() <= 4096}
                norms = {n: g.norm() for n, g in grads.items()}
                out = dict(ml_loss=torch.as_tensor(float(ml.detach())), actions=np.stack(env.actions_log),
                           path_len=np.array([len(t["path"]) for t in traj]))
                if kind == "monitor":
                    out["progress_loss"] = torch.as_tensor(float(agent.progress_loss))
                save(f"agent_{kind}_{mode}", cfg={k: (int(v) if isinstance(v, bool) else v) for k, v in cfgd.items()
                                                   if not isinstance(v, (str, list))},
                     param=sd, out=out, grad=small, gradnorm=norms)
    finally:
        torch.Tensor.cpu = orig_cpu
        init.__defaults__ = orig_defaults


def gen_eval_scores():
    """Evaluation-path golden: the reference's `Evaluation.score` (engine/evaluator.py:101-146) on a
    synthetic weighted navigation graph with hand-made trajectories (success, overshoot, early stop, detour).  The
    object is built without its constructor (which reads the R2R json files); only the fields score() touches are set."""
    import json
    from collections import defaultdict
    import networkx as nx
    _import_reference_agents()
    from src.engine.evaluator import Evaluation
    rng = np.random.default_rng(5)
    G = nx.Graph()
    W, H = 5, 4
    edges = []
    for x in range(W):
        for y in range(H):
            for dx, dy in ((1, 0), (0, 1)):
                if x + dx < W and y + dy < H:
                    w = float(np.round(1.0 + 2.0 * rng.random(), 3))
                    edges.append((f"n{x}_{y}", f"n{x + dx}_{y + dy}", w))
    G.add_weighted_edges_from(edges)
    ev = object.__new__(Evaluation)
    ev.error_margin, ev.splits, ev.dataset = 3.0, ["val_unseen"], "R2R"
    ev.distances = {"s": dict(nx.all_pairs_dijkstra_path_length(G))}
    paths = {1: ["n0_0", "n1_0", "n2_0", "n3_0"], 2: ["n0_3", "n1_3", "n1_2", "n2_2", "n3_2"], 3: ["n4_0", "n4_1", "n4_2"],
             4: ["n2_1", "n2_2", "n2_3"]}
    ev.gt = {pid: {"path_id": pid, "scan": "s", "path": p} for pid, p in paths.items()}
    ev.instr_ids = {f"{pid}_0" for pid in paths}
    traj = {"1_0": ["n0_0", "n1_0", "n2_0", "n3_0"],                      # exact
            "2_0": ["n0_3", "n0_2", "n1_2", "n2_2", "n3_2", "n4_2"],      # detour + overshoot
            "3_0": ["n4_0"],                                              # never moved
            "4_0": ["n2_1", "n3_1", "n3_2", "n3_3", "n2_3"]}              # long way round
    results = [{"instr_id": k, "trajectory": [(v, 0.0, 0.0) for v in p]} for k, p in traj.items()]
    summary, scores = ev.score(results)
    out = {"edges": edges, "gt": {f"{pid}_0": {"scan": "s", "path": p} for pid, p in paths.items()}, "results": results,
           "summary": {k: float(v) for k, v in summary.items()}, "scores": {k: [float(x) for x in v] for k, v in scores.items()}}
    with open(os.path.join(OUT, "eval_scores.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("eval_scores:", {k: round(v, 4) for k, v in out["summary"].items()})


def gen_speaker(U):
    """Speaker modules (units.py:286-395), eval mode: encoder over a 5-step path with 36 views per step,
    decoder teacher-forced over 7 words from a zero state and, separately, one word from a carried (non-zero) state."""
    torch.manual_seed(31)
    g = torch.Generator().manual_seed(3131)
    B, Lp, V, IMG, ANG, H, E, VOC, Lw = 3, 5, 36, 24, 8, 32, 16, 40, 7
    F = IMG + ANG
    for bidir in (True, False):
        enc = U.SpeakerEncoder(F, H, 0.5, bidir, ANG, 0.3).eval()
        act = feats(g, B, Lp, IMG, ANG)
        feat = torch.stack([feats(g, B, V, IMG, ANG) for _ in range(Lp)], 1)          # [B, Lp, 36, F]
        r = torch.randn(B, Lp, H, generator=g)
        ctx = enc(act.clone(), feat.clone(), None)
        loss = (ctx * r).sum()
        save(f"speaker_encoder_{'bi' if bidir else 'uni'}", cfg=dict(F=F, H=H, ANG=ANG, bidir=int(bidir)),
             param=dict(enc.state_dict()), inp=dict(act=act, feat=feat, r=r), out=dict(ctx=ctx), grad=grads_of(enc, loss)[0])
    dec = U.SpeakerDecoder(VOC, E, 0, H, 0.5).eval()
    words = torch.randint(1, VOC, (B, Lw), generator=g)
    words[1, 5:] = 0
    ctx = torch.randn(B, Lp, H, generator=g).requires_grad_(True)
    mask = torch.zeros(B, Lp, dtype=torch.bool)
    mask[2, 3:] = True
    h0 = torch.zeros(1, B, H); c0 = torch.zeros(1, B, H)
    logit, h1, c1 = dec(words, ctx, mask, h0, c0)
    r = torch.randn(B, Lw, VOC, generator=g)
    loss = (logit * r).sum() + (h1 * 0.3).sum() + (c1 * 0.2).sum()
    gr, (gctx,) = grads_of(dec, loss, extra=[ctx])
    gr = dict(gr, ctx=gctx)
    hs = torch.randn(1, B, H, generator=g) * 0.5; cs = torch.randn(1, B, H, generator=g) * 0.5
    with torch.no_grad():
        l2, h2, c2 = dec(words[:, :1], ctx, mask, hs, cs)                              # word-by-word inference step
    save("speaker_decoder", cfg=dict(VOC=VOC, E=E, H=H), param=dict(dec.state_dict()),
         inp=dict(words=words, ctx=ctx.detach(), mask=mask, r=r, hs=hs, cs=cs),
         out=dict(logit=logit, h1=h1, c1=c1, step_logit=l2, step_h=h2, step_c=c2), grad=gr)


def gen_speaker_loop(U):
    """The speaker LOOP (agent/speaker.py:235-376 + the back-translation hook envdrop.py:105-121) on the reference's
    own SpeakerEncoder / SpeakerDecoder, eval mode, driven by oracle/rollout.py's restated loop (the reference's Speaker
    object needs the simulator and does not run as shipped): teacher-forcing loss (mean + un-reduced) with gradients,
    greedy inference with a shared environment-dropout mask, instructions as the follower receives them."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from oracle import rollout as R
    torch.manual_seed(int(os.environ.get('SPK_SEED', '75')))
    g = torch.Generator().manual_seed(7707)
    B, Lp, V, IMG, ANG, H, E, VOC, Lw, MAXD = 4, 5, 36, 24, 8, 32, 16, 40, 9, 10
    F = IMG + ANG
    EOS_BIAS = float(os.environ.get('EOS_BIAS', '1.0'))
    ATT_SCALE = float(os.environ.get('ATT_SCALE', '1.5'))
    enc = U.SpeakerEncoder(F, H, 0.5, True, ANG, 0.3).eval()
    dec = U.SpeakerDecoder(VOC, E, 0, H, 0.5).eval()
    with torch.no_grad():
        dec.projection.weight *= 12.0                      # random-init logits barely depend on the input: spread them
        dec.embedding.weight *= 4.0
        dec.attention_layer.linear_out.weight[:, :H] *= ATT_SCALE   # ... and make them depend on the encoded path
        for n_, p_ in enc.named_parameters():
            if 'weight' in n_:
                p_ *= 3.0
        dec.projection.bias[2] += EOS_BIAS                 # make <EOS> likely enough that some rows end early
    can = feats(g, B, Lp, IMG, ANG)
    img = torch.stack([feats(g, B, V, IMG, ANG) for _ in range(Lp)], 1)
    lengths = [5, 4, 4, 2]
    insts = torch.randint(4, VOC, (B, Lw), generator=g)
    insts[:, 0] = 3
    for i, n in enumerate((9, 7, 5, 4)):
        insts[i, n - 1] = 2
        insts[i, n:] = 0
    encode = lambda c, f, l, dropped: enc(c.clone(), f.clone(), l, already_dropfeat=dropped)
    decode = lambda w, ctx, m, h, c: dec(w, ctx, m, h, c)
    loss, per_word, predict = R.speaker_teacher_forcing(encode, decode, can, img, lengths, insts, H)
    gr_e, _ = grads_of(enc, loss)
    loss2, _, _ = R.speaker_teacher_forcing(encode, decode, can, img, lengths, insts, H)
    gr_d, _ = grads_of(dec, loss2)
    noise = (torch.rand(IMG, generator=g) > 0.3).float() / 0.7
    with torch.no_grad():
        words, step_logits = R.speaker_infer_batch(encode, decode, can, img, lengths, H, MAXD, featdropmask=noise, angle=ANG)
        words_plain, _ = R.speaker_infer_batch(encode, decode, can, img, lengths, H, MAXD, angle=ANG)
    bt = R.back_translate_instructions(words)
    print("speaker_loop greedy words:\n", words, "\nback-translated:\n", bt)
    save("speaker_loop", cfg=dict(F=F, H=H, ANG=ANG, VOC=VOC, E=E, MAXD=MAXD),
         enc=dict(enc.state_dict()), dec=dict(dec.state_dict()),
         inp=dict(can=can, img=img, lengths=np.array(lengths), insts=insts, noise=noise),
         out=dict(loss=loss, per_word=per_word, predict=predict, words=words, words_plain=words_plain,
                  step_logits=step_logits, instr_encoding=bt),
         grad_enc=gr_e, grad_dec=gr_d)


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(1)
    U, P = load_reference()
    g = torch.Generator().manual_seed(2020)
    gen_encoder(U, "encoder_envdrop", g, vocab=64, E=32, H=64, bidir=True, layers=1)
    gen_encoder(U, "encoder_follower", g, vocab=64, E=24, H=32, bidir=True, layers=2)
    gen_encoder(U, "encoder_monitor", g, vocab=64, E=32, H=48, bidir=False, layers=1)
    gen_attention(U, g)
    gen_envdrop(P, g, 1, "envdrop_step")
    gen_envdrop(P, g, 3, "envdrop_chain3")
    gen_follower(P, g, 1, "follower_step")
    gen_follower(P, g, 3, "follower_chain3")
    gen_monitor(P, g, True, "monitor_step_train")
    gen_monitor(P, g, False, "monitor_step_eval")
    gen_critic(P, g)
    gen_losses(g)
    gen_speaker(U)
    gen_speaker_loop(U)
    gen_angle_tables()
    gen_agent_tapes()
    gen_agent_tapes_more()
    gen_eval_scores()


if __name__ == "__main__":
    main()
