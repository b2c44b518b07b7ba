"""CPU ORACLE (test infrastructure, NOT product code).

A functional, from-scratch restatement of the reference's navigation-agent hot
path (`/root/reference/tasks/R2R-judy/src/model/{units,policy}.py` and the loss
arithmetic in `src/agent/{follower,envdrop,monitor}.py`) in plain torch CPU ops
over a flat parameter dict keyed by the reference's `state_dict` names.

Who may import this file: `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` -- only as the checker / the timed CPU
baseline.  The product package (`curriculum-learning-for-vln_amd/`) never
imports it and fails loudly if its HIP library is missing.

Parity pinning: the reference ships no tests for this path, so
this oracle is pinned against golden vectors captured by importing the
reference modules in the build container (`oracle/make_goldens.py` ->
`tests/golden/*.npz`; checked by `tests/test_oracle_golden.py`).

Design notes
  * No `nn.Module`, no `nn.LSTM`: the packed (bi)LSTM is an explicit masked
    time loop so that the algorithm the HIP kernels implement is spelled out.
  * Dropout is never sampled here.  Every dropout site takes an optional
    pre-scaled keep mask (`mask * 1/(1-p)`), so the HIP path's Philox masks
    can be exported and injected for exact parity with dropout ON.
  * Works in float32 or float64 (dtype follows the inputs/params).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence, Tuple

import torch

Tensor = torch.Tensor
Params = Dict[str, Tensor]


def _mul(x: Tensor, m: Optional[Tensor]) -> Tensor:
    return x if m is None else x * m


# ---------------------------------------------------------------------------
# A4  LSTM cell  (torch.nn.LSTMCell used at policy.py:30,96,192)
# ---------------------------------------------------------------------------
def lstm_cell(x: Tensor, h: Tensor, c: Tensor, w_ih: Tensor, w_hh: Tensor,
              b_ih: Optional[Tensor], b_hh: Optional[Tensor]) -> Tuple[Tensor, Tensor]:
    """Gate order i,f,g,o; c' = s(f)*c + s(i)*tanh(g); h' = s(o)*tanh(c')."""
    gates = x @ w_ih.t() + h @ w_hh.t()
    if b_ih is not None:
        gates = gates + b_ih
    if b_hh is not None:
        gates = gates + b_hh
    H = h.shape[1]
    i = torch.sigmoid(gates[:, 0 * H:1 * H])
    f = torch.sigmoid(gates[:, 1 * H:2 * H])
    g = torch.tanh(gates[:, 2 * H:3 * H])
    o = torch.sigmoid(gates[:, 3 * H:4 * H])
    c1 = f * c + i * g
    h1 = o * torch.tanh(c1)
    return h1, c1


# ---------------------------------------------------------------------------
# A1  EncoderLSTM  (units.py:12-74)
# ---------------------------------------------------------------------------
def packed_lstm_direction(x: Tensor, lengths: Sequence[int], w_ih, w_hh, b_ih, b_hh,
                          reverse: bool) -> Tuple[Tensor, Tensor, Tensor]:
    """One direction of one layer over a right-padded batch.

    Semantics of pack_padded_sequence -> nn.LSTM -> pad_packed_sequence
    (units.py:58-60,71): a row only advances on its own valid steps t < len;
    the reverse direction therefore starts at the row's last valid token;
    outputs at padded steps are exactly 0; the returned (h, c) are the states
    after the row's last processed token.
    """
    B, L, _ = x.shape
    H = w_hh.shape[1]
    lens = torch.as_tensor(list(lengths), dtype=torch.long)
    h = x.new_zeros(B, H)
    c = x.new_zeros(B, H)
    out = x.new_zeros(B, L, H)
    steps = range(L - 1, -1, -1) if reverse else range(L)
    outs = [None] * L
    for t in steps:
        valid = (lens > t).to(x.dtype).unsqueeze(1)          # [B,1]
        h_new, c_new = lstm_cell(x[:, t], h, c, w_ih, w_hh, b_ih, b_hh)
        h = valid * h_new + (1 - valid) * h
        c = valid * c_new + (1 - valid) * c
        outs[t] = valid * h_new
    out = torch.stack(outs, dim=1)
    return out, h, c


def encoder_forward(P: Params, tokens: Tensor, lengths: Sequence[int], *,
                    num_layers: int, bidirectional: bool,
                    emb_mask: Optional[Tensor] = None,
                    inter_masks: Optional[Sequence[Tensor]] = None,
                    ctx_mask_drop: Optional[Tensor] = None,
                    prefix: str = "") -> Tuple[Tensor, Tensor, Tensor]:
    """EncoderLSTM.forward (units.py:48-74).

    tokens [B,L] int64 (pad=0), lengths sorted descending.  Returns
    (ctx [B,L,H*dirs], decoder_init [B,H*dirs], c_t [B,H*dirs]).
    `emb_mask` = dropout on embeddings (units.py:55-56), `inter_masks[k]` =
    nn.LSTM inter-layer dropout applied to layer k's output for k < layers-1
    (units.py:41), `ctx_mask_drop` = dropout on ctx (units.py:72).
    """
    emb = P[prefix + "embedding.weight"][tokens]              # units.py:54
    x = _mul(emb, emb_mask)
    h_last, c_last = None, None
    for k in range(num_layers):
        outs, hs, cs = [], [], []
        for d in range(2 if bidirectional else 1):
            sfx = f"_l{k}" + ("_reverse" if d == 1 else "")
            o, h, c = packed_lstm_direction(
                x, lengths,
                P[prefix + "lstm.weight_ih" + sfx], P[prefix + "lstm.weight_hh" + sfx],
                P[prefix + "lstm.bias_ih" + sfx], P[prefix + "lstm.bias_hh" + sfx],
                reverse=(d == 1))
            outs.append(o); hs.append(h); cs.append(c)
        x = torch.cat(outs, dim=2)
        h_last, c_last = torch.cat(hs, dim=1), torch.cat(cs, dim=1)   # units.py:63-67
        if k < num_layers - 1 and inter_masks is not None:
            x = _mul(x, inter_masks[k])
    dec_init = torch.tanh(h_last @ P[prefix + "enc2dec.weight"].t() + P[prefix + "enc2dec.bias"])  # units.py:69
    ctx = _mul(x, ctx_mask_drop)                              # units.py:71-72
    return ctx, dec_init, c_last


# ---------------------------------------------------------------------------
# A2  SoftDotAttention (units.py:77-122)
# ---------------------------------------------------------------------------
def masked_softmax(logits: Tensor, mask: Optional[Tensor]) -> Tensor:
    """softmax over dim 1 with -inf at mask==True (units.py:111-114)."""
    if mask is not None:
        logits = logits.masked_fill(mask, -float("inf"))
    return torch.softmax(logits, dim=1)


def softdot_attention(h: Tensor, ctx: Tensor, mask: Optional[Tensor], w_in: Tensor,
                      w_out: Optional[Tensor], score_ctx: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """`w_out is None` == context_only.  Returns (h_tilde or weighted ctx, attn).
    `score_ctx` (default: ctx, the reference) = the copy of the context the LOGITS are taken on when a build under test stores it
    apart from the copy its weighted sum streams (a rounding detail of that build: bf16 stream copy vs fp32 projected copy)."""
    target = h @ w_in.t()                                     # units.py:106
    logits = torch.einsum("bsd,bd->bs", ctx if score_ctx is None else score_ctx, target)          # units.py:109
    attn = masked_softmax(logits, mask)
    wc = torch.einsum("bs,bsd->bd", attn, ctx)                # units.py:117
    if w_out is None:
        return wc, attn                                       # units.py:118
    h_tilde = torch.tanh(torch.cat((wc, h), 1) @ w_out.t())   # units.py:120-121
    return h_tilde, attn


# ---------------------------------------------------------------------------
# A3  VisualSoftDotAttention (units.py:125-160)
# ---------------------------------------------------------------------------
def visual_softdot_attention(h: Tensor, v: Tensor, mask: Optional[Tensor], w_h: Tensor, b_h: Tensor,
                             w_v: Optional[Tensor] = None, b_v: Optional[Tensor] = None):
    target = h @ w_h.t() + b_h                                # units.py:144
    keys = v if w_v is None else v @ w_v.t() + b_v            # units.py:146-147
    logits = torch.einsum("bsd,bd->bs", keys, target)
    attn = masked_softmax(logits, mask)
    wc = torch.einsum("bs,bsd->bd", attn, v)                  # un-projected v, units.py:158-159
    return wc, attn


# ---------------------------------------------------------------------------
# ActionScoring (units.py:163-185)
# ---------------------------------------------------------------------------
def action_scoring(P: Params, prefix: str, cands: Tensor, h_tilde: Tensor) -> Tensor:
    tgt = h_tilde @ P[prefix + "linear_hid.weight"].t() + P[prefix + "linear_hid.bias"]
    key = cands @ P[prefix + "linear_act.weight"].t() + P[prefix + "linear_act.bias"]
    prod = key * tgt.unsqueeze(1)
    return (prod @ P[prefix + "linear_out.weight"].t() + P[prefix + "linear_out.bias"]).squeeze(2)


# ---------------------------------------------------------------------------
# A5  AttnDecoderLSTM.forward (policy.py:37-60) -- Speaker-Follower step
# ---------------------------------------------------------------------------
def follower_step(P: Params, img: Tensor, a_prev: Tensor, cands: Tensor, h0: Tensor, c0: Tensor,
                  ctx: Tensor, ctx_mask: Optional[Tensor], *, drop: Optional[Dict[str, Tensor]] = None,
                 score_ctx: Optional[Tensor] = None):
    drop = drop or {}
    wv, alpha_v = visual_softdot_attention(
        h0, img, None, P["visual_attn.linear_in_h.weight"], P["visual_attn.linear_in_h.bias"],
        P["visual_attn.linear_in_v.weight"], P["visual_attn.linear_in_v.bias"])
    x = _mul(torch.cat((a_prev, wv), 1), drop.get("x"))       # policy.py:50-51
    h1, c1 = lstm_cell(x, h0, c0, P["lstm.weight_ih"], P["lstm.weight_hh"], P["lstm.bias_ih"], P["lstm.bias_hh"])
    h1d = _mul(h1, drop.get("h1"))
    h_tilde, alpha_c = softdot_attention(h1d, ctx, ctx_mask, P["text_attn.linear_in.weight"],
                                         P["text_attn.linear_out.weight"], score_ctx=score_ctx)
    logit = action_scoring(P, "decode_action.", cands, h_tilde)
    return logit, (h1, c1), (alpha_c, alpha_v)


# ---------------------------------------------------------------------------
# A6  EnvDropDecoder.forward (policy.py:208-246)
# ---------------------------------------------------------------------------
def envdrop_step(P: Params, a_prev: Tensor, img: Tensor, cand: Tensor, h_tilde_prev: Tensor, c0: Tensor,
                 ctx: Tensor, ctx_mask: Optional[Tensor], *, drop: Optional[Dict[str, Tensor]] = None,
                 score_ctx: Optional[Tensor] = None):
    """`img`/`cand` are the features AFTER feature dropout (policy.py:226-231
    overwrites the caller's tensors; apply `feature_dropout` first).  h_0 is
    unused by the reference (policy.py:238) and is not an argument here.
    Dropout sites: 'act' (policy.py:224), 'hprev' (:234), 'h1' (:240), 'htilde' (:243).
    """
    drop = drop or {}
    e = torch.tanh(a_prev @ P["act_embed.0.weight"].t() + P["act_embed.0.bias"])
    e = _mul(e, drop.get("act"))
    hq = _mul(h_tilde_prev, drop.get("hprev"))
    vis, alpha_v = softdot_attention(hq, img, None, P["visual_attn.linear_in.weight"], None)
    x = torch.cat((e, vis), 1)
    h1, c1 = lstm_cell(x, h_tilde_prev, c0, P["lstm.weight_ih"], P["lstm.weight_hh"],
                       P["lstm.bias_ih"], P["lstm.bias_hh"])
    h1d = _mul(h1, drop.get("h1"))
    h_tilde, alpha_c = softdot_attention(h1d, ctx, ctx_mask, P["text_attn.linear_in.weight"],
                                         P["text_attn.linear_out.weight"], score_ctx=score_ctx)
    htd = _mul(h_tilde, drop.get("htilde"))
    tgt = htd @ P["cand_attn.weight"].t()                     # policy.py:204
    logit = torch.einsum("bcf,bf->bc", cand, tgt)             # policy.py:205
    return logit, (h1, c1), h_tilde, (alpha_c, alpha_v)


def feature_dropout(x: Tensor, keep_scaled: Tensor, angle: int) -> Tensor:
    """policy.py:228-231: dropout on x[..., :-angle]; the angle tail is untouched."""
    out = x.clone()
    out[..., :-angle] = x[..., :-angle] * keep_scaled
    return out


# ---------------------------------------------------------------------------
# A8  Critic (policy.py:249-267)
# ---------------------------------------------------------------------------
def critic(P: Params, state: Tensor, drop: Optional[Tensor] = None, prefix: str = "", relu_on: Optional[Tensor] = None,
           pre_out: Optional[list] = None) -> Tensor:
    """`relu_on` (bool, default None = the reference's ReLU): the on / off decision of every hidden unit given from outside -- a
    ReLU is a discontinuity, and a build whose input differs in the last bits decides differently for pre-activations near
    zero; with the decisions shared (like a dropout mask) everything else can be compared.  `pre_out`: receives the
    pre-activations (so a test can bound WHERE the decisions differ)."""
    pre = state @ P[prefix + "state2value.0.weight"].t() + P[prefix + "state2value.0.bias"]
    if pre_out is not None:
        pre_out.append(pre.detach())
    z = torch.relu(pre) if relu_on is None else pre * relu_on.to(pre.dtype)
    z = _mul(z, drop)
    return (z @ P[prefix + "state2value.3.weight"].t() + P[prefix + "state2value.3.bias"]).squeeze(1)


# ---------------------------------------------------------------------------
# A7  MonitorDecoder (policy.py:108-166), MLPwithBN (units.py:210-242),
#     PositionalEncoding (units.py:188-207)
# ---------------------------------------------------------------------------
def positional_table(d_model: int, max_len: int, dtype=torch.float32) -> Tensor:
    pe = torch.zeros(max_len, d_model, dtype=torch.float32)
    pos = torch.arange(0, max_len).float().unsqueeze(1)
    div = torch.exp(torch.arange(0, d_model, 2).float() * -(math.log(10000.0) / d_model))
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe.to(dtype)


def batchnorm1d(x: Tensor, w: Tensor, b: Tensor, rm: Tensor, rv: Tensor, training: bool,
                momentum: float = 0.1, eps: float = 1e-5):
    """Returns (y, new_running_mean, new_running_var). Train mode normalises by
    the biased batch variance and updates running_var with the unbiased one."""
    if training:
        n = x.shape[0]
        mean = x.mean(0)
        var = x.var(0, unbiased=False)
        y = (x - mean) / torch.sqrt(var + eps) * w + b
        new_rm = (1 - momentum) * rm + momentum * mean.detach()
        new_rv = (1 - momentum) * rv + momentum * (var.detach() * n / max(n - 1, 1))
        return y, new_rm, new_rv
    y = (x - rm) / torch.sqrt(rv + eps) * w + b
    return y, rm, rv


def bn_mlp(P: Params, prefix: str, x: Tensor, training: bool, drop: Optional[Tensor] = None):
    """MLPwithBN(use_bn=True, one hidden layer): BN -> Linear -> BN -> Dropout -> ReLU.
    Returns (y, stats) with stats = updated running stats for both BNs."""
    y0, rm0, rv0 = batchnorm1d(x, P[prefix + "mlp.0.weight"], P[prefix + "mlp.0.bias"],
                               P[prefix + "mlp.0.running_mean"], P[prefix + "mlp.0.running_var"], training)
    z = y0 @ P[prefix + "mlp.1.weight"].t() + P[prefix + "mlp.1.bias"]
    y1, rm1, rv1 = batchnorm1d(z, P[prefix + "mlp.2.weight"], P[prefix + "mlp.2.bias"],
                               P[prefix + "mlp.2.running_mean"], P[prefix + "mlp.2.running_var"], training)
    y1 = _mul(y1, drop)
    return torch.relu(y1), {"rm0": rm0, "rv0": rv0, "rm1": rm1, "rv1": rv1}


def monitor_step(P: Params, a_prev: Tensor, cands: Tensor, h0: Tensor, c0: Tensor, ctx: Tensor,
                 ctx_mask: Optional[Tensor], cand_mask: Tensor, *, training: bool,
                 drop: Optional[Dict[str, Tensor]] = None):
    """MonitorDecoder.forward (policy.py:132-166).  Dropout sites: 'mlp_prev',
    'mlp_cands' (units.py MLP dropout), 'pe' (units.py:207), 'h1' (policy.py:160),
    'pm' (policy.py:128).  BN running stats are updated twice (prev rows, then
    B*C candidate rows incl. zero-padded ones); the second call sees the stats
    written by the first."""
    drop = drop or {}
    P = dict(P)
    B, C, F = cands.shape
    proj_prev, st = bn_mlp(P, "proj_navigable_mlp.", a_prev, training, drop.get("mlp_prev"))
    if training:
        P["proj_navigable_mlp.mlp.0.running_mean"], P["proj_navigable_mlp.mlp.0.running_var"] = st["rm0"], st["rv0"]
        P["proj_navigable_mlp.mlp.2.running_mean"], P["proj_navigable_mlp.mlp.2.running_var"] = st["rm1"], st["rv1"]
    proj_c, st2 = bn_mlp(P, "proj_navigable_mlp.", cands.reshape(B * C, F), training, drop.get("mlp_cands"))
    proj_c = proj_c.reshape(B, C, -1) * (1 - cand_mask.to(proj_c.dtype)).unsqueeze(2)   # policy.py:148-149
    L = ctx.shape[1]
    pctx = _mul(ctx + P["position.pe"][0, :L].to(ctx.dtype), drop.get("pe"))          # units.py:205-207
    w_ctx, ctx_attn = softdot_attention(h0, pctx, ctx_mask, P["text_attn.linear_in.weight"], None)
    w_cands, cand_attn = visual_softdot_attention(h0, proj_c, cand_mask, P["visual_attn.linear_in_h.weight"],
                                                  P["visual_attn.linear_in_h.bias"])
    x = torch.cat((proj_prev, w_cands, w_ctx), 1)
    h1, c1 = lstm_cell(x, h0, c0, P["lstm.weight_ih"], P["lstm.weight_hh"], P["lstm.bias_ih"], P["lstm.bias_hh"])
    h1d = _mul(h1, drop.get("h1"))
    ht = torch.cat((w_ctx, h1d), 1) @ P["action_linear.weight"].t() + P["action_linear.bias"]   # policy.py:115
    logit = torch.einsum("bcm,bm->bc", proj_c, ht)
    pm_in = torch.cat((h0, w_cands), 1) @ P["monitor_linear.weight"].t() + P["monitor_linear.bias"]
    h_pm = _mul(torch.sigmoid(pm_in) * torch.tanh(c1), drop.get("pm"))                 # policy.py:128
    prog = torch.tanh(torch.cat((ctx_attn, h_pm), 1) @ P["critic.0.weight"].t() + P["critic.0.bias"]).squeeze(1)
    stats = st2 if training else None
    return (logit, prog), (h1, c1), (ctx_attn, cand_attn), stats


# ---------------------------------------------------------------------------
# A9  losses and action selection (follower.py:123-139, envdrop.py:173-270,
#     monitor.py:146-176) and A10 helpers (misc.py:481-486)
# ---------------------------------------------------------------------------
def length2mask(lengths: Sequence[int], size: Optional[int] = None) -> Tensor:
    """mask[i,j] = j >= len_i  (misc.py:481-486)."""
    size = int(max(lengths)) if size is None else size
    ar = torch.arange(size).unsqueeze(0)
    return ar >= torch.as_tensor(list(lengths)).unsqueeze(1)


# ---------------------------------------------------------------------------
# A10  per-step marshalling (agent/base.py:141-157, environ/common_env.py:307-308,
#      utils/misc.py:285-317): [36 x 2048 view features | 36 x 128 angle features]
# ---------------------------------------------------------------------------
def angle_feat(heading: float, elevation: float, feat_size: int = 128) -> Tensor:
    """[sin h]*n, [cos h]*n, [sin e]*n, [cos e]*n with n = feat_size/4  (misc.py:285-293)."""
    n = feat_size // 4
    v = torch.tensor([math.sin(heading), math.cos(heading), math.sin(elevation), math.cos(elevation)], dtype=torch.float32)
    return v.repeat_interleave(n)


def loc_embedding_table(feat_size: int = 128) -> Tensor:
    """table[viewIndex, absView] = angle feature of view absView seen from viewIndex (misc.py:296-317):
    12 headings x 3 elevations, 30 degrees apart; heading is relative to the agent's view, elevation absolute."""
    inc = math.pi / 6.0
    t = torch.zeros(36, 36, feat_size)
    for vi in range(36):
        for av in range(36):
            rel = (av - vi) % 12 + (av // 12) * 12
            t[vi, av] = angle_feat((rel % 12) * inc, (rel // 12 - 1) * inc, feat_size)
    return t


def gather_pano(table: Tensor, rows: Tensor, view_index: Tensor, angle_table: Tensor) -> Tensor:
    """img_feature [B,36,IMG+ANG] = [features of the viewpoint | angle features for the current viewIndex]
    (common_env.py:307-308 + base.py:141-147)."""
    return torch.cat((table[rows].float(), angle_table[view_index]), dim=-1)


def gather_cands(table: Tensor, rows: Tensor, views: Tensor, heading: Tensor, elevation: Tensor, angle: int = 128) -> Tensor:
    """cand_feature [B,C,IMG+ANG]; rows[b,c] < 0 marks the STOP slot / padding, which stay all-zero
    (base.py:149-157; candidate feature = view feature | make_angle_feat(loc heading, loc elevation),
    common_env.py:272,287-291)."""
    B, C = rows.shape
    IMG = table.shape[-1]
    out = torch.zeros(B, C, IMG + angle)
    for b in range(B):
        for c in range(C):
            if rows[b, c] >= 0:
                out[b, c, :IMG] = table[rows[b, c], views[b, c]].float()
                out[b, c, IMG:] = angle_feat(float(heading[b, c]), float(elevation[b, c]), angle)
    return out


def masked_cross_entropy(logits: Tensor, target: Tensor, cand_mask: Optional[Tensor],
                         reduction: str = "none", ignore_index: int = -1) -> Tensor:
    """logits.masked_fill_(-inf) + CrossEntropyLoss(ignore_index=-1).
    reduction: 'none' -> [B] (0 at ignored rows), 'sum', 'mean' (mean over
    non-ignored rows; NaN if none, like torch)."""
    if cand_mask is not None:
        logits = logits.masked_fill(cand_mask, -float("inf"))
    lse = torch.logsumexp(logits, dim=1)
    valid = target != ignore_index
    tsafe = torch.where(valid, target, torch.zeros_like(target))
    picked = logits.gather(1, tsafe.unsqueeze(1)).squeeze(1)
    per = torch.where(valid, lse - picked, torch.zeros_like(lse))
    if reduction == "none":
        return per
    if reduction == "sum":
        return per.sum()
    return per.sum() / valid.sum().to(per.dtype)


def categorical_logprob_entropy(logits: Tensor, action: Tensor, eps: float = 1.1920928955078125e-07):
    """Categorical(probs=softmax(logits)): log_prob(action), entropy
    (envdrop.py:189-194; torch.distributions clamps probs to [eps, 1-eps])."""
    p = torch.softmax(logits, dim=1)
    p = p / p.sum(-1, keepdim=True)
    logp = torch.log(p.clamp(min=eps, max=1 - eps))
    lp = logp.gather(1, action.unsqueeze(1)).squeeze(1)
    ent = -(p * logp).sum(1)
    return lp, ent


def a2c_loss(log_probs: Sequence[Tensor], entropies: Sequence[Tensor], values: Sequence[Tensor],
             rewards: Sequence[Tensor], masks: Sequence[Tensor], last_value: Tensor, ended: Tensor,
             gamma: float, normalize: str = "total", per_sample: bool = False):
    """envdrop.py:235-264.  rewards/masks [T][B]; values[t] = critic(h_1 at t)
    (with grad); last_value is detached.  Returns (loss, total)."""
    T = len(rewards)
    B = rewards[0].shape[0]
    R = (~ended).to(last_value.dtype) * last_value.detach()
    loss = torch.zeros(B, dtype=values[0].dtype, device=values[0].device)
    total = 0.0
    for t in range(T - 1, -1, -1):
        R = R * gamma + rewards[t]
        m = masks[t].to(values[t].dtype)
        adv = (R - values[t]).detach()
        cur = -log_probs[t] * adv * m + 0.5 * ((R - values[t]) ** 2) * m - 0.01 * entropies[t] * m
        loss = loss + cur
        total = total + float(masks[t].sum())
    if not per_sample:
        loss = loss.sum()
    if normalize == "total":
        loss = loss / total
    elif normalize == "batch":
        loss = loss / B
    return loss, total


def monitor_mixed_loss(logits: Tensor, target: Tensor, cand_mask: Tensor, progress: Tensor,
                       progress_target: Tensor, t: int, lam: float, per_sample: bool = False) -> Tensor:
    """monitor.py:146-165: t==0 -> CE only; else lam*MSE + (1-lam)*CE.
    Non-CL: CE is the mean over non-ignored rows, MSE the mean over all B rows."""
    if per_sample:
        ce = masked_cross_entropy(logits, target, cand_mask, "none")
        if t == 0:
            return ce
        return lam * (progress - progress_target) ** 2 + (1 - lam) * ce
    ce = masked_cross_entropy(logits, target, cand_mask, "mean")
    if t == 0:
        return ce
    return lam * torch.mean((progress - progress_target) ** 2) + (1 - lam) * ce


# ---------------------------------------------------------------------------
# N3  Speaker modules (units.py:286-395): SpeakerEncoder / SpeakerDecoder
# ---------------------------------------------------------------------------
def _plain_lstm(P: Params, prefix: str, x: Tensor, bidirectional: bool, h0: Optional[Tensor] = None,
                c0: Optional[Tensor] = None):
    """nn.LSTM(batch_first=True, 1 layer) on an UNPACKED batch (every row runs all L steps, units.py:325,338,366)."""
    B, L, _ = x.shape
    outs, hs, cs = [], [], []
    for d in range(2 if bidirectional else 1):
        sfx = "_l0" + ("_reverse" if d == 1 else "")
        w_ih, w_hh = P[prefix + "weight_ih" + sfx], P[prefix + "weight_hh" + sfx]
        b_ih, b_hh = P[prefix + "bias_ih" + sfx], P[prefix + "bias_hh" + sfx]
        H = w_hh.shape[1]
        h = x.new_zeros(B, H) if h0 is None else h0[d]
        c = x.new_zeros(B, H) if c0 is None else c0[d]
        o = [None] * L
        for t in (range(L - 1, -1, -1) if d == 1 else range(L)):
            h, c = lstm_cell(x[:, t], h, c, w_ih, w_hh, b_ih, b_hh)
            o[t] = h
        outs.append(torch.stack(o, 1)); hs.append(h); cs.append(c)
    return torch.cat(outs, 2), torch.stack(hs, 0), torch.stack(cs, 0)


def speaker_encoder(P: Params, action_embeds: Tensor, feature: Tensor, bidirectional: bool, *,
                    drop: Optional[Dict[str, Tensor]] = None) -> Tensor:
    """SpeakerEncoder.forward (units.py:313-341); `action_embeds` / `feature` are the tensors AFTER the feature dropout.
    Dropout sites: 'ctx' (:326), 'att' (:336), 'out' (:339)."""
    drop = drop or {}
    B, Lp, _ = action_embeds.shape
    ctx, _, _ = _plain_lstm(P, "lstm.", action_embeds, bidirectional)
    ctx = _mul(ctx, drop.get("ctx"))
    H = ctx.shape[-1]
    x, _ = softdot_attention(ctx.reshape(B * Lp, H), feature.reshape(B * Lp, feature.shape[2], feature.shape[3]), None,
                             P["attention_layer.linear_in.weight"], P["attention_layer.linear_out.weight"])
    x = _mul(x.reshape(B, Lp, -1), drop.get("att"))
    x, _, _ = _plain_lstm(P, "post_lstm.", x, bidirectional)
    return _mul(x, drop.get("out"))


def speaker_decoder(P: Params, words: Tensor, ctx: Tensor, ctx_mask: Optional[Tensor], h0: Tensor, c0: Tensor, *,
                    drop: Optional[Dict[str, Tensor]] = None, padding_idx: int = 0):
    """SpeakerDecoder.forward (units.py:363-395).  Dropout sites: 'emb' (:365), 'lstm' (:368), 'att' (:392)."""
    drop = drop or {}
    Bw, Lw = words.shape
    emb = torch.nn.functional.embedding(words, P["embedding.weight"], padding_idx)   # the pad row gets no gradient (:352)
    emb = _mul(emb, drop.get("emb"))
    x, h1, c1 = _plain_lstm(P, "lstm.", emb, False, h0, c0)
    x = _mul(x, drop.get("lstm"))
    H = x.shape[-1]
    n = Bw * Lw
    mult = n // ctx.shape[0]
    ctx_e = ctx.unsqueeze(1).expand(-1, mult, -1, -1).reshape(n, ctx.shape[1], H)
    mask_e = ctx_mask.unsqueeze(1).expand(-1, mult, -1).reshape(n, -1) if ctx_mask is not None else None
    x, _ = softdot_attention(x.reshape(n, H), ctx_e, mask_e, P["attention_layer.linear_in.weight"],
                             P["attention_layer.linear_out.weight"])
    x = _mul(x.reshape(Bw, Lw, H), drop.get("att"))
    logit = x @ P["projection.weight"].t() + P["projection.bias"]
    return logit, h1, c1
