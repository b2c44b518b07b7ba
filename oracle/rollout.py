"""Rollout-level restatement of `EnvDropAgent.rollout()` (reference src/agent/envdrop.py:86-278) and of the
obs -> tensor marshalling it uses (src/agent/base.py:114-178) -- TEST INFRASTRUCTURE.

The loop is written against a tiny backend interface so the SAME harness drives
  * the CPU oracle (`OracleBackend`, oracle/torch_port.py functions over parameter dicts) and
  * the HIP modules (`ModuleBackend`, vln_amd.EncoderLSTM / EnvDropDecoder / Critic on the GPU),
both fed by `oracle/fake_env.py` and pinned by `tests/golden/agent_envdrop_{teacher,sample}.npz`, which were
captured from the reference's own agent.  Sampled actions cannot be RNG-matched, so the tape's actions are injected.
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np
import torch

from . import torch_port as O
from .fake_env import ANG, angle_feat


def marshal_instructions(obs, device):
    """base.py:114-139: token matrix trimmed to the longest instruction, pad mask, lengths (already sorted desc)."""
    seq = np.array([ob["instr_encoding"] for ob in obs])
    lens = np.array([ob["instr_length"] for ob in obs])
    seq = seq[:, :lens[0]]
    tokens = torch.from_numpy(seq).long().to(device)
    return tokens, (tokens == 0), torch.from_numpy(lens)


def marshal_step(obs, device):
    """envdrop.py:75-84 + base.py:141-157: angle input, [B,36,F] view features, zero-padded candidates (+STOP slot)."""
    B = len(obs)
    a = np.stack([angle_feat(ob["heading"], ob["elevation"]) for ob in obs])
    img = np.stack([ob["feature"] for ob in obs]).astype(np.float32)
    cl = [len(ob["candidates"]) + 1 for ob in obs]
    cand = np.zeros((B, max(cl), img.shape[-1]), np.float32)
    for i, ob in enumerate(obs):
        for j, c in enumerate(ob["candidates"]):
            cand[i, j] = c["feature"]
    t = lambda x: torch.from_numpy(x).to(device)
    return t(a), t(img), t(cand), cl


def teacher_action(obs, ended):
    """base.py:159-178."""
    a = np.zeros(len(obs), np.int64)
    for i, ob in enumerate(obs):
        if ended[i]:
            a[i] = -1
            continue
        for k, c in enumerate(ob["candidates"]):
            if c["nextViewpointId"] == ob["teacher"]:
                a[i] = k
                break
        else:
            assert ob["teacher"] == ob["viewpointId"]
            a[i] = len(ob["candidates"])
    return a


class OracleBackend:
    def __init__(self, P_enc, P_dec, P_cri, dtype=torch.float64):
        cv = lambda P: {k: (v.to(dtype) if v.is_floating_point() else v).clone().requires_grad_(v.is_floating_point()) for k, v in P.items()}
        self.Pe, self.Pd, self.Pc = cv(P_enc), cv(P_dec), cv(P_cri)
        self.dtype, self.device = dtype, torch.device("cpu")

    def encode(self, tokens, lengths):
        return O.encoder_forward(self.Pe, tokens, lengths.tolist(), num_layers=1, bidirectional=True)

    def decode(self, a, img, cand, h_tilde, h, c, ctx, mask, dropped=False):
        d = self.dtype                     # no dropout masks are injected here: `dropped` (already_dropfeat) changes nothing
        logit, (h1, c1), ht, _ = O.envdrop_step(self.Pd, a.to(d), img.to(d), cand.to(d), h_tilde, c, ctx, mask)
        return logit, h1, c1, ht

    def critic(self, h):
        return O.critic(self.Pc, h)

    def named_grads(self):
        out = {}
        for pre, P in (("enc.", self.Pe), ("dec.", self.Pd), ("cri.", self.Pc)):
            for k, v in P.items():
                if v.is_floating_point():
                    out[pre + k] = v.grad if v.grad is not None else torch.zeros_like(v)
        return out


class ModuleBackend:
    """nn.Modules with the reference's forward contracts (the HIP drop-ins, or the reference's own modules)."""

    def __init__(self, enc, dec, cri, device, a2c_loss=None):
        self.enc, self.dec, self.cri, self.device = enc, dec, cri, device
        self.a2c_loss = a2c_loss          # optional replacement of torch_port.a2c_loss (losses.a2c_loss on the HIP path)

    def encode(self, tokens, lengths):
        return self.enc(tokens, lengths)

    def decode(self, a, img, cand, h_tilde, h, c, ctx, mask, dropped=False):
        logit, (h1, c1), ht = self.dec(a, img, cand, h_tilde, h, c, ctx, mask, dropped)
        return logit, h1, c1, ht

    def critic(self, h):
        return self.cri(h)

    def named_grads(self):
        out = {}
        for pre, m in (("enc.", self.enc), ("dec.", self.dec), ("cri.", self.cri)):
            if m is None:
                continue
            for k, p in m.named_parameters():
                out[pre + k] = p.grad if p.grad is not None else torch.zeros_like(p)
        return out


def envdrop_rollout(be, env, feedback: str, episode_len: int, inject_actions: Optional[np.ndarray] = None,
                    train_rl: bool = False, ml_weight=0.2, gamma=0.9, insts: Optional[np.ndarray] = None,
                    noise: Optional[torch.Tensor] = None):
    """Returns dict(ml_loss, rl_loss, total, loss, actions).  envdrop.py:86-278 with train_cl=False.
    Back translation (envdrop.py:105-121,155-157; `speaker is not None` there): `insts` [B, n] = the generated
    instructions (<BOS> .. <EOS> <PAD>..) that replace the batch's own, `noise` = the shared environment-dropout mask
    multiplied into the image part of every step's features, the decoder then being told `already_dropfeat`.  The
    reference leaves the stale `instr_length` in the batch and does not re-sort it (its `reset(batch=...)` path,
    common_env.py:332-343, skips the sort of `_next_minibatch`), which the packed encoder cannot take; as in the upstream
    EnvDrop agent the lengths are recomputed from the tokens and the batch is put in descending-length order by a
    permutation that is undone when the actions go back to the environment."""
    dev = be.device
    if feedback != "sample":
        train_rl = False
    env_obs = env.reset(restart=False)
    B = len(env_obs)
    perm = np.arange(B)
    if insts is not None:
        lens = np.array([int(np.argmax(r == 0)) if (r == 0).any() else len(r) for r in insts])
        perm = np.argsort(-lens, kind="stable")
        for ob, r, n in zip(env_obs, insts, lens):          # what reset(batch=...) hands back after the hook edited the batch
            ob["instr_encoding"], ob["instr_length"] = r, int(n)
    inv = np.argsort(perm)
    P = lambda o: [o[i] for i in perm]
    obs = P(env_obs)
    traj = [{"instr_id": ob["instr_id"], "path": [(ob["viewpointId"], ob["heading"], ob["elevation"])]} for ob in env_obs]
    tokens, seq_mask, lengths = marshal_instructions(obs, dev)
    ctx, h_t, c_t = be.encode(tokens, lengths)
    ended = np.zeros(B, bool)
    last_dist = np.array([ob["distance"] for ob in obs], np.float32)
    rewards, hidden, logps, masks, ents, acts = [], [], [], [], [], []
    ml = 0.0
    h_tilde = h_t
    for t in range(episode_len):
        a_in, img, cand, cl = marshal_step(obs, dev)
        if noise is not None:                                       # envdrop.py:155-157
            img[..., :-ANG] *= noise.to(img.dtype)
            cand[..., :-ANG] *= noise.to(cand.dtype)
        logit, h_t, c_t, h_tilde = be.decode(a_in, img, cand, h_tilde, h_t, c_t, ctx, seq_mask, noise is not None)
        hidden.append(h_t)
        cmask = O.length2mask(cl).to(dev)
        logit = logit.masked_fill(cmask, -float("inf"))            # envdrop.py:173 (in place there)
        target = torch.from_numpy(teacher_action(obs, ended)).to(dev)
        ml = ml + O.masked_cross_entropy(logit, target, None, "sum")
        if feedback == "teacher":
            a_t = target
        elif feedback == "argmax":                                  # envdrop.py:184-187 (student forcing; what test() runs)
            a_t = logit.max(1)[1]
        else:
            a_t = torch.from_numpy(np.where(inject_actions[t] < 0, 0, inject_actions[t])).to(dev) if inject_actions is not None \
                else torch.distributions.Categorical(torch.softmax(logit, 1)).sample()
            if inject_actions is not None:
                # the tape stores post-processed actions (-1 = stop/ended); the sampled index of a stop is the STOP slot
                stop_idx = torch.tensor([len(ob["candidates"]) for ob in obs], device=dev)
                a_t = torch.where(torch.from_numpy(inject_actions[t] < 0).to(dev), stop_idx, a_t)
            lp, en = O.categorical_logprob_entropy(logit, a_t)
            logps.append(lp); ents.append(en)
        cpu_a = a_t.detach().cpu().numpy().copy()                   # a COPY (the reference's CPU aliasing bug)
        for i in range(B):
            if cpu_a[i] == len(obs[i]["candidates"]) or cpu_a[i] == -1 or ended[i]:
                cpu_a[i] = -1
        acts.append(cpu_a.copy())
        env_obs = env.step(cpu_a[inv], env_obs, traj)               # actions back in the environment's order
        obs = P(env_obs)
        dist = np.array([ob["distance"] for ob in obs], np.float32)
        is_stop = cpu_a == -1
        reward = (is_stop * (2 * (dist < 3) - 1) * 2 + (1 - is_stop) * np.sign(last_dist - dist)) * (~ended)   # envdrop.py:209-212
        rewards.append(reward.astype(np.float32)); masks.append(~ended)
        last_dist[:] = dist
        ended[:] = np.logical_or(ended, is_stop)
        if ended.all():
            break
    rl, total = 0.0, 0
    if train_rl:
        a_in, img, cand, cl = marshal_step(obs, dev)
        _, last_h, _, _ = be.decode(a_in, img, cand, h_tilde, h_t, c_t, ctx, seq_mask)
        with torch.no_grad():
            last_v = be.critic(last_h).detach()
        vals = [be.critic(h) for h in hidden]
        dt = vals[0].dtype
        a2c = getattr(be, "a2c_loss", None) or O.a2c_loss          # a backend may bring its own (the fused HIP sweep)
        rl, total = a2c(logps, ents, vals, [torch.from_numpy(r).to(dev).to(dt) for r in rewards],
                        [torch.from_numpy(m).to(dev) for m in masks], last_v, torch.from_numpy(ended.copy()).to(dev),
                        gamma, "total")
        total = float(total)
    ml_loss = ml * ml_weight / B
    return dict(ml_loss=ml_loss, rl_loss=rl, total=total, loss=ml_loss + rl, actions=np.stack(acts), traj=traj)


# ---------------------------------------------------------------------------------------------------------------------
# FollowerAgent.rollout (follower.py:67-175) and SelfMonitorAgent.rollout (monitor.py:88-199), train_cl=False
# ---------------------------------------------------------------------------------------------------------------------
class FollowerOracle:
    def __init__(self, P_enc, P_dec, layers=2, bidirectional=True, dtype=torch.float64):
        cv = lambda P: {k: (v.to(dtype) if v.is_floating_point() else v).clone().requires_grad_(v.is_floating_point()) for k, v in P.items()}
        self.Pe, self.Pd = cv(P_enc), cv(P_dec)
        self.layers, self.bi, self.dtype, self.device = layers, bidirectional, dtype, torch.device("cpu")

    def encode(self, tokens, lengths):
        return O.encoder_forward(self.Pe, tokens, lengths.tolist(), num_layers=self.layers, bidirectional=self.bi)

    def decode(self, img, a_prev, cand, h, c, ctx, mask):
        d = self.dtype
        logit, (h1, c1), _ = O.follower_step(self.Pd, img.to(d), a_prev.to(d), cand.to(d), h, c, ctx, mask)
        return logit, h1, c1

    def named_grads(self):
        return {pre + k: (v.grad if v.grad is not None else torch.zeros_like(v)) for pre, P in (("enc.", self.Pe), ("dec.", self.Pd))
                for k, v in P.items() if v.is_floating_point() and v.requires_grad}


class MonitorOracle(FollowerOracle):
    """eval-mode BatchNorm (running statistics), like the captured tapes (agent.eval())."""

    def __init__(self, P_enc, P_dec, dtype=torch.float64):
        super().__init__(P_enc, P_dec, layers=1, bidirectional=False, dtype=dtype)
        for k, v in self.Pd.items():                 # buffers are not trained
            if "running_" in k or k.endswith("position.pe"):
                v.requires_grad_(False)

    def decode(self, a_prev, cand, h, c, ctx, mask, cmask):
        d = self.dtype
        (logit, prog), (h1, c1), _, _ = O.monitor_step(self.Pd, a_prev.to(d), cand.to(d), h, c, ctx, mask, cmask, training=False)
        return logit, prog, h1, c1


class FollowerModules:
    """nn.Modules with the reference's contracts (policy.py:37-60): the HIP drop-ins on the GPU."""

    def __init__(self, enc, dec, device):
        self.enc, self.dec, self.device = enc, dec, device

    def encode(self, tokens, lengths):
        return self.enc(tokens, lengths)

    def decode(self, img, a_prev, cand, h, c, ctx, mask):
        logit, (h1, c1), _ = self.dec(img, a_prev, cand, h, c, ctx, mask)
        return logit, h1, c1

    def named_grads(self):
        return {pre + k: (p.grad if p.grad is not None else torch.zeros_like(p)) for pre, m in (("enc.", self.enc), ("dec.", self.dec))
                for k, p in m.named_parameters()}


class MonitorModules(FollowerModules):
    def __init__(self, enc, dec, device, mixed_loss=None):
        super().__init__(enc, dec, device)
        if mixed_loss is not None:
            self.mixed_loss = mixed_loss      # optional replacement of the step-loss sequence (losses.monitor_mixed_loss on the HIP path)

    def decode(self, a_prev, cand, h, c, ctx, mask, cmask):
        (logit, prog), (h1, c1), _ = self.dec(None, a_prev, cand, h, c, ctx, mask, cmask)
        return logit, prog, h1, c1


def _select(feedback, logit, target, obs, inject, t, dev):
    if feedback == "teacher":
        return target
    if inject is not None:                          # tape actions are post-processed (-1 = stop / ended)
        stop_idx = torch.tensor([len(ob["candidates"]) for ob in obs], device=dev)
        a = torch.from_numpy(np.where(inject[t] < 0, 0, inject[t])).to(dev)
        return torch.where(torch.from_numpy(inject[t] < 0).to(dev), stop_idx, a)
    if feedback == "argmax":
        return logit.max(1)[1]
    return torch.distributions.Categorical(torch.softmax(logit, 1)).sample()


def _post(a_t, obs, ended):
    cpu_a = a_t.detach().cpu().numpy().copy()       # a COPY
    for i in range(len(obs)):
        if cpu_a[i] == len(obs[i]["candidates"]) or cpu_a[i] == -1 or ended[i]:
            cpu_a[i] = -1
    return cpu_a


def follower_rollout(be, env, feedback: str, episode_len: int, inject_actions: Optional[np.ndarray] = None):
    """follower.py:67-175: ml_loss = sum_t CrossEntropy(mean over non-ended rows)."""
    dev = be.device
    obs = env.reset(restart=False)
    B = len(obs)
    traj = [{"instr_id": ob["instr_id"], "path": [(ob["viewpointId"], ob["heading"], ob["elevation"])]} for ob in obs]
    tokens, seq_mask, lengths = marshal_instructions(obs, dev)
    ctx, h_t, c_t = be.encode(tokens, lengths)
    F = obs[0]["feature"].shape[-1]
    a_prev = torch.zeros(B, F, device=dev)
    ended = np.zeros(B, bool)
    ml, acts = 0.0, []
    for t in range(episode_len):
        _, img, cand, cl = marshal_step(obs, dev)
        logit, h_t, c_t = be.decode(img, a_prev, cand, h_t, c_t, ctx, seq_mask)
        cmask = O.length2mask(cl).to(dev)
        logit = logit.masked_fill(cmask, -float("inf"))                       # follower.py:123
        target = torch.from_numpy(teacher_action(obs, ended)).to(dev)
        ml = ml + O.masked_cross_entropy(logit, target, None, "mean")         # follower.py:62,127
        a_t = _select(feedback, logit, target, obs, inject_actions, t, dev)
        cpu_a = _post(a_t, obs, ended)
        acts.append(cpu_a.copy())
        obs = env.step(cpu_a, obs, traj)
        a_prev = cand[torch.arange(B, device=dev), torch.from_numpy(np.maximum(cpu_a, 0)).to(dev)].detach()   # follower.py:164
        ended[:] = np.logical_or(ended, cpu_a == -1)
        if ended.all():
            break
    return dict(ml_loss=ml, actions=np.stack(acts), traj=traj)


def monitor_rollout(be, env, feedback: str, episode_len: int, lamb: float = 0.5, inject_actions: Optional[np.ndarray] = None):
    """monitor.py:88-199: co-grounding step + progress monitor; loss = CE at t=0, lamb*MSE + (1-lamb)*CE after."""
    dev = be.device
    obs = env.reset(restart=False)
    B = len(obs)
    traj = [{"instr_id": ob["instr_id"], "path": [(ob["viewpointId"], ob["heading"], ob["elevation"])]} for ob in obs]
    seq = np.array([ob["instr_encoding"] for ob in obs])                      # monitor.py:68-86: NOT trimmed to the longest
    tokens = torch.from_numpy(seq).long().to(dev)
    seq_mask = tokens == 0
    lengths = torch.from_numpy(np.array([ob["instr_length"] for ob in obs]))
    ctx, h_t, c_t = be.encode(tokens, lengths)
    F = obs[0]["feature"].shape[-1]
    a_prev = torch.zeros(B, F, device=dev)
    ended = np.zeros(B, bool)
    start = np.array([ob["distance"] for ob in obs], np.float32)
    cur = start.copy()
    ml, prog_log, acts = 0.0, 0.0, []
    for t in range(episode_len):
        _, _, cand, cl = marshal_step(obs, dev)
        cmask = O.length2mask(cl).to(dev)
        raw_logit, prog, h_t, c_t = be.decode(a_prev, cand, h_t, c_t, ctx, seq_mask, cmask)
        logit = raw_logit.masked_fill(cmask, -float("inf"))
        target = torch.from_numpy(teacher_action(obs, ended)).to(dev)
        fused = getattr(be, "mixed_loss", None)
        if fused is not None:
            # the backend's own step loss (losses.monitor_mixed_loss on the HIP path): raw logits + mask + distances in, the
            # progress target is built on the device
            loss_t, pmse = fused(raw_logit, target, cmask, prog, torch.from_numpy(start).to(dev), torch.from_numpy(cur.copy()).to(dev),
                                 torch.from_numpy(ended.copy()).to(dev), t, lamb)
            if t > 0:
                prog_log += float(pmse)
            ml = ml + loss_t
        else:
            pt = (start - cur) / start                                            # monitor.py:154-157
            pt[cur <= 3.0] = 1.0
            pt[ended] = prog.detach().cpu().numpy()[ended]
            pt_t = torch.from_numpy(pt.astype(np.float32)).to(dev).to(prog.dtype)
            if t > 0:
                prog_log += float(torch.mean((prog.detach() - pt_t) ** 2))
            ml = ml + O.monitor_mixed_loss(logit, target, None, prog, pt_t, t, lamb)
        a_t = _select(feedback, logit, target, obs, inject_actions, t, dev)
        cpu_a = _post(a_t, obs, ended)
        acts.append(cpu_a.copy())
        obs = env.step(cpu_a, obs, traj)
        cur[:] = np.array([ob["distance"] for ob in obs], np.float32)
        ended[:] = np.logical_or(ended, cpu_a == -1)
        a_prev = cand[torch.arange(B, device=dev), torch.from_numpy(np.maximum(cpu_a, 0)).to(dev)].detach()
        if ended.all():
            break
    return dict(ml_loss=ml, progress_loss=prog_log, actions=np.stack(acts), traj=traj)


# ---------------------------------------------------------------------------------------------------------------
# N3  Speaker loop (reference src/agent/speaker.py:235-376, hook src/agent/envdrop.py:105-121) -- TEST INFRASTRUCTURE.
# Written over two callables so the SAME loop drives the reference's own modules (oracle/make_goldens.py, which is
# how tests/golden/speaker_loop.npz was captured: the reference's Speaker object is tied to the simulator and does
# not run as shipped) and the CPU restatement (`SpeakerOracle`).
#   encode(can_feats, img_feats, lengths, already_dropfeat) -> ctx [B, Lp, H]
#   decode(words [B, Lw], ctx, ctx_mask, h [1,B,H], c [1,B,H]) -> (logit [B, Lw, V], h1, c1)
# ---------------------------------------------------------------------------------------------------------------
class SpeakerOracle:
    def __init__(self, P_enc, P_dec, bidirectional: bool, dtype=torch.float64):
        self.Pe = {k: v.detach().to(dtype).requires_grad_(v.is_floating_point()) for k, v in P_enc.items()}
        self.Pd = {k: v.detach().to(dtype).requires_grad_(v.is_floating_point()) for k, v in P_dec.items()}
        self.bidir, self.dtype = bidirectional, dtype

    def encode(self, can_feats, img_feats, lengths, already_dropfeat=False):
        return O.speaker_encoder(self.Pe, can_feats.to(self.dtype), img_feats.to(self.dtype), self.bidir)

    def decode(self, words, ctx, ctx_mask, h, c):
        return O.speaker_decoder(self.Pd, words, ctx, ctx_mask, h.to(self.dtype), c.to(self.dtype))

    def named_grads(self):
        out = {}
        for pre, P in (("encoder.", self.Pe), ("decoder.", self.Pd)):
            for k, v in P.items():
                if v.requires_grad and v.grad is not None:
                    out[pre + k] = v.grad
        return out


def speaker_teacher_forcing(encode, decode, can_feats, img_feats, lengths, insts, rnn_dim, pad=0):
    """speaker.py:235-290 -> (mean CE over the non-pad targets, un-reduced [B, Lw-1] losses, argmax predictions)."""
    B = can_feats.shape[0]
    ctx = encode(can_feats, img_feats, lengths, False)
    h = torch.zeros(1, B, rnn_dim, dtype=ctx.dtype)
    c = torch.zeros(1, B, rnn_dim, dtype=ctx.dtype)
    ctx_mask = O.length2mask(lengths, ctx.shape[1])
    logits, _, _ = decode(insts, ctx, ctx_mask, h, c)
    lg = logits.permute(0, 2, 1)[:, :, :-1]                      # (B, vocab, Lw-1), speaker.py:268-271
    tgt = insts[:, 1:]
    per_word = torch.nn.functional.cross_entropy(lg, tgt, ignore_index=pad, reduction="none")
    loss = per_word.sum() / (tgt != pad).sum()
    return loss, per_word, logits.argmax(dim=2)


def speaker_infer_batch(encode, decode, can_feats, img_feats, lengths, rnn_dim, max_decode, featdropmask=None, angle=ANG,
                        pad=0, unk=1, eos=2, bos=3, inject_words=None):
    """speaker.py:292-376, greedy (or `inject_words` [B, n] in place of sampled words) -> (words [B, n] numpy, the
    per-step masked logits).  The word fed back is the model's choice even for rows that already ended (the reference
    pads only the CPU copy, speaker.py:355-360)."""
    B = can_feats.shape[0]
    if featdropmask is not None:
        img_feats = img_feats.clone(); can_feats = can_feats.clone()
        img_feats[..., :-angle] *= featdropmask
        can_feats[..., :-angle] *= featdropmask
    ctx = encode(can_feats, img_feats, lengths, featdropmask is not None)
    ctx_mask = O.length2mask(lengths, ctx.shape[1])
    h = torch.zeros(1, B, rnn_dim, dtype=ctx.dtype)
    c = torch.zeros(1, B, rnn_dim, dtype=ctx.dtype)
    ended = np.zeros(B, dtype=bool)
    word = torch.full((B, 1), bos, dtype=torch.int64)
    words, step_logits = [], []
    for i in range(max_decode):
        logits, h, c = decode(word, ctx, ctx_mask, h, c)
        logits = logits.reshape(B, -1).clone()
        logits[:, unk] = -float("inf")
        step_logits.append(logits.detach())
        w = logits.max(1)[1] if inject_words is None else torch.as_tensor(inject_words[:, i])
        cpu_word = w.numpy().copy()
        cpu_word[ended] = pad
        words.append(cpu_word)
        word = w.view(-1, 1)
        ended = np.logical_or(ended, cpu_word == eos)
        if ended.all():
            break
    return np.stack(words, 1), torch.stack(step_logits, 1)


def shortest_path_features(env, device="cpu"):
    """speaker.py:191-226 `from_shortest_path` over the (fake) environment: follow the teacher from the start of every
    episode; per step the 36 view features and the feature of the candidate the teacher takes (zeros for STOP / ended).
    -> (can_feats [B, Lp, F], img_feats [B, Lp, 36, F], lengths [B])."""
    obs = env.reset(restart=False)
    B = len(obs)
    ended = np.zeros(B, bool)
    length = np.zeros(B, np.int64)
    img_feats, can_feats = [], []
    while not ended.all():
        img_feats.append(np.stack([ob["feature"] for ob in obs]).astype(np.float32))
        a = teacher_action(obs, ended)
        for i, act in enumerate(a):
            if act < 0 or act == len(obs[i]["candidates"]):
                a[i] = -1
        cf = np.zeros((B, obs[0]["feature"].shape[-1]), np.float32)
        for i, (ob, act) in enumerate(zip(obs, a)):
            if act != -1:
                cf[i] = ob["candidates"][act]["feature"]
        can_feats.append(cf)
        obs = env.step(a, obs, None)
        length += (1 - ended)
        ended[:] = np.logical_or(ended, a == -1)
    t = lambda x: torch.from_numpy(np.stack(x, 1)).contiguous().to(device)
    return t(can_feats), t(img_feats), length.tolist()


def back_translate_instructions(insts, pad=0, eos=2, bos=3):
    """envdrop.py:109-114: prepend <BOS>, close sentences that did not end with <EOS>."""
    insts = np.concatenate((np.full((insts.shape[0], 1), bos, np.int64), insts), 1)
    for inst in insts:
        if inst[-1] != pad:
            inst[-1] = eos
    return insts
