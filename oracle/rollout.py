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

    def decode(self, a, img, cand, h_tilde, h, c, ctx, mask):
        d = self.dtype
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

    def __init__(self, enc, dec, cri, device):
        self.enc, self.dec, self.cri, self.device = enc, dec, cri, device

    def encode(self, tokens, lengths):
        return self.enc(tokens, lengths)

    def decode(self, a, img, cand, h_tilde, h, c, ctx, mask):
        logit, (h1, c1), ht = self.dec(a, img, cand, h_tilde, h, c, ctx, mask, False)
        return logit, h1, c1, ht

    def critic(self, h):
        return self.cri(h)

    def named_grads(self):
        out = {}
        for pre, m in (("enc.", self.enc), ("dec.", self.dec), ("cri.", self.cri)):
            for k, p in m.named_parameters():
                out[pre + k] = p.grad if p.grad is not None else torch.zeros_like(p)
        return out


def envdrop_rollout(be, env, feedback: str, episode_len: int, inject_actions: Optional[np.ndarray] = None,
                    train_rl: bool = False, ml_weight=0.2, gamma=0.9):
    """Returns dict(ml_loss, rl_loss, total, loss, actions).  envdrop.py:86-278 with train_cl=False, speaker=None."""
    dev = be.device
    if feedback != "sample":
        train_rl = False
    obs = env.reset(restart=False)
    B = len(obs)
    tokens, seq_mask, lengths = marshal_instructions(obs, dev)
    ctx, h_t, c_t = be.encode(tokens, lengths)
    ended = np.zeros(B, bool)
    last_dist = np.array([ob["distance"] for ob in obs], np.float32)
    rewards, hidden, logps, masks, ents, acts = [], [], [], [], [], []
    ml = 0.0
    h_tilde = h_t
    for t in range(episode_len):
        a_in, img, cand, cl = marshal_step(obs, dev)
        logit, h_t, c_t, h_tilde = be.decode(a_in, img, cand, h_tilde, h_t, c_t, ctx, seq_mask)
        hidden.append(h_t)
        cmask = O.length2mask(cl).to(dev)
        logit = logit.masked_fill(cmask, -float("inf"))            # envdrop.py:173 (in place there)
        target = torch.from_numpy(teacher_action(obs, ended)).to(dev)
        ml = ml + O.masked_cross_entropy(logit, target, None, "sum")
        if feedback == "teacher":
            a_t = target
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
        obs = env.step(cpu_a, obs, None)
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
        rl, total = O.a2c_loss(logps, ents, vals, [torch.from_numpy(r).to(dev).to(dt) for r in rewards],
                               [torch.from_numpy(m).to(dev) for m in masks], last_v, torch.from_numpy(ended.copy()).to(dev),
                               gamma, "total")
    ml_loss = ml * ml_weight / B
    return dict(ml_loss=ml_loss, rl_loss=rl, total=total, loss=ml_loss + rl, actions=np.stack(acts))
