"""Rollout-level parity (SURVEY §8c last row, rows A9/A10): the reference's own `EnvDropAgent.rollout()` was driven
by `oracle/fake_env.py` (`oracle/make_goldens.py::gen_agent_tapes`); the same env + injected actions must give
the same IL loss, A2C loss, `total` count, chosen actions and parameter gradients
  * through the CPU oracle (pins oracle/rollout.py + the A2C / CE / marshalling restatements)   [CPU test]
  * through the HIP drop-in modules on the GPU                                                   [-m gpu test]
"""
import numpy as np
import pytest
import torch

from conftest import load_golden
from parity import check, grad_floor
from oracle import rollout as R
from oracle.fake_env import FakeR2REnv


def split(P):
    return ({k[4:]: v for k, v in P.items() if k.startswith("enc.")}, {k[4:]: v for k, v in P.items() if k.startswith("dec.")},
            {k[4:]: v for k, v in P.items() if k.startswith("cri.")})


def compare(res, grads, G, tol, gtol):
    assert np.array_equal(res["actions"], G["out"]["actions"].numpy())
    ml = float(torch.as_tensor(res["ml_loss"]).detach())
    assert abs(ml - float(G["out"]["ml_loss"])) <= tol * max(1.0, abs(float(G["out"]["ml_loss"])))
    if "rl_loss" in G["out"]:
        rl = float(torch.as_tensor(res["rl_loss"]).detach())
        assert abs(rl - float(G["out"]["rl_loss"])) <= tol * max(1.0, abs(float(G["out"]["rl_loss"])))
    if "progress_loss" in G["out"]:
        assert abs(float(res["progress_loss"]) - float(G["out"]["progress_loss"])) <= tol * max(1.0, abs(float(G["out"]["progress_loss"])))
    if "path_len" in G["out"]:          # trajectories as BaseAgent.test() records them (base.py:63-82)
        assert [len(t["path"]) for t in res["traj"]] == G["out"]["path_len"].numpy().tolist()
    if "total" in G["out"]:
        assert int(res["total"]) == int(G["out"]["total"])
    gmax = max([float(r.abs().max()) for r in G["grad"].values()] + [1e-30])
    for n, ref in G["gradnorm"].items():
        got = grads[n].detach().double().cpu().norm().item()
        assert abs(got - float(ref)) <= gtol * max(float(ref), 1e-3), f"grad norm {n}: {got} vs {float(ref)}"
    for n, ref in G["grad"].items():
        # ActionScoring's output bias receives sum_{b,c} d logit: zero in exact arithmetic for a CE loss (softmax rows sum to
        # their one-hot), so BOTH sides hold fp32 rounding noise of O(1e-7) x the O(1) summands (the reference's own value is
        # 8.9e-8): it is compared on the scale of the summands, not of the (vanishing) result
        zero_sum = n.endswith("decode_action.linear_out.bias")
        check(grads[n], ref, gtol, f"grad[{n}]", floor=max(1e-2 * gmax, 1e-1 if zero_sum else 0.0))


@pytest.mark.parametrize("mode", ["teacher", "sample"])
def test_oracle_rollout_matches_reference_agent(mode):
    G = load_golden("agent_envdrop_" + mode)
    be = R.OracleBackend(*split(G["param"]))
    env = FakeR2REnv(batch_size=4, max_len=8, vocab=40, seed=7)
    res = R.envdrop_rollout(be, env, mode, 6, inject_actions=G["out"]["actions"].numpy(), train_rl=(mode == "sample"))
    res["loss"].backward()
    compare(res, be.named_grads(), G, 1e-5, 1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["teacher", "sample"])
def test_hip_rollout_matches_reference_agent(mode):
    import vln_amd as vln
    vln._lib.load()
    dev = torch.device("cuda:0")
    G = load_golden("agent_envdrop_" + mode)
    Pe, Pd, Pc = split(G["param"])
    enc = vln.EncoderLSTM(40, 16, 32, 0, 0.5, True, 1)
    dec = vln.EnvDropDecoder(32, 0.5, 0.3, 8, 128, 2176)
    cri = vln.Critic(32, 0.5)
    enc.load_state_dict(Pe, strict=True); dec.load_state_dict(Pd, strict=True); cri.load_state_dict(Pc, strict=True)
    for m in (enc, dec, cri):
        m.to(dev).eval()
    # the sample tape also runs the A2C sweep as the fused HIP launch (losses.a2c_loss) instead of torch arithmetic
    be = R.ModuleBackend(enc, dec, cri, dev, a2c_loss=vln.losses.a2c_loss)
    env = FakeR2REnv(batch_size=4, max_len=8, vocab=40, seed=7)
    res = R.envdrop_rollout(be, env, mode, 6, inject_actions=G["out"]["actions"].numpy(), train_rl=(mode == "sample"))
    res["loss"].backward()
    compare(res, be.named_grads(), G, 1e-4, 1e-4)


# ---- the other two agents + the evaluation (argmax) path: tapes from gen_agent_tapes_more -------------------------------
# features are 64+128 wide there (make_goldens.py), everything else like the reference configs in miniature
def _env(mode):
    return FakeR2REnv(batch_size=5, max_len=8, vocab=40, seed=11 if mode == "teacher" else 13, img=64)


def _run(kind, be, mode):
    if kind == "follower":
        return R.follower_rollout(be, _env(mode), mode, 6)
    if kind == "monitor":
        return R.monitor_rollout(be, _env(mode), mode, 6, lamb=0.5)
    r = R.envdrop_rollout(be, _env(mode), mode, 6)
    r["ml_loss"] = r["ml_loss"]
    return r


CASES = [("follower", "teacher"), ("follower", "argmax"), ("monitor", "teacher"), ("monitor", "argmax"), ("envdrop", "argmax")]


@pytest.mark.parametrize("kind,mode", CASES)
def test_oracle_rollouts_match_reference_agents(kind, mode):
    G = load_golden(f"agent_{kind}_{mode}")
    Pe, Pd, Pc = split(G["param"])
    if kind == "follower":
        be = R.FollowerOracle(Pe, Pd, layers=2, bidirectional=True)
    elif kind == "monitor":
        be = R.MonitorOracle(Pe, Pd)
    else:
        be = R.OracleBackend(Pe, Pd, {})
    res = _run(kind, be, mode)                       # argmax: the rollout's OWN greedy actions must equal the tape's
    res["ml_loss"].backward()
    compare(res, be.named_grads(), G, 1e-5, 1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,mode", CASES)
def test_hip_rollouts_match_reference_agents(kind, mode):
    import vln_amd as vln
    vln._lib.load()
    dev = torch.device("cuda:0")
    G = load_golden(f"agent_{kind}_{mode}")
    Pe, Pd, _ = split(G["param"])
    F = 64 + 128
    if kind == "follower":
        enc = vln.EncoderLSTM(40, 16, 32, 0, 0.5, True, 2)
        dec = vln.AttnDecoderLSTM(32, 0.5, F, F)
    elif kind == "monitor":
        enc = vln.EncoderLSTM(40, 16, 32, 0, 0.5, False, 1)
        dec = vln.MonitorDecoder(32, 0.5, 8, [24], F, F)
    else:
        enc = vln.EncoderLSTM(40, 16, 32, 0, 0.5, True, 1)
        dec = vln.EnvDropDecoder(32, 0.5, 0.3, 8, 128, F)
    enc.load_state_dict(Pe, strict=True); dec.load_state_dict(Pd, strict=True)
    for m in (enc, dec):
        m.to(dev).eval()
    if kind == "follower":
        be = R.FollowerModules(enc, dec, dev)
    elif kind == "monitor":
        be = R.MonitorModules(enc, dec, dev)
    else:
        be = R.ModuleBackend(enc, dec, None, dev)
    res = _run(kind, be, mode)
    res["ml_loss"].backward()
    compare(res, be.named_grads(), G, 1e-4, 1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["teacher", "argmax"])
def test_hip_monitor_rollout_with_the_fused_step_loss(mode):
    """The reference Self-Monitor agent's tapes again, with the step loss of monitor.py:146-165 taken by
    losses.monitor_mixed_loss (one launch each way, progress target built on the device, no per-step copy of cur_prog_val
    to the host): same ml_loss, progress_loss, actions and gradients as the reference's own run."""
    import vln_amd as vln
    vln._lib.load()
    dev = torch.device("cuda:0")
    G = load_golden(f"agent_monitor_{mode}")
    Pe, Pd, _ = split(G["param"])
    F = 64 + 128
    enc = vln.EncoderLSTM(40, 16, 32, 0, 0.5, False, 1)
    dec = vln.MonitorDecoder(32, 0.5, 8, [24], F, F)
    enc.load_state_dict(Pe, strict=True); dec.load_state_dict(Pd, strict=True)
    for m in (enc, dec):
        m.to(dev).eval()
    be = R.MonitorModules(enc, dec, dev, mixed_loss=vln.losses.monitor_mixed_loss)
    res = _run("monitor", be, mode)
    res["ml_loss"].backward()
    compare(res, be.named_grads(), G, 1e-4, 1e-4)


# ---- back translation (envdrop.py:105-121,155-157 + speaker.py:292-376) ----------------------------------
# No tape of the reference exists for this branch: its hook calls attributes that do not exist (`decoder.drop_env`) and
# leaves the batch unsorted with stale lengths (oracle/rollout.py::envdrop_rollout).  The CPU oracle -- itself pinned
# module by module and loop by loop (tests/golden/speaker_*.npz) -- is the reference here.
def _speaker_modules(make_enc, make_dec):
    torch.manual_seed(91)
    enc, dec = make_enc(), make_dec()
    with torch.no_grad():                       # spread the random-init logits so that greedy decoding is not a tie-break
        dec.projection.weight *= 12.0
        dec.embedding.weight *= 4.0
        dec.projection.bias[2] += 1.0
    return enc, dec


def test_shortest_path_features():
    env = FakeR2REnv(batch_size=4, max_len=8, vocab=40, seed=7)
    can, img, lengths = R.shortest_path_features(env)
    assert lengths == [int(n) + 1 for n in env.n_moves]                    # n moves + the STOP step
    assert can.shape == (4, max(lengths), 2176) and img.shape == (4, max(lengths), 36, 2176)
    for i, n in enumerate(lengths):
        assert can[i, n - 1:].abs().sum() == 0                             # STOP and the steps after it: zero features
        assert can[i, :n - 1].abs().sum() > 0


@pytest.mark.gpu
def test_hip_back_translation_rollout():
    import vln_amd as vln
    vln._lib.load()
    dev = torch.device("cuda:0")
    G = load_golden("agent_envdrop_teacher")
    Pe, Pd, Pc = split(G["param"])
    enc = vln.EncoderLSTM(40, 16, 32, 0, 0.5, True, 1)
    dec = vln.EnvDropDecoder(32, 0.5, 0.3, 8, 128, 2176)
    enc.load_state_dict(Pe, strict=True); dec.load_state_dict(Pd, strict=True)
    enc.to(dev).eval()
    dec.to(dev).train()                         # the environment mask is a training-mode draw; the other dropouts off
    dec.drop_ratio = 0.0
    senc, sdec = _speaker_modules(lambda: vln.SpeakerEncoder(2176, 32, 0.5, True, 128, 0.3),
                                  lambda: vln.SpeakerDecoder(40, 16, 0, 32, 0.5))
    spk = vln.Speaker(senc.to(dev), sdec.to(dev), max_decode=7)
    env = FakeR2REnv(batch_size=4, max_len=8, vocab=40, seed=7)
    can, img, lengths = R.shortest_path_features(env)
    nlen = lambda ins: len({int(np.argmax(r == 0)) if (r == 0).any() else len(r) for r in ins})
    for _ in range(8):          # every call draws a new environment mask; take one under which the sentences differ in length
        insts, noise = vln.back_translate(spk, dec, can.to(dev), img.to(dev), lengths)      # (the re-sort below is then exercised)
        if nlen(insts) > 1:
            break
    assert set(noise.unique().cpu().tolist()) <= {0.0, float(torch.tensor(1 / 0.7, dtype=torch.float32))}
    # the oracle speaker, same weights, same mask -> the same instructions
    ora = R.SpeakerOracle({k.replace(".rnn.", "."): v.cpu() for k, v in senc.state_dict().items()},
                          {k.replace(".rnn.", "."): v.cpu() for k, v in sdec.state_dict().items()}, True)
    with torch.no_grad():
        words, _ = R.speaker_infer_batch(ora.encode, ora.decode, can, img, lengths, 32, 7, featdropmask=noise.cpu().double())
    assert np.array_equal(R.back_translate_instructions(words), insts), (words, insts)
    assert len({int(np.argmax(r == 0)) if (r == 0).any() else len(r) for r in insts}) > 1     # the re-sort is exercised
    # the follower trained on the generated instructions under the shared mask: HIP modules == CPU oracle
    be = R.ModuleBackend(enc, dec, None, dev)
    res = R.envdrop_rollout(be, FakeR2REnv(batch_size=4, max_len=8, vocab=40, seed=7), "teacher", 6, insts=insts, noise=noise)
    res["loss"].backward()
    bo = R.OracleBackend(Pe, Pd, Pc)
    ref = R.envdrop_rollout(bo, FakeR2REnv(batch_size=4, max_len=8, vocab=40, seed=7), "teacher", 6, insts=insts,
                            noise=noise.cpu().double())
    ref["loss"].backward()
    assert np.array_equal(res["actions"], ref["actions"])
    assert abs(float(res["ml_loss"].detach()) - float(ref["ml_loss"].detach())) <= 1e-4 * max(1.0, abs(float(ref["ml_loss"].detach())))
    g, go = be.named_grads(), bo.named_grads()
    gmax = max(float(r.abs().max()) for n, r in go.items() if not n.startswith("cri."))
    for n, r in go.items():
        if n.startswith("cri."):
            continue
        check(g[n], r, 1e-4, f"grad[{n}]", floor=grad_floor(n, gmax))


def test_oracle_rollout_instruction_override_is_consistent():
    """The back-translation plumbing of the harness (re-sort by recomputed length, actions mapped back to the environment's
    order): feeding the batch's OWN instructions in a shuffled-length order must give the loss of the plain rollout."""
    G = load_golden("agent_envdrop_teacher")
    env = FakeR2REnv(batch_size=4, max_len=8, vocab=40, seed=7)
    own = np.stack([b["instr_encoding"] for b in env.batch])
    be = R.OracleBackend(*split(G["param"]))
    res = R.envdrop_rollout(be, env, "teacher", 6, insts=own)
    assert abs(float(res["ml_loss"].detach()) - float(G["out"]["ml_loss"])) <= 1e-5 * max(1.0, abs(float(G["out"]["ml_loss"])))
    assert np.array_equal(res["actions"], G["out"]["actions"].numpy())
    # rows swapped so that the lengths are no longer sorted: same episodes, other instructions -> still runs, finite, and
    # the actions (teacher forcing) are the environment's own
    swapped = own[[3, 2, 1, 0]]
    res2 = R.envdrop_rollout(R.OracleBackend(*split(G["param"])), FakeR2REnv(batch_size=4, max_len=8, vocab=40, seed=7),
                             "teacher", 6, insts=swapped)
    assert np.isfinite(float(res2["ml_loss"].detach()))
    perm = np.argsort(-np.array([int(np.argmax(r == 0)) if (r == 0).any() else len(r) for r in swapped]), kind="stable")
    assert np.array_equal(res2["actions"], G["out"]["actions"].numpy()[:, perm])
