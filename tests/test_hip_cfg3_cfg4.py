"""GPU: BASELINE configs 3 and 4 -- ONE RANK's workload at its full size against the fp64 CPU oracle.

  cfg3  EnvDrop IL + A2C (trainer.py:412-416): a teacher-forced rollout (T = 7) whose IL loss is `ml_loss * ML_WEIGHT / B`
        (envdrop.py:173-179,268) PLUS a sampled rollout up to the reference's episode cap (MAX_EPISODE_LEN 35,
        configs/envdrop/envdrop_config.yaml:31) with the A2C loss of envdrop.py:222-264 (critic bootstrap from one more
        decoder step, gamma 0.9, RL_NORMALIZE total), B = 64 episodes, L = 80 tokens, H = 512, 36 x 2176 views, C <= 8.
  cfg4  the same iteration with the SELF-PACE curriculum's per-episode weights: both rollouts keep their losses as [B]
        vectors (train_cl, envdrop.py:70,178-179,251-253) and the batch loss is `dot(weight[idx], loss)` (curriculum.py:296),
        also in the weight-normalised form of curriculum.py:301.

Through the product path of the sampled rollout: `forward(gather=...)` from a resident feature table, `losses.RolloutCE`,
`losses.RolloutSampler` with the actions INJECTED (as the reference tapes do: sampled actions cannot be RNG-matched),
`Critic` over (steps x batch) rows, `losses.a2c_loss`.  Training mode, every dropout ON: the kernels' Philox masks are exported
(`vln_dropout_mask`) and injected into the oracle (encoder embedding / context, the six sites of every decoder step, the
critic), so the comparison is exact, not statistical.  Compared: both losses, `total`, the log-probs / entropies / values of
all 35 steps, and EVERY parameter gradient of encoder, decoder and critic.  fp32: 1e-4.  bf16: the oracle on the rounded
weights the kernels stream (1e-4 outputs, same_bf16_grad_tol() gradients) and the oracle on the unrounded masters (1e-2;
tensors that cannot meet it carry the measured bound below).
"""
import pytest
import torch

from parity import check, bf16_weights, bf16_round_st, same_bf16_grad_tol, grad_floor, fp32_streamed_keys, ENVDROP_FP32_KEYS, FP32, BF16, SAME_BF16

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ML_WEIGHT, GAMMA = 0.2, 0.9          # configs/envdrop/envdrop_config.yaml:45,42

# bf16 vs the UNROUNDED fp64 oracle (north_star's 1e-2): met by both losses, the log-probs / entropies / values of all 35
# steps and every encoder / decoder gradient (measured 2e-3 .. 7e-3, gpurun_out/parity_report.json).  The one exception is the
# critic's hidden layer: its input h_1 differs by ~5e-3 from the oracle's (the rounding of the streamed weights), which flips
# the ReLU of the units whose pre-activation is near zero -- each flip changes a gradient term by its full size (rounds 3-4
# measured 6e-2 max-abs, 3e-2 in L2 with the oracle deciding for itself).
CFG3_BF16_EXC = {}
# Round 5: the critic's ReLU decisions are SHARED with the oracle in the bf16 comparisons (Critic.relu_record -> O.critic(relu_on=)),
# like the dropout masks and the sampled actions: with them every critic gradient meets the bound of its variant.  What is asserted
# about the decisions themselves: they differ from the oracle's own ReLU for less than CRITIC_RELU_FLIPS[0] of the (row, unit)
# pairs, and only where the oracle's pre-activation is within CRITIC_RELU_FLIPS[1] of the mean |pre-activation| of zero.
# (Rounds 3-4 compared with the oracle's own decisions and carried the flipped units as a 0.12 exception on the hidden layer.)
CRITIC_RELU_FLIPS = (2e-3, 0.03)       # measured: 6.4e-4 of the units, 1.1e-2 of the mean (unrounded oracle); one unit of 1.2 M (same weights)
# bf16 on the SAME rounded weights: activations enter the MFMAs as hi + lo bf16 planes (2^-17 relative, against 2^-24 in fp32
# mode), and 35 recurrent steps compound that: the entropies of the last steps reach 1.2e-4 (single steps: <= 5e-5)
CFG3_SAME_EXC = {"entropy": 3e-4, "logp": 3e-4, "values": 3e-4}


def _tol(exc, tol, what):
    for k, v in (exc or {}).items():
        if what == k or what.startswith(k):
            return v
    return tol


@pytest.fixture(scope="module")
def vln():
    import vln_amd
    vln_amd._lib.load()
    return vln_amd


OWN_RELU = "bf16 unrounded, the oracle's own critic ReLU (recorded)"


def _mask(vln, n, seed, offset, p, shape):
    return vln.ops.dropout_mask(n, seed, offset, p, DEV).cpu().double().view(shape)


def _rl_tape(B, T, ncands, g):
    """Injected actions of the sampled rollout + what the environment answers (rewards, running masks): episode b runs for
    len_b steps, takes a random non-STOP candidate each step and STOP (the last real slot) at its last one."""
    lens = torch.randint(4, T + 1, (B,), generator=g)
    lens[0] = T
    lens[1] = T + 5                                    # an episode the cap cuts off: never ends, bootstraps from the critic
    acts, masks, rewards = [], [], []
    for t in range(T):
        nc = ncands[t]
        a = (torch.rand(B, generator=g) * (nc - 1).float()).long()
        a = torch.where(t == lens - 1, nc - 1, a)
        running = t < lens
        a = torch.where(running, a, torch.zeros_like(a))            # ended episodes: any valid index (their terms are masked)
        r = torch.where(t == lens - 1, torch.where(torch.rand(B, generator=g) < 0.5, 2.0, -2.0), torch.randn(B, generator=g).sign())
        acts.append(a); masks.append(running); rewards.append((r * running).float())
    return acts, masks, rewards, lens <= T


def _iteration(vln, cdt, mode, T_il=7, T_rl=35, B=64, L=80, N=768, normalised=False, only=None, captured=False, merged=False):
    """mode 'sum' = cfg3, 'self_pace' = cfg4.  `only`: run just the bf16 oracle of that name.  captured: the iteration (both
    rollouts, the critic, the losses, the backward) runs as ONE replayed hipGraph on a device clock (graphs.IterationGraph) -- what
    is compared with the oracle is then the REPLAY's output, one hop away, with the masks of the clock's offsets."""
    from oracle import torch_port as O
    H, E, AE, ANG, IMG, V = 512, 256, 64, 128, 2048, 36
    dev = torch.device(DEV)
    g = torch.Generator().manual_seed(3030)
    torch.manual_seed(3030)
    enc = vln.EncoderLSTM(992, E, H, 0, 0.5, True, 1, compute_dtype=cdt).to(dev).train()
    dec = vln.EnvDropDecoder(H, 0.5, 0.3, AE, ANG, IMG + ANG, compute_dtype=cdt).to(dev).train()
    cri = vln.Critic(H, 0.5).to(dev).train()
    table = (torch.randn(N, V, IMG, generator=g).abs() * 0.5).to(cdt)
    store = vln.DeviceFeatureStore(table, device=dev, dtype=cdt, angle_size=ANG)
    cpu_tape = vln.synthetic.make_tape(B, L, T_rl + 1, 8, seed=3031, n_rows=N)
    tape = vln.synthetic.tape_to(cpu_tape, dev, store=store)
    ncands = [(~s["cand_mask"]).sum(1) for s in cpu_tape["steps"]]
    acts, masks, rewards, ended = _rl_tape(B, T_rl, ncands, g)
    weight = (torch.rand(B, generator=g) * 0.99 + 0.01) if mode == "self_pace" else None      # uniform [0.01, 1]
    lp = cdt != torch.float32
    acts_d, rewards_d, masks_d, ended_d = [a.to(dev) for a in acts], [r.to(dev) for r in rewards], [m.to(dev) for m in masks], ended.to(dev)
    weight_d = None if weight is None else weight.to(dev)
    clock = vln.DeviceClock(dev).attach(enc, dec, cri) if captured else None

    # ---- the HIP path, recording the Philox offsets every module used ------------------------------------------------
    offs = {"enc": [], "dec": [], "cri": []}

    both = {}

    def gpu_encode(sample):
        """merged (round 6, trainers.EnvDropA2CIteration.merge_encoders): ONE encoder call over 2B rows -- the same instructions twice --
        whose halves feed the two rollouts, instead of two calls."""
        if not merged:
            offs["enc"].append(enc._calls + 1)
            return enc(tape["tokens"], tape["lengths32"])
        if not sample:
            offs["enc"].append(enc._calls + 1)
            ctx2, h2, c2 = enc(torch.cat((tape["tokens"], tape["tokens"]), 0), torch.cat((tape["lengths32"], tape["lengths32"]), 0))
            lp = getattr(ctx2, "_vln_lp", None)
            parts = [vln.trainers._SplitRows.apply(t, B) for t in (ctx2, h2, c2)]
            for k in range(2):
                if lp is not None:
                    parts[0][k]._vln_lp = lp[:B] if k == 0 else lp[B:]
            both["rl"] = tuple(p_[1] for p_ in parts)
            return tuple(p_[0] for p_ in parts)
        return both.pop("rl")

    def gpu_rollout(T, sample):
        ctx, h, c = gpu_encode(sample)
        ht = h
        hidden = []
        dec.defer_logits = not sample
        ce = vln.losses.RolloutCE()
        sampler = vln.losses.RolloutSampler()
        for t, s in enumerate(tape["steps"][:T]):
            offs["dec"].append(dec._step_counter + 1)
            logit, (h, c), ht = dec(s["angle"], None, None, ht, h, c, ctx, tape["seq_mask"],
                                    gather=(store, s["rows"], s["vidx"], s["crow"], s["cview"], s["chead"], s["celev"]))
            hidden.append(h)
            if sample:
                sampler.step(logit, s["cand_mask"], action=acts_d[t])
            else:
                ce.add(logit, s["target"], s["cand_mask"])
        if not sample:
            return dict(ml=ce.per_sample(scale=ML_WEIGHT / B) if weight is not None else ce.sum(scale=ML_WEIGHT / B))
        logps, ents = sampler.stats()
        sl = tape["steps"][T]                                               # envdrop.py:224-230: one more step for the bootstrap
        offs["dec"].append(dec._step_counter + 1)
        _, (last_h, _), _ = dec(sl["angle"], None, None, ht, h, c, ctx, tape["seq_mask"],
                                gather=(store, sl["rows"], sl["vidx"], sl["crow"], sl["cview"], sl["chead"], sl["celev"]))
        offs["cri"].append(cri._calls + 1)
        with torch.no_grad():
            last_v = cri(last_h).detach()
        offs["cri"].append(cri._calls + 1)
        vals = cri(torch.cat(hidden, 0)).view(T, B)
        rl, total = vln.losses.a2c_loss(logps, ents, vals, rewards_d, masks_d, last_v, ended_d, GAMMA, "total", per_sample=weight is not None)
        return dict(rl=rl, total=total, logp=logps, ent=ents, vals=vals)

    def batch_loss(ml, rl, w):
        if w is None:
            return ml + rl
        bl = torch.dot(w.to(ml.dtype), ml + rl)                              # curriculum.py:296
        return bl / w.sum().to(ml.dtype) if normalised else bl               # curriculum.py:301

    def hip_iteration():
        if clock is not None:
            clock.tick()
        for mod in (enc, dec, cri):
            for prm in mod.parameters():
                prm.grad = None
        cri.relu_record = []         # the ReLU decisions of the critic's two calls (bootstrap value, all steps' values)
        il = gpu_rollout(T_il, False)
        rlr = gpu_rollout(T_rl, True)
        relu = list(cri.relu_record)
        cri.relu_record = None
        loss = batch_loss(il["ml"], rlr["rl"], weight_d)
        loss.backward()
        return il, rlr, relu, loss

    if captured:
        graph = vln.IterationGraph(hip_iteration, clock).capture(warmup=2)
        graph.replay()                                   # (a first replay; the second one below is what is compared)
        il, rlr, relu, loss = graph.replay()
        torch.cuda.synchronize()
        vln._lib.check(vln._lib.load().vln_persistent_check(), "vln_persistent_check")
        host = clock.host            # call r of a module since the tick draws with the host-counter value host + r (runtime.DeviceClock)
        assert host == clock.STRIDE * 4    # two eager warm-up iterations + two replays
        offs = {"enc": [host + 1] if merged else [host + 1, host + 2], "dec": [host + 1 + i for i in range(T_il + T_rl + 1)],
                "cri": [(host + 1) * 8, (host + 2) * 8]}
    else:
        il, rlr, relu, loss = hip_iteration()
        torch.cuda.synchronize()
    relu_on = [m.cpu() for m in relu]

    # ---- the oracle(s) ---------------------------------------------------------------------------------------------------
    variants = [("fp32", FP32, False, None)] if not lp else \
        [("bf16 same-weights", SAME_BF16, True, dict(CFG3_SAME_EXC, **{"grad[": same_bf16_grad_tol()})),
         ("bf16 unrounded", BF16, False, CFG3_BF16_EXC),
         # RECORDED, not asserted: the same comparison against the oracle's OWN ReLU decisions in the critic -- the
         # reference's exact function -- so that a regression of that number (round 4: 3.5e-2 / 4.9e-2 on the critic's first layer) stays
         # visible next to the shared-decision assertion above; every check of this variant runs with tol = 1.0
         (OWN_RELU, 1.0, False, None)]
    if only is None:
        variants = [v for v in variants if v[0] != OWN_RELU]       # (the recorded own-ReLU variant runs where it is asked for by name)
    else:
        names = (only,) if isinstance(only, str) else tuple(only)
        variants = [v for v in variants if v[0] in names]
    sd = {"enc": enc.state_dict(), "dec": dec.state_dict(), "cri": cri.state_dict()}
    seq_mask = cpu_tape["seq_mask"]
    lengths = cpu_tape["lengths"].tolist()
    p, pf = 0.5, 0.3
    flips = []
    for name, tol, same, exc in variants:
        P = {k: {n: v.detach().cpu().double().requires_grad_(True) for n, v in d.items()} for k, d in sd.items()}
        Pe = bf16_weights(P["enc"], skip=("embedding.weight",)) if same else P["enc"]       # embedding rows are gathered in fp32
        # the 128 -> 64 embedding runs in fp32, and so do the matrices the module streams in fp32 (EnvDropDecoder.fp32_weights)
        Pd = bf16_weights(P["dec"], skip=("act_embed.0.weight",) + fp32_streamed_keys(dec, ENVDROP_FP32_KEYS)) if same else P["dec"]
        Pc = P["cri"]                                                                       # the critic computes in fp32
        it = {k: iter(v) for k, v in offs.items()}

        obo = {}

        def ora_encode(sample):
            if not merged:
                oe = next(it["enc"])
                return O.encoder_forward(Pe, cpu_tape["tokens"], lengths, num_layers=1, bidirectional=True,
                                         emb_mask=_mask(vln, B * L * E, enc.dropout_seed, oe * 8 + 0, p, (B, L, E)),
                                         ctx_mask_drop=_mask(vln, B * L * H, enc.dropout_seed, oe * 8 + 1, p, (B, L, H)))
            if not sample:       # one call over 2B rows: row r of the call draws the Philox positions of row r
                oe = next(it["enc"])
                cx2, h2, c2 = O.encoder_forward(Pe, torch.cat((cpu_tape["tokens"], cpu_tape["tokens"]), 0), lengths + lengths, num_layers=1,
                                                bidirectional=True,
                                                emb_mask=_mask(vln, 2 * B * L * E, enc.dropout_seed, oe * 8 + 0, p, (2 * B, L, E)),
                                                ctx_mask_drop=_mask(vln, 2 * B * L * H, enc.dropout_seed, oe * 8 + 1, p, (2 * B, L, H)))
                obo["rl"] = (cx2[B:], h2[B:], c2[B:])
                return cx2[:B], h2[:B], c2[:B]
            return obo.pop("rl")

        def ora_rollout(T, sample):
            cx, h, c = ora_encode(sample)
            cxs = bf16_round_st(cx) if same else cx              # the text attention streams a bf16 copy of the context
            sc = cx if (same and dec.last_projected) else None    # ... and scores on K = ctx W_in formed from the fp32 context (round 5)
            ht = h
            hidden, logps, ents, ml = [], [], [], 0.0

            def step(s, ht, c):
                od = next(it["dec"])
                m = lambda site, n, pp, shape: _mask(vln, n, dec.dropout_seed, od * 8 + site, pp, shape)
                f = vln.synthetic.materialize_step(s, table.float(), ANG)
                Ct = s["cand_mask"].shape[1]
                img = O.feature_dropout(f["img"].double(), m(4, B * V * IMG, pf, (B, V, IMG)), ANG)
                cand = O.feature_dropout(f["cand"].double(), m(5, B * Ct * IMG, pf, (B, Ct, IMG)), ANG)
                if lp:                                           # the features are DATA: the kernels stream bf16 rows
                    img, cand = img.float().bfloat16().double(), cand.float().bfloat16().double()
                drop = {"act": m(0, B * AE, p, (B, AE)), "hprev": m(1, B * H, p, (B, H)), "h1": m(2, B * H, p, (B, H)),
                        "htilde": m(3, B * H, p, (B, H))}
                lo, (h1, c1), ht1, _ = O.envdrop_step(Pd, s["angle"].double(), img, cand, ht, c, cxs, seq_mask, drop=drop, score_ctx=sc)
                return lo, h1, c1, ht1

            for t, s in enumerate(cpu_tape["steps"][:T]):
                lo, h, c, ht = step(s, ht, c)
                hidden.append(h)
                lo = lo.masked_fill(s["cand_mask"], -float("inf"))                           # envdrop.py:173
                if sample:
                    l_, e_ = O.categorical_logprob_entropy(lo, acts[t])
                    logps.append(l_); ents.append(e_)
                else:
                    ml = ml + O.masked_cross_entropy(lo, s["target"], None, "none" if weight is not None else "sum")
            if not sample:
                return dict(ml=ml * ML_WEIGHT / B)
            _, last_h, _, _ = step(cpu_tape["steps"][T], ht, c)
            # bf16: the critic's ReLU DECISIONS are the kernels' (CRITIC_RELU above); fp32: the reference's own ReLU
            on = relu_on if (lp and name != OWN_RELU) else (None, None)
            pre = []
            oc = next(it["cri"])
            with torch.no_grad():
                last_v = O.critic(Pc, last_h, _mask(vln, B * H, cri.dropout_seed, oc, p, (B, H)), relu_on=on[0], pre_out=pre)
            oc = next(it["cri"])
            vals = O.critic(Pc, torch.cat(hidden, 0), _mask(vln, T * B * H, cri.dropout_seed, oc, p, (T * B, H)), relu_on=on[1],
                            pre_out=pre).view(T, B)
            if lp and name != OWN_RELU:       # where the decisions differ from the oracle's own: few units, all with a pre-activation next to zero
                for z, m in zip(pre, relu_on):
                    flip = (z > 0) != m
                    frac, worst = float(flip.double().mean()), float(z[flip].abs().max()) if bool(flip.any()) else 0.0
                    scale = float(z.abs().mean())
                    flips.append((name, frac, worst / scale))
                    assert frac < CRITIC_RELU_FLIPS[0] and worst < CRITIC_RELU_FLIPS[1] * scale, (name, frac, worst, scale)
            rl, total = O.a2c_loss(logps, ents, list(vals.unbind(0)), [r.double() for r in rewards], masks, last_v, ended, GAMMA,
                                   "total", per_sample=weight is not None)
            return dict(rl=rl, total=total, logp=torch.stack(logps), ent=torch.stack(ents), vals=vals)

        oil = ora_rollout(T_il, False)
        orl = ora_rollout(T_rl, True)
        oloss = batch_loss(oil["ml"], orl["rl"], None if weight is None else weight.double())
        oloss.backward()
        t = lambda what: _tol(exc, tol, what)
        assert int(round(float(rlr["total"]))) == int(round(float(orl["total"]))) == int(sum(int(m.sum()) for m in masks))
        check(il["ml"], oil["ml"], t("ml_loss"), f"{name}: ml_loss")
        check(rlr["rl"], orl["rl"], t("rl_loss"), f"{name}: rl_loss")
        check(loss, oloss, t("loss"), f"{name}: loss")
        check(rlr["logp"], orl["logp"], t("logp"), f"{name}: logp [T,B]")
        check(rlr["ent"], orl["ent"], t("entropy"), f"{name}: entropy [T,B]")
        check(rlr["vals"], orl["vals"], t("values"), f"{name}: values [T,B]")
        for key, mod in (("enc", enc), ("dec", dec), ("cri", cri)):
            refs = {n: P[key][n].grad for n, _ in mod.named_parameters()}
            gmax = max(float(r.abs().max()) for r in refs.values() if r is not None)
            for n, prm in mod.named_parameters():
                r = refs[n] if refs[n] is not None else torch.zeros_like(P[key][n])
                got = prm.grad if prm.grad is not None else torch.zeros_like(prm)
                check(got, r, t(f"grad[{key}.{n}]"), f"{name}: grad[{key}.{n}]", floor=grad_floor(n, gmax))
    for name, frac, worst in flips:
        print(f"critic ReLU decisions that differ from the oracle's own ({name}): {frac:.2e} of the units, |pre-activation| <= {worst:.2e} of the mean")


@pytest.mark.parametrize("cdt", [torch.float32, torch.bfloat16])
def test_cfg3_il_plus_a2c_full_size(vln, cdt):
    _iteration(vln, cdt, "sum")


@pytest.mark.parametrize("cdt", [torch.float32, torch.bfloat16])
def test_cfg3_captured_iteration_vs_oracle(vln, cdt):
    """the cfg3 iteration as ONE replayed hipGraph (device clock, no host code between the 50 decoder steps)
    against the oracle one hop away -- every loss, log-prob, entropy, value and gradient of the second replay."""
    _iteration(vln, cdt, "sum", captured=True, only=None if cdt == torch.float32 else "bf16 unrounded")


@pytest.mark.parametrize("cdt", [torch.float32, torch.bfloat16])
def test_cfg4_self_pace_weighted_full_size(vln, cdt):
    # bf16: north_star's comparison (the unrounded oracle); the kernels' own arithmetic is pinned by cfg3's same-weights run --
    # the two configs differ in the loss weighting only
    _iteration(vln, cdt, "self_pace", only=None if cdt == torch.float32 else "bf16 unrounded")


@pytest.mark.parametrize("cdt", [torch.float32, torch.bfloat16])
def test_cfg3_with_one_encoder_call_for_both_rollouts(vln, cdt):
    """Round 6: the iteration's two rollouts encode the same instructions (trainer.py:413-416) -- ONE encoder call over 2B rows whose
    halves feed them (what trainers.EnvDropA2CIteration does by default), against the oracle run the same way: every loss, log-prob,
    entropy, value and gradient, at BASELINE config 3's per-rank size with a shorter sampled rollout."""
    # bf16: also RECORDED (tol 1.0) against the oracle's OWN ReLU decisions in the critic
    _iteration(vln, cdt, "sum", T_rl=12, merged=True, only=None if cdt == torch.float32 else ("bf16 unrounded", OWN_RELU))


def test_cfg4_self_pace_weight_normalised_form(vln):
    """curriculum.py:301 (`dot(w, loss) / w.sum()`, the form the other agents' per-episode losses take) on the same per-episode
    vector, with a shorter sampled rollout."""
    _iteration(vln, torch.float32, "self_pace", T_rl=12, normalised=True)
