"""GPU: per-step input marshalling: the device-resident feature store must reproduce the
host-side construction of img_feature / cand_feature bit for bit (it is a gather + table lookup), its fused
feature dropout must use the same Philox stream the decoder would, and the pinned H2D ring must deliver the
bytes it was given."""
import numpy as np
import pytest
import torch

from parity import check

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def vln():
    import vln_amd
    vln_amd._lib.load()
    return vln_amd


def _problem(g, N=50, V=36, IMG=2048, B=16, C=7):
    table = torch.randn(N, V, IMG, generator=g).abs()
    rows = torch.randint(0, N, (B,), generator=g)
    vidx = torch.randint(0, 36, (B,), generator=g).int()
    crow = rows[:, None].repeat(1, C).clone()
    ncand = torch.randint(1, C, (B,), generator=g)                  # real candidates; slot ncand = STOP, rest padding
    for b in range(B):
        crow[b, ncand[b]:] = -1
    cview = torch.randint(0, 36, (B, C), generator=g).int()
    head = (torch.rand(B, C, generator=g) - 0.5) * 6
    elev = (torch.rand(B, C, generator=g) - 0.5)
    return table, rows, vidx, crow, cview, head, elev


@pytest.mark.parametrize("tdt", [torch.float32, torch.bfloat16])
def test_device_feature_store_matches_host_marshalling(vln, tdt):
    from oracle import torch_port as O
    g = torch.Generator().manual_seed(5)
    table, rows, vidx, crow, cview, head, elev = _problem(g)
    store = vln.DeviceFeatureStore(table, device=DEV, dtype=tdt)
    ref_table = table.to(tdt).float()
    ang = O.loc_embedding_table(128)
    img, _ = store.gather_pano(rows.to(DEV), vidx.to(DEV))
    assert torch.equal(img.cpu(), O.gather_pano(ref_table, rows, vidx.long(), ang))
    cand, lp, _ = store.gather_cands(crow.to(DEV), cview.to(DEV), head.to(DEV), elev.to(DEV), want_bf16=True)
    ref = O.gather_cands(ref_table, crow, cview.long(), head, elev)
    assert torch.equal(cand[..., :2048].cpu(), ref[..., :2048])
    assert (cand[..., 2048:].cpu() - ref[..., 2048:]).abs().max() < 1e-6       # device sinf/cosf vs libm
    assert torch.equal(lp, cand.bfloat16())
    for b in range(crow.shape[0]):                                            # STOP slot + padding are all-zero rows
        assert cand[b][crow[b] < 0].abs().sum().item() == 0.0


def test_store_feature_dropout_uses_the_philox_stream(vln):
    g = torch.Generator().manual_seed(6)
    table, rows, vidx, *_ = _problem(g, N=20, B=8)
    store = vln.DeviceFeatureStore(table, device=DEV)
    clean, _ = store.gather_pano(rows.to(DEV), vidx.to(DEV))
    dropped, (seed, off) = store.gather_pano(rows.to(DEV), vidx.to(DEV), p_feat=0.3)
    m = vln.ops.dropout_mask(8 * 36 * 2048, seed, off, 0.3, DEV).view(8, 36, 2048)
    assert torch.equal(dropped[..., :2048], clean[..., :2048] * m)
    assert torch.equal(dropped[..., 2048:], clean[..., 2048:])                # angle tail never dropped
    assert abs((m > 0).float().mean().item() - 0.7) < 0.01


@pytest.mark.parametrize("tdt", [torch.float32, torch.bfloat16])
def test_gather_step_is_both_gathers_in_one_launch(vln, tdt):
    """vln_gather_step == vln_gather_pano + vln_gather_cands with the same Philox positions, bit for bit (fp32 rows,
    bf16 rows, dropped elements, STOP / padding rows)."""
    g = torch.Generator().manual_seed(12)
    table, rows, vidx, crow, cview, head, elev = _problem(g)
    store = vln.DeviceFeatureStore(table, device=DEV, dtype=tdt)
    d = lambda t: t.to(DEV)
    a = store.gather_pano(d(rows), d(vidx), 0.3, want_bf16=True)
    b = store.gather_cands(d(crow), d(cview), d(head), d(elev), 0.3, want_bf16=True)
    store._calls -= 2
    (img, img_lp), (cand, cand_lp), (k1, k2) = store.gather_step(d(rows), d(vidx), d(crow), d(cview), d(head), d(elev), 0.3,
                                                                want_bf16=True)
    assert k1 == a[2] and k2 == b[2]
    assert torch.equal(img, a[0]) and torch.equal(img_lp, a[1]) and torch.equal(cand, b[0]) and torch.equal(cand_lp, b[1])
    store._calls -= 2
    (i2, i2lp), (c2, c2lp), _ = store.gather_step(d(rows), d(vidx), d(crow), d(cview), d(head), d(elev), 0.3, want_bf16=True,
                                                  want_f32=False)
    assert i2 is None and c2 is None and torch.equal(i2lp, a[1]) and torch.equal(c2lp, b[1])


def test_bf16_only_gather_feeds_the_decoder_like_fp32_rows_plus_copy(vln):
    """gather_*(want_f32=False): the bf16 rows alone, bit-identical to the bf16 copy of the full gather, and a bf16
    EnvDropDecoder given them as `img_feature` / `cand_feature` computes exactly what it computes from fp32 rows + copies."""
    g = torch.Generator().manual_seed(9)
    table, rows, vidx, crow, cview, head, elev = _problem(g, IMG=256)
    store = vln.DeviceFeatureStore(table, device=DEV, dtype=torch.bfloat16)
    d = lambda t: t.to(DEV)
    full = store.gather_pano(d(rows), d(vidx), 0.3, want_bf16=True)
    store._calls -= 1                                               # same Philox stream position for the second pass
    only = store.gather_pano(d(rows), d(vidx), 0.3, want_bf16=True, want_f32=False)
    assert only[0] is None and torch.equal(full[1], only[1])
    cfull = store.gather_cands(d(crow), d(cview), d(head), d(elev), 0.0, want_bf16=True)
    conly = store.gather_cands(d(crow), d(cview), d(head), d(elev), 0.0, want_bf16=True, want_f32=False)
    assert conly[0] is None and torch.equal(cfull[1], conly[1])
    torch.manual_seed(3)
    B, H, L = rows.shape[0], 64, 9
    dec = vln.EnvDropDecoder(H, 0.5, 0.3, 16, 128, 256 + 128, compute_dtype=torch.bfloat16).to(DEV).eval()
    a = torch.randn(B, 128, device=DEV); h = torch.randn(B, H, device=DEV); c = torch.randn(B, H, device=DEV)
    ctx = torch.randn(B, L, H, device=DEV)
    with torch.no_grad():
        o1 = dec(a, full[0].clone(), cfull[0].clone(), h, h, c, ctx, None, True, img_lp=full[1], cand_lp=cfull[1])
        o2 = dec(a, only[1], conly[1], h, h, c, ctx, None, True)
    for x, y in zip((o1[0], o1[1][0], o1[1][1], o1[2]), (o2[0], o2[1][0], o2[1][1], o2[2])):
        assert torch.equal(x, y)
    with pytest.raises(TypeError):
        dec(a, only[1], conly[1], h, h, c, ctx, None, False)        # bf16 rows that were not produced by the store pass


def test_fused_masked_ce_and_action_stats_golden(vln):
    """Row A9: the fused CE / Categorical kernel against the golden captured from torch's own ops (what the
    reference agents call inline)."""
    from conftest import load_golden
    G = load_golden("losses")
    I = {k: v.to(DEV) for k, v in G["inp"].items()}
    B = I["logits"].shape[0]
    w = torch.arange(1, B + 1, device=DEV).float()
    for red in ("none", "sum", "mean"):
        lg = I["logits"].clone().requires_grad_(True)
        ce = vln.losses.masked_cross_entropy(lg, I["target"], I["cand_mask"], red)
        assert torch.allclose(ce.cpu(), G["out"][f"ce_{red}"], rtol=1e-5, atol=1e-6), red
        (ce * (w if red == "none" else 1.0)).sum().backward()
        assert torch.allclose(lg.grad.cpu(), G["grad"][f"ce_{red}"], rtol=1e-5, atol=1e-6), red
    probs, lp, ent = vln.losses.action_stats(I["logits"], I["action"], I["cand_mask"])
    assert torch.allclose(lp.cpu(), G["out"]["log_prob"], rtol=1e-5, atol=1e-6)
    assert torch.allclose(ent.cpu(), G["out"]["entropy"], rtol=1e-5, atol=1e-6)
    assert probs[I["cand_mask"]].abs().max().item() == 0.0


@pytest.mark.parametrize("per_sample", [False, True])
@pytest.mark.parametrize("t", [0, 3])
def test_fused_monitor_step_loss_equals_reference_sequence(vln, t, per_sample):
    """Row A9 / monitor.py:146-165: losses.monitor_mixed_loss (ONE launch each way, the progress target built on the device)
    against the reference's sequence restated by the oracle -- CE(ignore_index) on the masked logits, prog_target from the
    distances in numpy with the `cur_dist <= 3` and `ended` rules, MSE, the t == 0 rule and the lambda mix -- value, d logits,
    d progress and the logged progress MSE; means and the curriculum (reduction="none") form; ignored rows, masked slots, a
    wide (> 16) row block."""
    import numpy as np
    from oracle import torch_port as O
    g = torch.Generator().manual_seed(11 + t)
    for B, C_ in ((37, 9), (128, 15), (5, 20)):
        n = torch.randint(1, C_ + 1, (B,), generator=g)
        mask = torch.arange(C_)[None, :] >= n[:, None]
        tgt = (torch.rand(B, generator=g) * n.float()).long()
        tgt[torch.rand(B, generator=g) < 0.2] = -1
        tgt[0] = 0
        logits = torch.randn(B, C_, generator=g) * 3
        prog = torch.tanh(torch.randn(B, generator=g))
        start = torch.rand(B, generator=g) * 15 + 4
        cur = start - torch.rand(B, generator=g) * start * 1.2
        cur = cur.clamp_min(0.2)
        ended = torch.rand(B, generator=g) < 0.3
        lam = 0.5 if B != 5 else 0.3
        w = torch.arange(1, B + 1).float() / B
        # reference sequence (monitor.py:155-158), numpy on the host
        pt = (start.numpy() - cur.numpy()) / start.numpy()
        pt[cur.numpy() <= 3.0] = 1.0
        pt[ended.numpy()] = prog.numpy()[ended.numpy()]
        lg0 = logits.clone().double().requires_grad_(True); pr0 = prog.clone().double().requires_grad_(True)
        ref = O.monitor_mixed_loss(lg0, tgt, mask, pr0, torch.from_numpy(pt).double(), t, lam, per_sample)
        ((ref * w.double()).sum() if per_sample else ref).backward()
        lg = logits.to(DEV).requires_grad_(True); pr = prog.to(DEV).requires_grad_(True)
        loss, pmse = vln.losses.monitor_mixed_loss(lg, tgt.to(DEV), mask.to(DEV), pr, start.to(DEV), cur.to(DEV), ended.to(DEV), t, lam,
                                                   per_sample)
        ((loss * w.to(DEV)).sum() if per_sample else loss).backward()
        assert loss.shape == ref.shape
        check(loss, ref.detach(), 1e-5, f"monitor loss B={B}")                 # fp32 kernel vs the fp64 restatement
        check(lg.grad, lg0.grad, 1e-5, f"d logits B={B}")
        if t > 0:
            check(pr.grad, pr0.grad, 1e-5, f"d progress B={B}")
            assert pr.grad[ended.to(DEV)].abs().sum().item() == 0.0            # ended episodes: target = prediction (detached)
            mse = float(((prog.double() - torch.from_numpy(pt).double()) ** 2).mean())
            assert abs(float(pmse) - mse) <= 1e-5 * max(1.0, mse)
        else:
            assert pr.grad is None or pr.grad.abs().max().item() == 0.0


def test_rollout_ce_equals_per_step_ce(vln):
    """losses.RolloutCE: the IL loss of a whole rollout in one launch == the sum of the per-step fused CE launches, value and
    every d logits bit for bit; steps with different candidate counts, ignored rows (-1), masked slots, a wide (>16) step;
    and against torch's CrossEntropyLoss on the masked logits."""
    from conftest import load_golden
    G = load_golden("losses")
    I = {k: v.to(DEV) for k, v in G["inp"].items()}
    g = torch.Generator().manual_seed(5)
    B = 37
    steps = []
    for C_ in (3, 8, 8, 5, 14, 20, 1):
        n = torch.randint(1, C_ + 1, (B,), generator=g)
        mask = torch.arange(C_)[None, :] >= n[:, None]
        tgt = (torch.rand(B, generator=g) * n.float()).long()
        tgt[torch.rand(B, generator=g) < 0.25] = -1
        steps.append((torch.randn(B, C_, generator=g) * 3, tgt, mask if C_ != 5 else None))
    res = []
    for mode in ("rollout", "per_step", "torch"):
        lgs = [s[0].to(DEV).clone().requires_grad_(True) for s in steps]
        if mode == "rollout":
            ce = vln.losses.RolloutCE()
            for lg, s in zip(lgs, steps):
                ce.add(lg, s[1].to(DEV), None if s[2] is None else s[2].to(DEV))
            total = ce.sum(scale=0.5) * 2.0                      # the in-launch scale, undone exactly (powers of two)
        elif mode == "per_step":
            total = sum(vln.losses.masked_cross_entropy(lg, s[1].to(DEV), None if s[2] is None else s[2].to(DEV), "sum") for lg, s in zip(lgs, steps))
        else:
            total = sum(torch.nn.functional.cross_entropy(lg if s[2] is None else lg.masked_fill(s[2].to(DEV), float("-inf")), s[1].to(DEV),
                                                          ignore_index=-1, reduction="sum") for lg, s in zip(lgs, steps))
        (total * 0.37).backward()
        res.append((total.detach().cpu(), [lg.grad.cpu() for lg in lgs]))
    assert abs(res[0][0].item() - res[1][0].item()) <= 1e-5 * abs(res[1][0].item())     # same rows, another summation order
    for a, b in zip(res[0][1], res[1][1]):
        assert torch.equal(a, b)
    assert torch.allclose(res[0][0], res[2][0], rtol=1e-5)
    for a, b in zip(res[0][1], res[2][1]):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-6)
    # the golden rows (captured from torch's ops on the reference's shapes) as a one-step rollout
    lg = I["logits"].clone().requires_grad_(True)
    ce = vln.losses.RolloutCE(); ce.add(lg, I["target"], I["cand_mask"])
    tot = ce.sum(); tot.backward()
    assert torch.allclose(tot.cpu(), G["out"]["ce_sum"], rtol=1e-5, atol=1e-6)
    assert torch.allclose(lg.grad.cpu(), G["grad"]["ce_sum"], rtol=1e-5, atol=1e-6)
    # per-episode form (SELF-PACE): [B] vector == sum over steps of the reduction="none" launches, gradients to rounding
    wrow = torch.rand(B, generator=g).to(DEV)
    pv = []
    for mode in ("rollout", "per_step"):
        lgs = [s[0].to(DEV).clone().requires_grad_(True) for s in steps]
        if mode == "rollout":
            ce = vln.losses.RolloutCE()
            for lg, s in zip(lgs, steps):
                ce.add(lg, s[1].to(DEV), None if s[2] is None else s[2].to(DEV))
            vec = ce.per_sample()
        else:
            vec = sum(vln.losses.masked_cross_entropy(lg, s[1].to(DEV), None if s[2] is None else s[2].to(DEV), "none") for lg, s in zip(lgs, steps))
        assert vec.shape == (B,)
        (vec * wrow).sum().backward()
        pv.append((vec.detach().cpu(), [lg.grad.cpu() for lg in lgs]))
    assert torch.allclose(pv[0][0], pv[1][0], rtol=1e-6, atol=1e-6)
    for a, b in zip(pv[0][1], pv[1][1]):      # the "none" launch is the wave-per-row kernel: same maths, other rounding
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-7)
    # more steps than one launch takes (VLN_CE_MAX_STEPS = 40)
    many = [torch.randn(4, 6, generator=g).to(DEV).requires_grad_(True) for _ in range(45)]
    tg = torch.randint(0, 6, (4,), generator=g).to(DEV)
    ce = vln.losses.RolloutCE()
    for lg in many:
        ce.add(lg, tg)
    tot = ce.sum(); tot.backward()
    ref = sum(torch.nn.functional.cross_entropy(lg.detach(), tg, reduction="sum") for lg in many)
    assert torch.allclose(tot, ref, rtol=1e-5) and all(lg.grad is not None for lg in many)


def test_fused_rmsprop_clip_matches_torch(vln):
    """Row N1: parameter trajectories of the fused clip+RMSprop step vs torch.optim.RMSprop + clip_grad_norm_ per
    group (trainer.py:423-427), 4 steps, one group clipped hard, one not at all; odd sizes exercise the tails."""
    torch.manual_seed(3)
    shapes = [[(37, 5), (11,), (64, 33)], [(7,), (129, 3), (2, 2)]]
    ref = [[torch.nn.Parameter(torch.randn(s, device=DEV)) for s in g] for g in shapes]
    mine = [[torch.nn.Parameter(p.detach().clone()) for p in g] for g in ref]
    opt_ref = torch.optim.RMSprop([p for g in ref for p in g], lr=1e-2)
    opt = vln.optim.FusedRMSprop(mine, lr=1e-2, clip_norm=2.0)
    for step in range(4):
        scale = [10.0, 0.01]
        opt.zero_grad(); opt_ref.zero_grad()
        for gi, (gr, gm) in enumerate(zip(ref, mine)):
            for pr, pm in zip(gr, gm):
                gval = torch.randn_like(pr) * scale[gi]
                pr.grad = gval.clone()
                pm.grad.add_(gval)                     # accumulate INTO the flat-bucket view, like autograd does
        norms = [torch.nn.utils.clip_grad_norm_(g, 2.0) for g in ref]
        opt_ref.step(); opt.step()
        assert torch.allclose(opt.norms, torch.stack(norms), rtol=1e-5)
        for gr, gm in zip(ref, mine):
            for pr, pm in zip(gr, gm):
                assert torch.allclose(pm, pr, rtol=2e-5, atol=1e-6), step
    v0 = mine[0][0]._version
    opt.step()
    assert mine[0][0]._version > v0                    # version-keyed weight shadows see the in-place update


@pytest.mark.parametrize("name", ["adam", "sgd"])
def test_fused_adam_sgd_match_torch(vln, name):
    """The other two entries of the reference's optim_switcher (trainer.py:17-21; Follower / Self-Monitor train with
    Adam, :65-67,219-222): parameter trajectories over 10 steps vs torch.optim with torch defaults, no clipping in
    one run and per-group clipping in the other."""
    for clip in (0.0, 1.5):
        torch.manual_seed(5)
        shapes = [[(37, 5), (11,), (64, 33)], [(7,), (129, 3), (2, 2)]]
        ref = [[torch.nn.Parameter(torch.randn(s, device=DEV)) for s in g] for g in shapes]
        mine = [[torch.nn.Parameter(p.detach().clone()) for p in g] for g in ref]
        flat_ref = [p for g in ref for p in g]
        opt_ref = torch.optim.Adam(flat_ref, lr=1e-3) if name == "adam" else torch.optim.SGD(flat_ref, lr=1e-2)
        opt = vln.optim.optim_switcher[name](mine, lr=1e-3 if name == "adam" else 1e-2, clip_norm=clip)
        for step in range(10):
            opt.zero_grad(); opt_ref.zero_grad()
            for gr, gm in zip(ref, mine):
                for pr, pm in zip(gr, gm):
                    gval = torch.randn_like(pr) * (3.0 if step % 2 else 0.3)
                    pr.grad = gval.clone()
                    pm.grad.add_(gval)
            if clip > 0:
                for g in ref:
                    torch.nn.utils.clip_grad_norm_(g, clip)
            opt_ref.step(); opt.step()
            for gr, gm in zip(ref, mine):
                for pr, pm in zip(gr, gm):
                    assert torch.allclose(pm, pr, rtol=2e-5, atol=2e-6), (name, clip, step)


def test_pinned_stager_roundtrip_and_reuse(vln):
    st = vln.PinnedStager(DEV, depth=2)
    rng = np.random.default_rng(0)
    for it in range(5):
        a = rng.standard_normal((64, 36, 2176)).astype(np.float32)
        t = rng.integers(0, 9, (64,)).astype(np.int64)
        d = st.put({"img": a, "target": t})
        s = d["img"].sum() + d["target"].sum()                                # consume on the compute stream
        st.release()
        assert torch.equal(d["img"].cpu(), torch.from_numpy(a)) and torch.equal(d["target"].cpu(), torch.from_numpy(t))
        assert torch.isfinite(s)
    assert st.slots[0]["host"]["img"].is_pinned()


def test_fused_rmsprop_per_group_clip_like_the_reference_trainer(vln):
    """trainer.py:380-381,423-427: ONE RMSprop over encoder + decoder + critic, but clip_grad_norm(40) on encoder and decoder
    only -- `clip_norm=[c, c, 0]`: the third group's (large) gradients must pass unclipped, the first two are clipped."""
    torch.manual_seed(9)
    shapes = [[(40, 8), (13,)], [(9, 9)], [(33, 3), (5,)]]
    ref = [[torch.nn.Parameter(torch.randn(s, device=DEV)) for s in g] for g in shapes]
    mine = [[torch.nn.Parameter(p.detach().clone()) for p in g] for g in ref]
    opt_ref = torch.optim.RMSprop([p for g in ref for p in g], lr=1e-2)
    opt = vln.optim.FusedRMSprop(mine, lr=1e-2, clip_norm=[1.5, 1.5, 0.0])
    for step in range(3):
        opt.zero_grad(); opt_ref.zero_grad()
        for gr, gm in zip(ref, mine):
            for pr, pm in zip(gr, gm):
                gval = torch.randn_like(pr) * 5.0
                pr.grad = gval.clone(); pm.grad.add_(gval)
        for g in ref[:2]:
            torch.nn.utils.clip_grad_norm_(g, 1.5)
        opt_ref.step(); opt.step()
        for gr, gm in zip(ref, mine):
            for pr, pm in zip(gr, gm):
                assert torch.allclose(pm, pr, rtol=2e-5, atol=1e-6), step
    with pytest.raises(ValueError):
        vln.optim.FusedRMSprop(mine, lr=1e-2, clip_norm=[1.0, 2.0])


def test_fused_step_can_clear_the_gradients_it_consumed(vln):
    """step(zero_grads=True): same parameter trajectory as a separate memset, the flat gradient buffer is zero afterwards and
    the next zero_grad() does not launch anything it does not need to."""
    torch.manual_seed(21)
    shapes = [[(40, 8), (13,)], [(9, 9), (5,)]]
    a = [[torch.nn.Parameter(torch.randn(s, device=DEV)) for s in g] for g in shapes]
    b = [[torch.nn.Parameter(p.detach().clone()) for p in g] for g in a]
    oa = vln.optim.FusedRMSprop(a, lr=1e-2, clip_norm=1.0)
    ob = vln.optim.FusedRMSprop(b, lr=1e-2, clip_norm=1.0)
    for step in range(3):
        oa.zero_grad(); ob.zero_grad()
        assert float(ob.flat_g.abs().max()) == 0.0
        for ga, gb in zip(a, b):
            for pa, pb in zip(ga, gb):
                gval = torch.randn_like(pa) * 3.0
                pa.grad.add_(gval); pb.grad.add_(gval)
        oa.step(); ob.step(zero_grads=True)
        assert float(ob.flat_g.abs().max()) == 0.0 and float(oa.flat_g.abs().max()) > 0.0
        for ga, gb in zip(a, b):
            for pa, pb in zip(ga, gb):
                assert torch.equal(pa, pb), step


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_rollout_gather_equals_per_step_gathers(vln, dtype):
    """DeviceFeatureStore.gather_rollout: every step of a teacher-forced rollout gathered in ONE launch == gather_step called
    for the steps in order (same Philox offsets): bit for bit, feature dropout on, both output precisions."""
    dev_ = torch.device(DEV)
    cpu_tape = vln.synthetic.make_tape(16, 24, 14, 6, seed=77)          # 14 steps: more than one chunk of the launch's argument block
    tape = vln.synthetic.tape_to(cpu_tape, dev_, store_dtype=dtype)
    lp = dtype != torch.float32
    res = []
    for rollout in (True, False):
        store = vln.DeviceFeatureStore(tape["store"].table, device=dev_, dtype=dtype)
        steps = [(s["rows"], s["vidx"], s["crow"], s["cview"], s["chead"], s["celev"]) for s in tape["steps"]]
        if rollout:
            out = store.gather_rollout(steps, 0.3, want_bf16=lp, want_f32=True)
        else:
            out = [store.gather_step(*st, 0.3, want_bf16=lp, want_f32=True)[:2] for st in steps]
        res.append(out)
    for (ia, ca), (ib, cb) in zip(*res):
        for x, y in zip(ia + ca, ib + cb):
            assert (x is None) == (y is None)
            if x is not None:
                assert torch.equal(x, y)
                assert x.float().abs().sum() > 0


def test_rollout_sampler_long_rollout_chunks(vln):
    """losses.RolloutSampler beyond one launch's argument block (VLN_CE_MAX_STEPS = 40 steps per vln_categorical_multi_bwd call):
    45 steps with differing candidate counts == sample_action per step, draws / log-probs / entropies bit for bit and the
    d logits of every step (plain tensors as logits: the materialised-gradient route)."""
    g = torch.Generator().manual_seed(3)
    B, T = 9, 45
    steps = []
    for t in range(T):
        C_ = 2 + (t % 13)
        n = torch.randint(1, C_ + 1, (B,), generator=g)
        steps.append((torch.randn(B, C_, generator=g) * 2, torch.arange(C_)[None, :] >= n[:, None]))
    wl, we = torch.randn(T, B, generator=g).to(DEV), torch.randn(T, B, generator=g).to(DEV)
    res = []
    for wide in (True, False):
        lgs = [lg.to(DEV).requires_grad_(True) for lg, _ in steps]
        if wide:
            s = vln.losses.RolloutSampler(seed=5, capacity=64)
            acts = [s.step(lg, m.to(DEV), offset=10 + t) for t, (lg, (_, m)) in enumerate(zip(lgs, steps))]
            lp, en = s.stats()
        else:
            outs = [vln.losses.sample_action(lg, m.to(DEV), seed=5, offset=10 + t) for t, (lg, (_, m)) in enumerate(zip(lgs, steps))]
            acts = [o[0] for o in outs]; lp = torch.stack([o[1] for o in outs]); en = torch.stack([o[2] for o in outs])
        ((lp * wl).sum() + (en * we).sum()).backward()
        res.append((torch.stack(acts), lp.detach().clone(), en.detach().clone(), [lg.grad.clone() for lg in lgs]))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])
    for a, b in zip(res[0][3], res[1][3]):
        assert torch.equal(a, b)
    with pytest.raises(ValueError):
        s = vln.losses.RolloutSampler(capacity=2)
        for t in range(3):
            s.step(steps[0][0].to(DEV))


@pytest.mark.parametrize("variant", ["pipelined", "both_outputs", "thirteen_steps", "per_step_recurrence", "batch128_no_idle_cus",
                                     "batch144_two_passes_with_passengers"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gather_riding_in_the_recurrence_launch_equals_the_rollout_gather(vln, dtype, variant):
    """EncoderLSTM.forward(ride=store.rollout_ride(...)): the rollout's feature rows gathered by PASSENGER workgroups of the
    persistent recurrence launch are the rows of gather_rollout bit for bit (same Philox offsets), and the encoder's own
    outputs are untouched by the passengers.  Variants: the software-pipelined passenger loop (one output precision); both
    precisions at once (the plain 8-rows-per-pass passenger loop); 13 steps (more than one argument block holds: the ride runs
    as its own launch in front of the recurrence); the per-step recurrence (no persistent launch to ride in: the same); B = 128
    (two directions x 16 unit slices x 8 row blocks = 256 workgroups fill every CU of an MI355X, no passenger fits:
    the library must notice BEFORE it commits to passengers and issue the gather as its own launch -- round 3 dropped it).
    Every variant also hands the ride four weight-shadow jobs (vln_gather_ride::shadow_jobs): refreshed by the passengers, or by
    their own launch where there are none, bit for bit what vln_shadow_refresh writes."""
    dev_ = torch.device(DEV)
    lib = vln._lib.load()
    torch.manual_seed(11)
    T = 13 if variant == "thirteen_steps" else 7
    # (B = 144, round 6: 9 row blocks -> the recurrence runs in two passes of 5 row blocks, 160 workgroups, and STILL carries passengers)
    cpu_tape = vln.synthetic.make_tape({"batch128_no_idle_cus": 128, "batch144_two_passes_with_passengers": 144}.get(variant, 64), 80, T, 8, seed=77)
    cpu_tape["table"] = cpu_tape["table"].bfloat16().float()
    tape = vln.synthetic.tape_to(cpu_tape, dev_, store_dtype=dtype)
    store = tape["store"]
    enc = vln.EncoderLSTM(992, 256, 512, 0, 0.5, True, 1, compute_dtype=dtype).to(dev_).train()
    steps = [(s["rows"], s["vidx"], s["crow"], s["cview"], s["chead"], s["celev"]) for s in tape["steps"]]
    lp = dtype != torch.float32
    want = dict(want_bf16=True, want_f32=True) if variant == "both_outputs" else dict(want_bf16=lp, want_f32=not lp)
    try:
        if variant == "per_step_recurrence":
            lib.vln_set_persistent(0)
        store._calls = 0; enc._calls = 0
        ref = store.gather_rollout(steps, 0.3, **want)
        ctx0, h0, c0 = enc(tape["tokens"], tape["lengths32"])
        store._calls = 0; enc._calls = 0
        ride = store.rollout_ride(steps, 0.3, **want)
        # ... and the ride carries weight-shadow jobs (ABI v17): an aligned matrix with both copies, a ragged one (scalar tile
        # path), a transposed-only fp32 copy and a two-source bias sum == vln_shadow_refresh on the same jobs
        gs = torch.Generator().manual_seed(5)
        srcs = [torch.randn(n, k, generator=gs).to(dev_) for n, k in ((192, 320), (70, 45), (128, 64), (1, 1024), (1, 1024))]
        def shadow_batch():
            sb = vln.ops.ShadowBatch()
            outs = [torch.full((192, 320), 7.0, dtype=torch.bfloat16, device=dev_), torch.full((320, 192), 7.0, dtype=torch.bfloat16, device=dev_),
                    torch.full((70, 45), 7.0, dtype=torch.bfloat16, device=dev_), torch.full((45, 70), 7.0, dtype=torch.bfloat16, device=dev_),
                    torch.full((64, 128), 7.0, device=dev_), torch.full((1, 1024), 7.0, device=dev_)]
            sb.add(srcs[0], outs[0], outs[1]); sb.add(srcs[1], outs[2], outs[3]); sb.add(srcs[2], None, outs[4])
            sb.add(srcs[3], outs[5], None, src2=srcs[4])
            if variant == "both_outputs":          # nine jobs: more than one argument block of the carrier holds -> their own launch
                for _ in range(5):
                    outs.append(torch.full((64, 128), 7.0, device=dev_))
                    sb.add(srcs[2], None, outs[-1])
            return sb, outs
        sb_ref, shadows_ref = shadow_batch()
        sb_ref.run()
        sb_ride, shadows_ride = shadow_batch()
        ride.carry_shadow_jobs(sb_ride.jobs, sb_ride.keep)
        ctx1, h1, c1 = enc(tape["tokens"], tape["lengths32"], ride=ride)
        torch.cuda.synchronize()
    finally:
        lib.vln_set_persistent(1)
    vln._lib.check(lib.vln_persistent_check(), "vln_persistent_check")
    assert torch.equal(ctx0, ctx1) and torch.equal(h0, h1) and torch.equal(c0, c1)
    for x, y in zip(shadows_ref, shadows_ride):
        assert torch.equal(x, y) and not bool((y.float() == 7.0).all())
    assert len(ride.outputs) == T
    for (a_img, a_cand), (b_img, b_cand) in zip(ref, ride.outputs):
        for x, y in list(zip(a_img, b_img)) + list(zip(a_cand, b_cand)):
            assert (x is None) == (y is None)
            if x is not None:
                assert torch.equal(x, y)


def test_out_of_range_gather_indices_give_zero_rows_and_a_sticky_error(vln):
    """a bad viewpoint row / view index must not become a silent out-of-bounds read of the 1.5 GB table.  The
    store registers its extent (vln_feature_table_extent); every gather -- stand-alone, per step, rollout-wide, riding in the
    recurrence launch -- zeroes the offending output row and raises the device's sticky word, which the next
    vln_persistent_check() reports once."""
    dev_ = torch.device(DEV)
    lib = vln._lib.load()
    vln._lib.check(lib.vln_persistent_check(), "clean start")
    g = torch.Generator().manual_seed(3)
    N, V, IMG, B, C = 40, 36, 2048, 16, 6
    table = torch.rand(N, V, IMG, generator=g)
    store = vln.DeviceFeatureStore(table, device=dev_, dtype=torch.float32)
    rows = torch.randint(0, N, (B,), generator=g).to(dev_)
    vidx = torch.randint(0, 36, (B,), generator=g, dtype=torch.int32).to(dev_)
    crows = torch.randint(0, N, (B, C), generator=g).to(dev_)
    crows[:, -1] = -1                                                   # legitimate empty slots: not an error
    cviews = torch.randint(0, V, (B, C), generator=g, dtype=torch.int32).to(dev_)
    head = torch.rand(B, C, generator=g).to(dev_); elev = torch.rand(B, C, generator=g).to(dev_)
    store.validate_indices(rows, vidx, crows, cviews)
    (img0, _), (cand0, _), _ = store.gather_step(rows, vidx, crows, cviews, head, elev)
    torch.cuda.synchronize()
    vln._lib.check(lib.vln_persistent_check(), "in-range indices")
    assert float(img0.abs().sum()) > 0 and float(cand0[:, -1].abs().sum()) == 0.0

    bad_rows = rows.clone(); bad_rows[3] = N + 5                        # past the table
    bad_vidx = vidx.clone(); bad_vidx[5] = 36                           # past the angle table
    bad_crows = crows.clone(); bad_crows[2, 1] = 10 ** 9
    bad_cviews = cviews.clone(); bad_cviews[7, 0] = -2
    with pytest.raises(ValueError, match="rows"):
        store.validate_indices(bad_rows, vidx, crows, cviews)
    with pytest.raises(ValueError, match="cand_views"):
        store.validate_indices(rows, vidx, crows, bad_cviews)

    def expect(img, cand):
        torch.cuda.synchronize()
        assert float(img[3].abs().sum()) == 0.0 and float(img[5].abs().sum()) == 0.0
        assert float(cand[2, 1].abs().sum()) == 0.0 and float(cand[7, 0].abs().sum()) == 0.0
        keep = torch.ones(B, dtype=torch.bool, device=dev_); keep[3] = keep[5] = False
        assert torch.equal(img[keep], img0[keep])
        ck = torch.ones(B, C, dtype=torch.bool, device=dev_); ck[2, 1] = ck[7, 0] = False
        assert torch.equal(cand[ck], cand0[ck])
        with pytest.raises(vln.VlnError, match="out of range"):
            vln._lib.check(lib.vln_persistent_check(), "bad indices")
        vln._lib.check(lib.vln_persistent_check(), "reported once")

    (img, _), (cand, _), _ = store.gather_step(bad_rows, bad_vidx, bad_crows, bad_cviews, head, elev)
    expect(img, cand)
    img, _ = store.gather_pano(bad_rows, bad_vidx)
    cand, _ = store.gather_cands(bad_crows, bad_cviews, head, elev)
    expect(img, cand)
    steps = [(bad_rows, bad_vidx, bad_crows, bad_cviews, head, elev)] * 2
    out = store.gather_rollout(steps)
    expect(out[1][0][0], out[1][1][0])
    # ... and as passengers of the recurrence launch (the pipelined loop: IMG 2048, ANG 128, one output precision)
    enc = vln.EncoderLSTM(50, 32, 512, 0, 0.0, True, 1).to(dev_).eval()
    tokens = torch.randint(1, 50, (64, 12), generator=g).to(dev_)
    lengths = torch.full((64,), 12, dtype=torch.int32, device=dev_)
    ride = store.rollout_ride(steps)
    with torch.no_grad():
        enc(tokens, lengths, ride=ride)
    expect(ride.outputs[0][0][0], ride.outputs[0][1][0])
