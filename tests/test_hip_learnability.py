"""GPU: a LEARNABILITY stand-in for north_star's SR / SPL clause.  The real clause -- SR/SPL on R2R val_unseen
after 80k iterations within 0.3 of the reference -- needs the Matterport simulator and the R2R data, which do not exist on the box.
What can be held here: on a synthetic navigation task that IS learnable (synthetic.GoalGridWorld: follow the direction the
instruction names until the goal landmark is in view, then STOP; teacher = shortest path), the HIP training iteration
(trainers.EnvDropILIteration: encoder, teacher-forced decoder steps, CE, backward, clip 40 per module, RMSprop lr 1e-4 -- the
reference's loop, engine/trainer.py:405-427 / 459-500) must

  * follow the CPU oracle's loss curve from the same initial parameters with the kernels' dropout masks: fp32 within 1e-3 relative
    at every one of the first 30 iterations (30 optimizer steps of drift, not one);
  * learn: greedy (argmax, eval mode) success rate and SPL of 64 held-out episodes scored by metrics.score_trajectories
    (Evaluation.score, engine/evaluator.py:101-146) go from ~0 to >= 0.95 in 150 iterations, in fp32 AND in the default bf16 mode;
  * bf16 tracks fp32: same masks, same initial parameters -- the loss averaged over the last 50 iterations within 25 %, success rate
    and SPL within 0.05."""
import pytest
import torch

from parity import check

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
B, L, T = 32, 12, 3
H, E, AE, ANG, IMG, V = 512, 256, 64, 128, 2048, 36


@pytest.fixture(scope="module")
def vln():
    import vln_amd
    vln_amd._lib.load()
    return vln_amd


def _mask(vln, n, seed, offset, p, shape):
    return vln.ops.dropout_mask(n, seed, offset, p, DEV).cpu().double().view(shape)


def _greedy(vln, ag, world, store, n_ep=64, seed=999, max_steps=5):
    """feedback="argmax" rollout (BaseAgent.test, agent/base.py:63-82) on the HIP modules in eval mode; trajectories scored like
    Evaluation.score."""
    dev = torch.device(DEV)
    eps, tokens, lens = world.episodes(n_ep, seed, L)
    ag.enc.eval(); ag.dec.eval()
    defer, chain = ag.dec.defer_logits, ag.dec.chain_steps
    ag.dec.defer_logits = ag.dec.chain_steps = False
    try:
        with torch.no_grad():
            tok = tokens.to(dev)
            ctx, h, c = ag.enc(tok, lens.to(dev, torch.int32))
            ht = h
            nodes = torch.tensor([e["start"] for e in eps])
            ended = torch.zeros(n_ep, dtype=torch.bool)
            paths = [[int(n)] for n in nodes]
            for _ in range(max_steps):
                st = {k: v.to(dev) for k, v in world.observe(nodes).items()}
                logit, (h, c), ht = ag.dec(st["angle"], None, None, ht, h, c, ctx, tok == 0,
                                           gather=(store, st["rows"], st["vidx"], st["crow"], st["cview"], st["chead"], st["celev"]))
                a = logit.argmax(1).cpu()
                nxt, ended_new = world.step(nodes, a, ended)
                for i in range(n_ep):
                    if not bool(ended[i]) and int(nxt[i]) != int(nodes[i]):
                        paths[i].append(int(nxt[i]))
                nodes, ended = nxt, ended_new
                if bool(ended.all()):
                    break
    finally:
        ag.enc.train(); ag.dec.train()
        ag.dec.defer_logits, ag.dec.chain_steps = defer, chain
    results, gt = [], {}
    for i, (e, p) in enumerate(zip(eps, paths)):
        iid = f"ep{i}"
        results.append({"instr_id": iid, "trajectory": [(f"v{n}", 0.0, 0.0) for n in p]})
        ref, n = [e["start"]], e["start"]
        while n != e["goal"]:
            n = world.neighbour(n, e["d"]); ref.append(n)
        gt[iid] = {"scan": "grid", "path": [f"v{n}" for n in ref]}
    dist = {"grid": vln.metrics.shortest_paths(world.edges())}
    summary, _ = vln.metrics.score_trajectories(results, gt, dist, error_margin=3.0)
    return summary


def _train(vln, world, dtype, iters, record=0):
    """-> (losses, greedy summaries before / after, recorded (tape, enc offset, first dec offset) of the first `record` iterations,
    initial state dicts)."""
    dev = torch.device(DEV)
    torch.manual_seed(4242)
    store = vln.DeviceFeatureStore(world.table, device=dev, dtype=dtype, angle_size=ANG)
    ag = vln.trainers.EnvDropILIteration(dev, dtype, 1, arena=True)
    ag.clear_grads_in_step = True
    sd0 = {"enc": {k: v.detach().cpu().double().clone() for k, v in ag.enc.state_dict().items()},
           "dec": {k: v.detach().cpu().double().clone() for k, v in ag.dec.state_dict().items()}}
    before = _greedy(vln, ag, world, store)
    losses, rec = [], []
    for it in range(iters):
        cpu_tape = world.tape(B, 1000 + it, L, T)
        tape = vln.synthetic.tape_to(cpu_tape, dev, store=store)
        if it < record:
            rec.append((cpu_tape, ag.enc._calls + 1, ag.dec._step_counter + 1))
        loss = ag.iteration(tape)
        losses.append(loss.detach().clone())              # (the arena hands the loss's buffer out again next iteration)
    torch.cuda.synchronize()
    vln._lib.check(vln._lib.load().vln_persistent_check(), "vln_persistent_check")
    losses = [float(x) for x in losses]
    after = _greedy(vln, ag, world, store)
    return losses, before, after, rec, sd0, (ag.enc.dropout_seed, ag.dec.dropout_seed)


@pytest.fixture(scope="module")
def runs(vln):
    world = vln.synthetic.GoalGridWorld(G=9, IMG=IMG, ANG=ANG, V=V, seed=1)
    out = {"world": world}
    out["fp32"] = _train(vln, world, torch.float32, 150, record=30)
    out["bf16"] = _train(vln, world, torch.bfloat16, 150)
    return out


def test_the_fp32_loss_curve_follows_the_oracle_for_30_updates(vln, runs):
    from oracle import torch_port as O
    world = runs["world"]
    losses, _, _, rec, sd0, (seed_e, seed_d) = runs["fp32"]
    P = {k: {n: v.clone().requires_grad_(True) for n, v in d.items()} for k, d in sd0.items()}
    opt = torch.optim.RMSprop([p for d in P.values() for p in d.values()], lr=vln.trainers.LR)       # trainer.py:380-381
    table = world.table.double()
    p, pf = 0.5, 0.3
    worst = 0.0
    for it, (tape, oe, od0) in enumerate(rec):
        opt.zero_grad()
        cx, h, c = O.encoder_forward(P["enc"], tape["tokens"], tape["lengths"].tolist(), num_layers=1, bidirectional=True,
                                     emb_mask=_mask(vln, B * L * E, seed_e, oe * 8 + 0, p, (B, L, E)),
                                     ctx_mask_drop=_mask(vln, B * L * H, seed_e, oe * 8 + 1, p, (B, L, H)))
        ht, ml = h, 0.0
        for t, s in enumerate(tape["steps"]):
            od = od0 + t
            m = lambda site, n, pp, shape: _mask(vln, n, seed_d, od * 8 + site, pp, shape)
            f = vln.synthetic.materialize_step(s, table, ANG)
            Ct = s["cand_mask"].shape[1]
            img = O.feature_dropout(f["img"].double(), m(4, B * V * IMG, pf, (B, V, IMG)), ANG)
            cand = O.feature_dropout(f["cand"].double(), m(5, B * Ct * IMG, pf, (B, Ct, IMG)), ANG)
            drop = {"act": m(0, B * AE, p, (B, AE)), "hprev": m(1, B * H, p, (B, H)), "h1": m(2, B * H, p, (B, H)),
                    "htilde": m(3, B * H, p, (B, H))}
            lo, (h, c), ht, _ = O.envdrop_step(P["dec"], s["angle"].double(), img, cand, ht, c, cx, tape["seq_mask"], drop=drop)
            ml = ml + O.masked_cross_entropy(lo.masked_fill(s["cand_mask"], -float("inf")), s["target"], None, "sum")
        oloss = ml * vln.trainers.ML_WEIGHT / B                                                   # envdrop.py:268
        oloss.backward()
        torch.nn.utils.clip_grad_norm_(list(P["enc"].values()), vln.trainers.CLIP)                # trainer.py:425-426
        torch.nn.utils.clip_grad_norm_(list(P["dec"].values()), vln.trainers.CLIP)
        opt.step()
        e = check(torch.tensor(losses[it]), oloss.detach(), 1e-3, f"learnability fp32: loss of iteration {it} vs the oracle after {it} updates")
        worst = max(worst, e)
    print(f"fp32 loss curve vs the oracle over {len(rec)} updates: worst relative error {worst:.2e}; loss {losses[0]:.4f} -> {losses[len(rec) - 1]:.4f}")


def test_both_precisions_learn_the_task_and_bf16_tracks_fp32(vln, runs):
    l32, b32, a32, *_ = runs["fp32"]
    l16, b16, a16, *_ = runs["bf16"]
    for name, before, after, losses in (("fp32", b32, a32, l32), ("bf16", b16, a16, l16)):
        print(f"{name}: success rate {before['success_rate']:.3f} -> {after['success_rate']:.3f}, SPL {before['spl']:.3f} -> {after['spl']:.3f}, "
              f"nDTW {after['ndtw']:.3f}, loss {losses[0]:.4f} -> mean of the last 50: {sum(losses[-50:]) / 50:.5f}")
        assert before["success_rate"] <= 0.5, (name, before)
        assert after["success_rate"] >= 0.95 and after["spl"] >= 0.95, (name, after)
        assert sum(losses[-50:]) / 50 < 0.05 * losses[0], (name, losses[0], losses[-50:])
    tail32, tail16 = sum(l32[-50:]) / 50, sum(l16[-50:]) / 50
    check(torch.tensor(tail16), torch.tensor(tail32), 0.25, "learnability: bf16 loss (mean of the last 50 iterations) vs fp32's")
    # early in training both modes see the same masks from the same parameters: the curves coincide to bf16 accuracy
    for it in (0, 1, 2, 5, 10):
        check(torch.tensor(l16[it]), torch.tensor(l32[it]), 5e-2, f"learnability: bf16 loss of iteration {it} vs fp32's")
    assert abs(a16["success_rate"] - a32["success_rate"]) <= 0.05 and abs(a16["spl"] - a32["spl"]) <= 0.05
