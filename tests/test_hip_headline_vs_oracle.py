"""GPU: the artefact `bench.py` TIMES -- `bench.GpuAgent`'s captured iteration: one hipGraph holding the prologue launch (batch
pull from pinned host memory + device-clock tick + weight-shadow refresh), the encoder with the rollout's feature gather and
the batch tail as passengers of its recurrence launch, seven chained decoder steps on the projected context, the rollout-wide
logits and cross-entropy, the backward with the decoder's weight gradients riding in the BPTT launch, per-module clip and
fused RMSprop -- against the CPU oracle ONE hop away: `oracle/torch_port.py` driven over the same episode batches with the
kernels' exported Philox masks, `torch.nn.utils.clip_grad_norm_(module, 40)` per module and `torch.optim.RMSprop(lr=1e-4)`
(reference loop: engine/trainer.py:411-427, agent/envdrop.py:86-278 teacher forcing).

Compared at BASELINE config 1's size (B 64, L 80, T 7, 36 x 2176 views, H 512): the loss of EVERY iteration (2 eager + 5
replays of the captured graph), every gradient of the first iteration, and the parameters after all K = 7 updates (as the
change from the initial parameters: RMSprop's first steps move every element by ~lr / sqrt(1 - alpha) whatever its gradient's
size, so elements whose gradient is noise carry no information -- they are compared in L2, the elements with a significant
gradient in max-abs).  fp32: north_star's 1e-4 on losses and gradients.  bf16: 1e-2 on losses and gradients (the unrounded
oracle: north_star's comparison); the parameter trajectory's achieved bounds are stated below."""
import pytest
import torch

from parity import check, FP32, BF16, grad_floor

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

# Bounds on the parameter change after the K = 7 updates = ~2-3x what round 5 measured:
#   fp32: L2 6.6e-5, over the 7.2 M elements with a significant first gradient 99th percentile 4.0e-5, max 1.7e-3;
#   bf16: L2 6.0e-2, 99th percentile 2.9e-2, max 0.38 -- an element whose LATER gradients change sign between the two arithmetics
#   moves the other way at full step size (RMSprop normalises the step), so the max over 10 M elements is not a bound one can
#   hold in bf16; the 99th percentile and the L2 error are.
TRAJ = {torch.float32: dict(l2=2e-4, p99=2e-4, max=4e-3), torch.bfloat16: dict(l2=0.12, p99=0.08)}


@pytest.fixture(scope="module")
def vln():
    import vln_amd
    vln_amd._lib.load()
    return vln_amd


def _mask(vln, n, seed, offset, p, shape):
    return vln.ops.dropout_mask(n, seed, offset, p, DEV).cpu().double().view(shape)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_headline_iteration_vs_oracle(vln, dtype):
    import bench
    from oracle import torch_port as O
    dev = torch.device(DEV)
    B, L, T, C = 64, 80, 7, 8
    H, E, AE, ANG, IMG, V = 512, 256, 64, 128, 2048, 36
    n_eager, n_replay = 2, 5
    lp = dtype != torch.float32
    torch.manual_seed(2020)
    store = bench.build_store(vln, dev, dtype, n_rows=512, seed=5)
    cpu_tapes = [bench.make_tape(B, L, T, C, seed=700 + k, n_rows=store.N) for k in range(4)]
    tapes = [bench.tape_to(t, dev, store=store) for t in cpu_tapes]
    live = bench.LiveBatch(tapes, source="pull")
    torch.manual_seed(2021)
    ag = bench.GpuAgent(vln, dev, dtype, 1, arena=True)          # defaults: rollout CE, deferred logits, chained steps
    ag.use_live(live)
    ag.ride_gather = True                                        # as bench.py main() sets them for --features store on one GPU
    ag.dec.ride_wgrads = lp
    ag.clear_grads_in_step = False                               # iteration 0 keeps its gradients for the comparison below
    clock = ag.use_clock(store)
    assert ag.dec.project_context and ag.dec.chain_steps and ag.dec.defer_logits and ag.use_prologue
    sd0 = {"enc": {k: v.detach().cpu().double().clone() for k, v in ag.enc.state_dict().items()},
           "dec": {k: v.detach().cpu().double().clone() for k, v in ag.dec.state_dict().items()}}
    table = store.table.detach().cpu().float()

    # ---- the HIP path: 2 eager iterations, capture, 5 replays ------------------------------------------------------------------------
    losses, hosts, grads0 = [], [], None
    for k in range(n_eager):
        loss = ag.iteration(live.load(k))
        torch.cuda.synchronize()
        losses.append(float(loss)); hosts.append(clock.host)
        if k == 0:
            grads0 = {key: {n: p.grad.detach().cpu().double().clone() for n, p in mod.named_parameters()}
                      for key, mod in (("enc", ag.enc), ("dec", ag.dec))}
            ag.clear_grads_in_step = True                        # from here on exactly the bench's configuration
    assert ag.dec.last_projected
    ag.capture(live.live)
    for k in range(n_eager, n_eager + n_replay):
        live.load(k)
        loss = ag.replay()
        torch.cuda.synchronize()
        losses.append(float(loss)); hosts.append(clock.host)
    vln._lib.check(vln._lib.load().vln_persistent_check(), "vln_persistent_check")
    assert hosts == [clock.STRIDE * (k + 1) for k in range(n_eager + n_replay)]
    final = {"enc": {k: v.detach().cpu().double() for k, v in ag.enc.state_dict().items()},
             "dec": {k: v.detach().cpu().double() for k, v in ag.dec.state_dict().items()}}

    # ---- the oracle: the same K iterations from the same initial parameters ---------------------------------------------------------
    P = {k: {n: v.clone().requires_grad_(True) for n, v in d.items()} for k, d in sd0.items()}
    params = [p for d in P.values() for p in d.values()]
    opt = torch.optim.RMSprop(params, lr=bench.LR)               # trainer.py:380-381: torch defaults (alpha 0.99, eps 1e-8)
    p, pf = 0.5, 0.3
    tol = FP32 if not lp else BF16
    def oracle_iteration(Pm, k, round_features):
        tape = cpu_tapes[k % len(cpu_tapes)]
        host = hosts[k]
        oe = host + 1
        cx, h, c = O.encoder_forward(Pm["enc"], tape["tokens"], tape["lengths"].tolist(), num_layers=1, bidirectional=True,
                                     emb_mask=_mask(vln, B * L * E, ag.enc.dropout_seed, oe * 8 + 0, p, (B, L, E)),
                                     ctx_mask_drop=_mask(vln, B * L * H, ag.enc.dropout_seed, oe * 8 + 1, p, (B, L, H)))
        ht, ml = h, 0.0
        for t, s in enumerate(tape["steps"]):
            f = bench.materialize_step(s, table, ANG)
            Ct = s["cand_mask"].shape[1]
            img = O.feature_dropout(f["img"].double(), _mask(vln, B * V * IMG, store.seed, host * 8 + 2 * t + 1, pf, (B, V, IMG)), ANG)
            cand = O.feature_dropout(f["cand"].double(), _mask(vln, B * Ct * IMG, store.seed, host * 8 + 2 * t + 2, pf, (B, Ct, IMG)), ANG)
            if round_features:
                img, cand = img.float().bfloat16().double(), cand.float().bfloat16().double()
            od = host + t + 1
            m = lambda site, n, shape: _mask(vln, n, ag.dec.dropout_seed, od * 8 + site, p, shape)
            drop = {"act": m(0, B * AE, (B, AE)), "hprev": m(1, B * H, (B, H)), "h1": m(2, B * H, (B, H)), "htilde": m(3, B * H, (B, H))}
            lo, (h, c), ht, _ = O.envdrop_step(Pm["dec"], s["angle"].double(), img, cand, ht, c, cx, tape["seq_mask"], drop=drop)
            ml = ml + O.masked_cross_entropy(lo.masked_fill(s["cand_mask"], -float("inf")), s["target"], None, "sum")
        return ml * bench.ML_WEIGHT / B

    if lp:
        # RECORDED, not asserted: iteration 0 against the oracle on the UN-rounded fp32 feature rows (the store's
        # table holds bf16 rows here, so only the feature DROPOUT's 1 / (1 - p) scaling is un-rounded: the table itself is the data)
        P32 = {k: {n: v.clone().requires_grad_(True) for n, v in d.items()} for k, d in sd0.items()}
        l32 = oracle_iteration(P32, 0, False)
        l32.backward()
        check(torch.tensor(losses[0]), l32.detach(), 1.0, "fp32 feature rows (recorded): loss of iteration 0")
        for key in ("enc", "dec"):
            gmax = max(float(q.grad.abs().max()) for q in P32[key].values() if q.grad is not None)
            for n, g in grads0[key].items():
                r = P32[key][n].grad if P32[key][n].grad is not None else torch.zeros_like(P32[key][n])
                check(g, r, 1.0, f"fp32 feature rows (recorded): iteration 0: grad[{key}.{n}]", floor=grad_floor(n, gmax))
    for k in range(n_eager + n_replay):
        tape = cpu_tapes[k % len(cpu_tapes)]
        host = hosts[k]
        opt.zero_grad()
        # encoder: its first call since the tick -> host counter value host + 1, sites 0 (embedding) / 1 (context)
        oe = host + 1
        cx, h, c = O.encoder_forward(P["enc"], tape["tokens"], tape["lengths"].tolist(), num_layers=1, bidirectional=True,
                                     emb_mask=_mask(vln, B * L * E, ag.enc.dropout_seed, oe * 8 + 0, p, (B, L, E)),
                                     ctx_mask_drop=_mask(vln, B * L * H, ag.enc.dropout_seed, oe * 8 + 1, p, (B, L, H)))
        ht, ml = h, 0.0
        for t, s in enumerate(tape["steps"]):
            f = bench.materialize_step(s, table, ANG)
            Ct = s["cand_mask"].shape[1]
            # the rollout's gather rode in the recurrence launch: the store's Philox stream, offsets word * 8 + (2 t + 1 | 2 t + 2)
            img = O.feature_dropout(f["img"].double(), _mask(vln, B * V * IMG, store.seed, host * 8 + 2 * t + 1, pf, (B, V, IMG)), ANG)
            cand = O.feature_dropout(f["cand"].double(), _mask(vln, B * Ct * IMG, store.seed, host * 8 + 2 * t + 2, pf, (B, Ct, IMG)), ANG)
            if lp:                                               # the streamed rows are bf16 DATA
                img, cand = img.float().bfloat16().double(), cand.float().bfloat16().double()
            od = host + t + 1
            m = lambda site, n, shape: _mask(vln, n, ag.dec.dropout_seed, od * 8 + site, p, shape)
            drop = {"act": m(0, B * AE, (B, AE)), "hprev": m(1, B * H, (B, H)), "h1": m(2, B * H, (B, H)), "htilde": m(3, B * H, (B, H))}
            lo, (h, c), ht, _ = O.envdrop_step(P["dec"], s["angle"].double(), img, cand, ht, c, cx, tape["seq_mask"], drop=drop)
            lo = lo.masked_fill(s["cand_mask"], -float("inf"))                                     # envdrop.py:173
            ml = ml + O.masked_cross_entropy(lo, s["target"], None, "sum")                         # envdrop.py:178-179
        oloss = ml * bench.ML_WEIGHT / B                                                           # envdrop.py:268
        oloss.backward()
        check(torch.tensor(losses[k]), oloss.detach(), tol, f"loss of iteration {k} ({'eager' if k < n_eager else 'replay'})")
        if k == 0:
            for key in ("enc", "dec"):
                gmax = max(float(q.grad.abs().max()) for q in P[key].values() if q.grad is not None)
                for n, g in grads0[key].items():
                    r = P[key][n].grad if P[key][n].grad is not None else torch.zeros_like(P[key][n])
                    check(g, r, tol, f"iteration 0: grad[{key}.{n}]", floor=grad_floor(n, gmax))
            sig = {key: {n: (q.grad.abs() >= 1e-2 * q.grad.abs().max()) if q.grad is not None else None for n, q in P[key].items()}
                   for key in P}
        torch.nn.utils.clip_grad_norm_(list(P["enc"].values()), bench.CLIP)                        # trainer.py:425-426
        torch.nn.utils.clip_grad_norm_(list(P["dec"].values()), bench.CLIP)
        opt.step()

    # ---- the trajectory: parameters after K updates, as the change from the start -----------------------------------------------------
    num = den = 0.0
    worst_sig, worst_name, errs = 0.0, "", []
    for key in ("enc", "dec"):
        for n, p0 in sd0[key].items():
            if n not in P[key] or not torch.is_floating_point(p0):
                continue
            d_ref = P[key][n].detach() - p0
            d_got = final[key][n] - p0
            num += float(((d_got - d_ref) ** 2).sum()); den += float((d_ref ** 2).sum())
            m = sig[key][n]
            if m is not None and bool(m.any()):
                e = (d_got - d_ref)[m].abs() / d_ref.abs().max().clamp_min(1e-30)
                errs.append(e.flatten())
                if float(e.max()) > worst_sig:
                    worst_sig, worst_name = float(e.max()), f"{key}.{n}"
    l2 = (num / max(den, 1e-300)) ** 0.5
    errs = torch.cat(errs).sort().values
    p99 = float(errs[int(0.99 * (errs.numel() - 1))])
    print(f"trajectory after {n_eager + n_replay} updates ({dtype}): L2 error of the parameter change {l2:.2e}; elements with a significant "
          f"first gradient ({errs.numel()}): 99th percentile {p99:.2e}, max {worst_sig:.2e} of the tensor's largest change ({worst_name})")
    check(torch.tensor(l2), torch.tensor(0.0), 1.0, f"trajectory L2 (recorded: {l2:.3e})", floor=1.0)
    check(torch.tensor(p99), torch.tensor(0.0), 1.0, f"trajectory p99 over significant elements (recorded: {p99:.3e})", floor=1.0)
    bound = TRAJ[dtype]
    assert l2 < bound["l2"], f"parameter change after K updates: L2 error {l2:.3e}"
    assert p99 < bound["p99"], f"parameter change, significant-gradient elements: 99th percentile {p99:.3e}"
    if "max" in bound:
        assert worst_sig < bound["max"], f"parameter change, significant-gradient elements: max {worst_sig:.3e} ({worst_name})"
