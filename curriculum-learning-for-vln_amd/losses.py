"""Loss / action-selection stage of the rollouts on the HIP path.

The reference does this inline with torch ops every decoder step (follower.py:123-139, envdrop.py:173-195,
monitor.py:146-176): `logits.masked_fill_(candidate_mask, -inf)`, `CrossEntropyLoss(ignore_index=-1)`,
`softmax` -> `Categorical.log_prob / entropy`.  The drop-in modules leave that code untouched (it runs on
PyTorch-ROCm); these functions are the fused alternative: ONE launch forward, ONE backward.
`a2c_loss` is the A2C sweep (envdrop.py:235-264) as one launch forward, one backward.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib, ops

_p = ops._p


def _not_deferred(logits, who):
    ops.check_live(logits, who)
    return _not_deferred_(logits, who)


def _not_deferred_(logits, who):
    """EnvDropDecoder.defer_logits leaves the logits unwritten until losses.RolloutCE forms them: anything of this module that
    would read them earlier says so instead of computing on uninitialised memory."""
    rec = getattr(logits, "_vln_rec", None)
    if rec is not None and rec.io is not None and rec.io.defer_logits and not rec.flushed:
        raise _lib.VlnError(f"{who}: these logits come from a decoder with defer_logits=True and have not been formed yet; "
                            "use losses.RolloutCE for the rollout's loss, or defer_logits=False when the logits are read per step")


def _mask8(cand_mask):
    if cand_mask is None:
        return None
    m = cand_mask.contiguous()
    return m.view(torch.uint8) if m.dtype == torch.bool else m.to(torch.uint8)


class _MaskedCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, cand_mask, ignore_index, fused_sum):
        B, C = logits.shape
        dev = logits.device
        lg = logits.detach()
        if not lg.is_contiguous():
            lg = lg.contiguous()
        probs = ops.empty(B, C, dtype=torch.float32, device=dev)
        m8 = _mask8(cand_mask)
        tgt = target if target.is_contiguous() else target.contiguous()
        if fused_sum:        # reduction="sum" inside the same launch: a 0-dim result, no [B] vector round trip
            out = ops.empty((), dtype=torch.float32, device=dev)
            loss_p, sum_p = None, out.data_ptr()
        else:
            out = ops.empty(B, dtype=torch.float32, device=dev)
            loss_p, sum_p = out.data_ptr(), None
        st = _lib.load().vln_masked_ce_fwd(lg.data_ptr(), lg.stride(0), tgt.data_ptr(), _p(m8), loss_p, sum_p, probs.data_ptr(),
                                           None, None, None, B, C, ignore_index, 0, _lib.raw_stream())
        if st:
            _lib.check(st, "vln_masked_ce_fwd")
        ctx.save_for_backward(probs, tgt)
        ctx.ignore_index, ctx.fused_sum = ignore_index, fused_sum
        return out

    @staticmethod
    def backward(ctx, dloss):
        probs, tgt = ctx.saved_tensors
        B, C = probs.shape
        dl = ops.empty_like(probs)
        if ctx.fused_sum or dloss.stride(0) == 0:
            stride = 0                                  # one upstream scalar (also what .sum().backward() hands down)
        else:
            stride = 1
            if not dloss.is_contiguous():
                dloss = dloss.contiguous()
        st = _lib.load().vln_masked_ce_bwd(probs.data_ptr(), tgt.data_ptr(), dloss.data_ptr(), stride, dl.data_ptr(), B, C,
                                           ctx.ignore_index, _lib.raw_stream())
        if st:
            _lib.check(st, "vln_masked_ce_bwd")
        return dl, None, None, None, None


class _MaskedCEMean(torch.autograd.Function):
    """reduction="mean": the count of rows with a target and the division inside the loss launch (five torch launches per call
    before: ne, sum, cast, div and the div's backward)."""

    @staticmethod
    def forward(ctx, logits, target, cand_mask, ignore_index):
        B, C = logits.shape
        dev = logits.device
        lg = logits.detach()
        if not lg.is_contiguous():
            lg = lg.contiguous()
        probs = ops.empty(B, C, dtype=torch.float32, device=dev)
        out = ops.empty(2, dtype=torch.float32, device=dev)           # [mean, 1 / count]
        tgt = target if target.is_contiguous() else target.contiguous()
        rows = ops.empty(B, dtype=torch.float32, device=dev) if B * C > 16384 else None      # (VLN_CE_MEAN_ONE_LAUNCH_MAX: see the header)
        st = _lib.load().vln_masked_ce_mean_fwd(lg.data_ptr(), lg.stride(0), tgt.data_ptr(), _p(_mask8(cand_mask)), out.data_ptr(),
                                                probs.data_ptr(), B, C, ignore_index, _p(rows), _lib.raw_stream())
        if st:
            _lib.check(st, "vln_masked_ce_mean_fwd")
        ctx.save_for_backward(probs, tgt, out)
        ctx.ignore_index = ignore_index
        return out[0]

    @staticmethod
    def backward(ctx, dloss):
        probs, tgt, out = ctx.saved_tensors
        B, C = probs.shape
        dl = ops.empty_like(probs)
        dloss = dloss.contiguous()
        st = _lib.load().vln_masked_ce_mean_bwd(probs.data_ptr(), tgt.data_ptr(), dloss.data_ptr(), out.data_ptr() + 4, dl.data_ptr(), B, C,
                                                ctx.ignore_index, _lib.raw_stream())
        if st:
            _lib.check(st, "vln_masked_ce_mean_bwd")
        return dl, None, None, None


def masked_cross_entropy(logits: torch.Tensor, target: torch.Tensor, cand_mask: Optional[torch.Tensor] = None,
                         reduction: str = "none", ignore_index: int = -1) -> torch.Tensor:
    """== `CrossEntropyLoss(ignore_index, reduction)(logits.masked_fill(cand_mask, -inf), target)`.
    reduction: 'none' ([B], 0 at ignored rows; what SELF-PACE consumes, curriculum.py:296), 'sum' (0-dim, summed inside
    the same launch), 'mean' (mean over the non-ignored rows, follower.py:62)."""
    _not_deferred(logits, "masked_cross_entropy")
    if reduction == "none":
        return _MaskedCE.apply(logits, target, cand_mask, ignore_index, False)
    if reduction == "sum":
        return _MaskedCE.apply(logits, target, cand_mask, ignore_index, True)
    return _MaskedCEMean.apply(logits, target, cand_mask, ignore_index)


class _MonitorLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, progress, target, cand_mask, start_dist, cur_dist, ended, t, lam, per_sample, ignore_index):
        B, C = logits.shape
        dev = logits.device
        lg = logits.detach()
        if not lg.is_contiguous():
            lg = lg.contiguous()
        pr = progress.detach().reshape(B)
        if not pr.is_contiguous():
            pr = pr.contiguous()
        f32 = dict(dtype=torch.float32, device=dev)
        probs = ops.empty(B, C, **f32)
        pt = ops.empty(B, **f32)
        out = ops.empty(B, **f32) if per_sample else ops.empty((), **f32)
        stats = ops.empty(2, **f32)
        tgt = target if target.is_contiguous() else target.contiguous()
        st = _lib.load().vln_monitor_loss_fwd(lg.data_ptr(), lg.stride(0), tgt.data_ptr(), _p(_mask8(cand_mask)), pr.data_ptr(), 1,
                                              start_dist.data_ptr(), cur_dist.data_ptr(), _mask8(ended).data_ptr(), int(t), float(lam),
                                              int(per_sample), probs.data_ptr(), pt.data_ptr(), out.data_ptr(), stats.data_ptr(), B, C,
                                              ignore_index, _lib.raw_stream())
        if st:
            _lib.check(st, "vln_monitor_loss_fwd")
        ctx.save_for_backward(probs, tgt, pr, pt, stats)
        ctx.meta = (int(t), float(lam), int(per_sample), ignore_index, progress.shape)
        ctx.mark_non_differentiable(stats)
        return out, stats

    @staticmethod
    def backward(ctx, dloss, _dstats):
        probs, tgt, pr, pt, stats = ctx.saved_tensors
        t, lam, per_sample, ignore_index, pshape = ctx.meta
        B, C = probs.shape
        dl = ops.empty_like(probs)
        dp = ops.empty(B, dtype=torch.float32, device=probs.device)
        if per_sample and dloss.dim() > 0 and dloss.stride(0) != 0:
            stride = 1
            if not dloss.is_contiguous():
                dloss = dloss.contiguous()
        else:
            stride = 0
        st = _lib.load().vln_monitor_loss_bwd(probs.data_ptr(), tgt.data_ptr(), pr.data_ptr(), 1, pt.data_ptr(), stats.data_ptr(),
                                              dloss.data_ptr(), stride, t, lam, per_sample, dl.data_ptr(), dp.data_ptr(), B, C,
                                              ignore_index, _lib.raw_stream())
        if st:
            _lib.check(st, "vln_monitor_loss_bwd")
        return dl, dp.view(pshape), None, None, None, None, None, None, None, None, None


def monitor_mixed_loss(logits: torch.Tensor, target: torch.Tensor, cand_mask: Optional[torch.Tensor], progress: torch.Tensor,
                       start_dist: torch.Tensor, cur_dist: torch.Tensor, ended: torch.Tensor, t: int, lam: float,
                       per_sample: bool = False, ignore_index: int = -1):
    """The Self-Monitor agent's step loss (monitor.py:146-165) as ONE launch each way:

        cur_action_loss = CrossEntropyLoss(ignore_index)(logits.masked_fill(cand_mask, -inf), target)
        prog_target = (start_dist - cur_dist) / start_dist;  = 1 where cur_dist <= 3;  = cur_prog_val (detached) where ended
        cur_loss = cur_action_loss                                         (t == 0)
                 = lam * MSELoss()(cur_prog_val, prog_target) + (1 - lam) * cur_action_loss      (t > 0)

    `per_sample` = the curriculum criteria (reduction="none", monitor.py:151,162): a [B] loss.  start_dist / cur_dist are [B]
    float32 DEVICE tensors and `ended` a [B] bool/uint8 device tensor: the progress target is formed on the device, where the
    reference copies cur_prog_val to the host every step (monitor.py:157) to build it in numpy.
    Returns (loss, progress_mse): progress_mse = the mean MSE the agent logs as `progress_loss` (0-dim, no gradient;
    meaningless at t == 0, where the reference does not compute it)."""
    _not_deferred(logits, "monitor_mixed_loss")
    loss, stats = _MonitorLoss.apply(logits, progress, target, cand_mask, start_dist, cur_dist, ended, t, lam, per_sample, ignore_index)
    return loss, stats[0]


class _RolloutMonitorLoss(torch.autograd.Function):
    """sum_t monitor_mixed_loss(step t) in one launch each way (vln_monitor_loss_multi_fwd / _bwd); inputs: logits_0 .. logits_{T-1},
    progress_0 .. progress_{T-1}."""

    @staticmethod
    def forward(ctx, meta, *tensors):
        targets, masks, starts, curs, endeds, lam, ignore_index = meta
        T = len(targets)
        logits, progs = tensors[:T], tensors[T:]
        B = logits[0].shape[0]
        dev = logits[0].device
        f32 = dict(dtype=torch.float32, device=dev)
        lib = _lib.load()
        out = ops.empty((), **f32)
        stats = ops.empty(T, 2, **f32)
        keep, steps = [], []
        for lg, pr, tg, mk, sd, cd, en in zip(logits, progs, targets, masks, starts, curs, endeds):
            lg = lg.detach()
            if not lg.is_contiguous():
                lg = lg.contiguous()
            pr = pr.detach().reshape(B)
            if not pr.is_contiguous():
                pr = pr.contiguous()
            C_ = lg.shape[1]
            probs, pt = ops.empty(B, C_, **f32), ops.empty(B, **f32)
            tg = tg if tg.is_contiguous() else tg.contiguous()
            m8, e8 = _mask8(mk), _mask8(en)
            keep.append((lg, pr, tg, m8, sd, cd, e8, probs, pt))
            steps.append(_lib.MonitorLossStep(lg.data_ptr(), lg.stride(0), tg.data_ptr(), _p(m8), probs.data_ptr(), pr.data_ptr(), 1,
                                              sd.data_ptr(), cd.data_ptr(), e8.data_ptr(), pt.data_ptr(), None, None, C_))
        n = _lib.MONITOR_LOSS_MAX_STEPS
        for i in range(0, T, n):
            chunk = steps[i:i + n]
            arr = (_lib.MonitorLossStep * len(chunk))(*chunk)
            st = lib.vln_monitor_loss_multi_fwd(arr, len(chunk), B, i, lam, ignore_index, out.data_ptr(), stats.data_ptr() + 8 * i, 1 if i else 0,
                                                _lib.raw_stream())
            if st:
                _lib.check(st, "vln_monitor_loss_multi_fwd")
        ctx.keep, ctx.stats, ctx.cfg = keep, stats, (lam, ignore_index, [p.shape for p in progs])
        ctx.mark_non_differentiable(stats)
        return out, stats

    @staticmethod
    def backward(ctx, dloss, _dstats):
        keep, stats = ctx.keep, ctx.stats
        lam, ignore_index, pshapes = ctx.cfg
        T, B = len(keep), keep[0][0].shape[0]
        lib = _lib.load()
        dloss = dloss.contiguous()
        steps, dls, dps = [], [], []
        for lg, pr, tg, m8, sd, cd, e8, probs, pt in keep:
            dl, dp = ops.empty_like(probs), ops.empty(B, dtype=torch.float32, device=probs.device)
            dls.append(dl); dps.append(dp)
            steps.append(_lib.MonitorLossStep(lg.data_ptr(), lg.stride(0), tg.data_ptr(), _p(m8), probs.data_ptr(), pr.data_ptr(), 1,
                                              sd.data_ptr(), cd.data_ptr(), e8.data_ptr(), pt.data_ptr(), dl.data_ptr(), dp.data_ptr(), probs.shape[1]))
        n = _lib.MONITOR_LOSS_MAX_STEPS
        for i in range(0, T, n):
            chunk = steps[i:i + n]
            arr = (_lib.MonitorLossStep * len(chunk))(*chunk)
            st = lib.vln_monitor_loss_multi_bwd(arr, len(chunk), B, i, lam, ignore_index, stats.data_ptr() + 8 * i, dloss.data_ptr(), _lib.raw_stream())
            if st:
                _lib.check(st, "vln_monitor_loss_multi_bwd")
        ctx.keep = None
        return (None, *dls, *[dp.view(sh) for dp, sh in zip(dps, pshapes)])


class RolloutMonitorLoss:
    """The Self-Monitor agent's loss of a whole rollout, `sum_t cur_loss_t` (monitor.py:146-165,196: CE at t = 0, then lam * MSE(progress)
    + (1 - lam) * CE, both with their default mean reductions), evaluated ONCE after the last decoder step: `add(...)` per step only
    records the operands (in step order, from t = 0), `sum()` is one launch forward and one backward for all steps.  Same numbers as
    summing `monitor_mixed_loss` over the steps.  `progress_mse` (after `sum()`): the [T] mean progress MSEs the agent logs."""

    def __init__(self, lam: float, ignore_index: int = -1):
        self.lam, self.ignore_index = float(lam), ignore_index
        self.rows = []
        self.progress_mse = None

    def add(self, logits, target, cand_mask, progress, start_dist, cur_dist, ended):
        if logits.dim() != 2 or (self.rows and logits.shape[0] != self.rows[0][0].shape[0]):
            raise ValueError("RolloutMonitorLoss.add: logits must be [B, C] with the same B every step")
        _not_deferred(logits, "RolloutMonitorLoss.add")
        self.rows.append((logits, progress, target, cand_mask, start_dist, cur_dist, ended))

    def sum(self) -> torch.Tensor:
        if not self.rows:
            raise ValueError("RolloutMonitorLoss.sum: no steps recorded")
        cols = list(zip(*self.rows))
        out, stats = _RolloutMonitorLoss.apply((tuple(cols[2]), tuple(cols[3]), tuple(cols[4]), tuple(cols[5]), tuple(cols[6]), self.lam, self.ignore_index),
                                               *cols[0], *cols[1])
        self.progress_mse = stats[:, 0]
        self.rows = []
        return out


class _RolloutCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, meta, *logits):
        targets, masks, ignore_index, scale, per_sample = meta[:5]
        step_mean = len(meta) > 5 and bool(meta[5])
        T, B = len(logits), logits[0].shape[0]
        dev = logits[0].device
        lib = _lib.load()
        recs = [getattr(lg, "_vln_rec", None) for lg in logits]          # EnvDropDecoder steps (None for other producers)
        pending = [r for r in recs if r is not None and r.slot is not None and r.io.defer_logits and not r.flushed]
        if pending:                                                       # EnvDropDecoder.defer_logits: form the logits now, all steps
            for mod in {id(r.mod): r.mod for r in pending}.values():      # at once per decoder
                mod.logit_branch_forward([r for r in pending if r.mod is mod])
        out = ops.empty(B if per_sample else (), dtype=torch.float32, device=dev)
        keep, steps = [], []
        for lg, tg, mk in zip(logits, targets, masks):
            lg = lg.detach()
            if not lg.is_contiguous():
                lg = lg.contiguous()
            C_ = lg.shape[1]
            probs = ops.empty(B, C_, dtype=torch.float32, device=dev)
            tg = tg if tg.is_contiguous() else tg.contiguous()
            m8 = _mask8(mk)
            keep.append((lg, tg, m8, probs))
            steps.append(_lib.CeStep(lg.data_ptr(), lg.stride(0), tg.data_ptr(), _p(m8), probs.data_ptr(), None, C_))
        inv = ops.empty(T, dtype=torch.float32, device=dev) if step_mean else None       # 1 / (rows with a target) per step
        for i in range(0, T, _lib.CE_MAX_STEPS):
            chunk = steps[i:i + _lib.CE_MAX_STEPS]
            arr = (_lib.CeStep * len(chunk))(*chunk)
            st = lib.vln_masked_ce_multi_fwd(arr, len(chunk), B, ignore_index, scale, None if per_sample else out.data_ptr(),
                                             out.data_ptr() if per_sample else None, 1 if i else 0,
                                             None if inv is None else inv.data_ptr() + 4 * i, _lib.raw_stream())
            if st:
                _lib.check(st, "vln_masked_ce_multi_fwd")
        ctx.keep, ctx.ignore_index, ctx.scale, ctx.per_sample, ctx.inv = keep, ignore_index, scale, per_sample, inv
        ctx.recs = recs                                                   # their logit branch of the backward can be batched too
        return out

    @staticmethod
    def backward(ctx, dloss):
        keep = ctx.keep
        T = len(keep)
        B = keep[0][0].shape[0]
        lib = _lib.load()
        dloss = dloss.contiguous()
        recs = ctx.recs
        ctx.recs = None
        batched = bool(recs) and all(r is not None and r.slot is not None for r in recs)
        if batched:
            mod = recs[0].mod
            batched = bool(getattr(mod, "batch_logit_backward", False)) and all(r.mod is mod and r.B == B for r in recs)
        # The decoder's rollout-wide logit branch runs ONCE per step record: if another rollout-wide consumer of the same logits
        # (losses.RolloutSampler) took it already, these d logits go back to autograd and reach the steps as `dlogit`, which
        # the step backward ADDS to the branch's result (envdrop.hip) -- nothing is dropped or counted twice.
        if batched and any(r.dhtd_ext for r in recs):
            batched = False
        if batched and ctx.inv is not None:
            batched = False                     # (the decoder's rollout-wide branch forms sum-reduction d logits only)
        if batched and not ctx.per_sample:
            # every consumer of these d logits is the decoder's rollout-wide logit branch: it forms them on the fly from the
            # saved probabilities (no d logits tensors, no launch of its own here)
            mod.logit_branch_backward([(r, None) for r in recs],
                                      ce=([(k[3], k[1]) for k in keep], dloss, ctx.scale, ctx.ignore_index))
            ctx.keep = None
            return (None,) * (T + 1)
        steps, outs = [], []
        for lg, tg, m8, probs in keep:
            dl = ops.empty_like(probs)
            outs.append(dl)
            steps.append(_lib.CeStep(lg.data_ptr(), lg.stride(0), tg.data_ptr(), _p(m8), probs.data_ptr(), dl.data_ptr(), probs.shape[1]))
        for i in range(0, T, _lib.CE_MAX_STEPS):
            chunk = steps[i:i + _lib.CE_MAX_STEPS]
            arr = (_lib.CeStep * len(chunk))(*chunk)
            st = lib.vln_masked_ce_multi_bwd(arr, len(chunk), B, ctx.ignore_index, ctx.scale, dloss.data_ptr(), 1 if ctx.per_sample else 0,
                                             None if ctx.inv is None else ctx.inv.data_ptr() + 4 * i, _lib.raw_stream())
            if st:
                _lib.check(st, "vln_masked_ce_multi_bwd")
        ctx.keep = None
        if batched:
            # the decoder's rollout-wide branch consumes these d logits here and now: autograd gets None for them, so whatever
            # still reaches a step's backward as `dlogit` comes from OTHER consumers of the same logits (sampled log-probs,
            # entropy, a per-step loss) and is added there (envdrop.hip step_bwd_issue) -- nothing is dropped or counted twice
            mod.logit_branch_backward(list(zip(recs, outs)))
            return (None,) * (T + 1)
        return (None, *outs)


class RolloutCE:
    """The imitation loss of a whole rollout, `ml_loss = sum_t CrossEntropyLoss(ignore_index, reduction="sum")(
    logits_t.masked_fill(cand_mask_t, -inf), target_t)` (envdrop.py:173-179 accumulated over the steps), evaluated ONCE
    after the last decoder step: `add(...)` per step only records the operands, `sum()` is one launch forward and one
    backward for all T steps (nothing in the rollout depends on the loss value, so the per-step launches only lengthen the
    stream).  Same numbers as summing `masked_cross_entropy(..., "sum")` over the steps; steps may differ in C."""

    def __init__(self, ignore_index: int = -1):
        self.ignore_index = ignore_index
        self.logits, self.targets, self.masks = [], [], []

    def add(self, logits: torch.Tensor, target: torch.Tensor, cand_mask: Optional[torch.Tensor] = None):
        if logits.dim() != 2 or (self.logits and logits.shape[0] != self.logits[0].shape[0]):
            raise ValueError("RolloutCE.add: logits must be [B, C] with the same B every step")
        ops.check_live(logits, "RolloutCE.add")
        self.logits.append(logits); self.targets.append(target); self.masks.append(cand_mask)

    def sum(self, scale: float = 1.0) -> torch.Tensor:
        """`scale` multiplies the total inside the launch (the agents' `* ML_WEIGHT / batch_size`, envdrop.py:268)."""
        if not self.logits:
            raise ValueError("RolloutCE.sum: no steps recorded")
        return self._run(scale, False)

    def per_sample(self, scale: float = 1.0) -> torch.Tensor:
        """[B]: every episode's own loss summed over the steps -- the vector SELF-PACE multiplies by its weights
        (`reduction="none"` criterion, envdrop.py:70,178-179; curriculum.py:296) -- same single launch each way."""
        if not self.logits:
            raise ValueError("RolloutCE.per_sample: no steps recorded")
        return self._run(scale, True)

    def mean_per_step(self, scale: float = 1.0) -> torch.Tensor:
        """sum_t CrossEntropyLoss(ignore_index)(logits_t, target_t) with the criterion's DEFAULT (mean) reduction on every step's batch --
        the Speaker-Follower agent's loss (follower.py:62,123-139) -- in the same single launch each way."""
        if not self.logits:
            raise ValueError("RolloutCE.mean_per_step: no steps recorded")
        return self._run(scale, False, True)

    def _run(self, scale, per_sample, step_mean=False):
        out = _RolloutCE.apply((tuple(self.targets), tuple(self.masks), self.ignore_index, float(scale), bool(per_sample), bool(step_mean)),
                               *self.logits)
        self.logits, self.targets, self.masks = [], [], []
        return out


def action_stats(logits: torch.Tensor, action: torch.Tensor, cand_mask: Optional[torch.Tensor] = None):
    """(probs, log_prob(action), entropy) of Categorical(softmax(masked logits)) with torch.distributions' clamp
    (envdrop.py:189-194) -- no autograd (sampling / logging); the differentiable A2C terms use torch ops."""
    _not_deferred(logits, "action_stats")
    lib = _lib.load()
    B, C = logits.shape
    lg = logits.detach().contiguous()
    probs = ops.empty(B, C, dtype=torch.float32, device=logits.device)
    logp = ops.empty(B, dtype=torch.float32, device=logits.device)
    ent = ops.empty(B, dtype=torch.float32, device=logits.device)
    _lib.check(lib.vln_masked_ce_fwd(_p(lg), lg.stride(0), None, _p(_mask8(cand_mask)), None, None, _p(probs), _p(action.contiguous()),
                                     _p(logp), _p(ent), B, C, -1, 0, _lib.raw_stream()),
               "vln_masked_ce_fwd")
    return probs, logp, ent


class _A2C(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logp, ent, val, reward, mask8, last_value, ended8, gamma, ent_coef):
        T, B = val.shape
        dev = val.device
        f32 = dict(dtype=torch.float32, device=dev)
        loss_b, total = ops.empty(B, **f32), ops.empty((), **f32)
        dlogp, dval = ops.empty(T, B, **f32), ops.empty(T, B, **f32)
        dent = ops.empty(T, B, **f32) if ent is not None else None
        lg, vl = logp.detach().contiguous(), val.detach().contiguous()
        en = ent.detach().contiguous() if ent is not None else None
        st = _lib.load().vln_a2c_loss_fwd(_p(lg), _p(en), _p(vl), _p(reward), _p(mask8), _p(last_value), _p(ended8), T, B, gamma,
                                          ent_coef, _p(loss_b), _p(dlogp), _p(dval), _p(dent), _p(total), _lib.raw_stream())
        if st:
            _lib.check(st, "vln_a2c_loss_fwd")
        ctx.save_for_backward(dlogp, dval, *( [dent] if dent is not None else []))
        ctx.has_ent = dent is not None
        ctx.mark_non_differentiable(total)
        return loss_b, total

    @staticmethod
    def backward(ctx, dloss_b, _dtotal):
        saved = ctx.saved_tensors
        dlogp, dval = saved[0], saved[1]
        dent = saved[2] if ctx.has_ent else None
        T, B = dval.shape
        if dloss_b.stride(0) == 0:
            stride = 0
        else:
            stride = 1
            dloss_b = dloss_b.contiguous()
        glogp = ops.empty_like(dlogp) if ctx.needs_input_grad[0] else None
        gval = ops.empty_like(dval) if ctx.needs_input_grad[2] else None
        gent = ops.empty_like(dent) if (dent is not None and ctx.needs_input_grad[1]) else None
        st = _lib.load().vln_a2c_loss_bwd(_p(dloss_b), stride, _p(dlogp), _p(dval), _p(dent), T, B, _p(glogp), _p(gval), _p(gent),
                                          _lib.raw_stream())
        if st:
            _lib.check(st, "vln_a2c_loss_bwd")
        return glogp, gent, gval, None, None, None, None, None, None


def _stack(x, dtype=None):
    t = torch.stack(list(x)) if isinstance(x, (list, tuple)) else x
    return t if dtype is None or t.dtype == dtype else t.to(dtype)


def a2c_loss(log_probs, entropies, values, rewards, masks, last_value, ended, gamma: float, normalize: str = "total",
             per_sample: bool = False, entropy_coef: float = 0.01):
    """The A2C part of EnvDropAgent.rollout (envdrop.py:235-264) as ONE launch (+ one in backward).
    log_probs / entropies / values / rewards / masks: lists of T tensors [B] or stacked [T,B] on the GPU (entropies None
    when feedback != "sample"); last_value [B] (used detached), ended [B] bool.  Returns (loss, total): loss is [B] when
    per_sample (SELF-PACE, curriculum.py:296) else 0-dim; total = number of (step, episode) pairs that were running, a
    0-dim tensor -- the 'total' normaliser is applied on the device, nothing synchronises."""
    lp, vl = _stack(log_probs), _stack(values)
    en = _stack(entropies) if entropies is not None else None
    rw = _stack(rewards, torch.float32).contiguous()
    mk = _stack(masks)
    mk8 = (mk.contiguous().view(torch.uint8) if mk.dtype == torch.bool else mk.to(torch.uint8)).contiguous()
    en8 = ended.contiguous().view(torch.uint8) if ended.dtype == torch.bool else ended.to(torch.uint8).contiguous()
    lv = last_value.detach().to(torch.float32).contiguous()
    loss_b, total = _A2C.apply(lp, en, vl, rw, mk8, lv, en8, float(gamma), float(entropy_coef))
    loss = loss_b if per_sample else loss_b.sum()
    if normalize == "total":
        loss = loss / total
    elif normalize == "batch":
        loss = loss / lp.shape[1]
    elif normalize != "none":
        raise ValueError("normalize must be 'total', 'batch' or 'none'")
    return loss, total


class _Categorical(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, cand_mask, action, seed, offset):
        B, C = logits.shape
        dev = logits.device
        lg = logits.detach()
        if not lg.is_contiguous():
            lg = lg.contiguous()
        probs = ops.empty(B, C, dtype=torch.float32, device=dev)
        logp = ops.empty(B, dtype=torch.float32, device=dev)
        ent = ops.empty(B, dtype=torch.float32, device=dev)
        act = action.contiguous() if action is not None else ops.empty(B, dtype=torch.int64, device=dev)
        st = _lib.load().vln_categorical_fwd(_p(lg), lg.stride(0), _p(_mask8(cand_mask)), _p(action if action is None else act),
                                             None if action is not None else act.data_ptr(), _p(probs), _p(logp), _p(ent), B, C,
                                             seed, offset, None, _lib.raw_stream())
        if st:
            _lib.check(st, "vln_categorical_fwd")
        ctx.save_for_backward(probs, act)
        ctx.mark_non_differentiable(act)
        return act, logp, ent

    @staticmethod
    def backward(ctx, _da, dlogp, dent):
        probs, act = ctx.saved_tensors
        B, C = probs.shape
        dl = ops.empty_like(probs)
        gl = dlogp.contiguous() if dlogp is not None else None
        ge = dent.contiguous() if dent is not None else None
        st = _lib.load().vln_categorical_bwd(_p(probs), _p(act), _p(gl), _p(ge), _p(dl), B, C, _lib.raw_stream())
        if st:
            _lib.check(st, "vln_categorical_bwd")
        return dl, None, None, None, None


_sample_calls = [0]


def sample_action(logits: torch.Tensor, cand_mask: Optional[torch.Tensor] = None, action: Optional[torch.Tensor] = None,
                  seed: int = 0x5A3B1E, offset: Optional[int] = None):
    """The sampled-action branch of a rollout step (envdrop.py:186-195) as ONE launch, differentiable:
    `probs = softmax(logits.masked_fill(cand_mask, -inf)); c = Categorical(probs); a = c.sample();
    policy_log_probs.append(c.log_prob(a)); entropys.append(c.entropy())`  ->  (a, log_prob [B], entropy [B]).
    `action` given: its log-prob / entropy instead of a draw (teacher or injected actions).  Draws come from the kernels'
    Philox stream (seed, offset; offset defaults to a running counter), not from torch's generator."""
    _not_deferred(logits, "sample_action")
    if offset is None:
        _sample_calls[0] += 1
        offset = _sample_calls[0]
    return _Categorical.apply(logits, cand_mask, action, int(seed), int(offset))


class _RolloutStats(torch.autograd.Function):
    """logp / entropy [T,B] of a sampled rollout as ONE autograd node over the T steps' logits (RolloutSampler.stats)."""

    @staticmethod
    def forward(ctx, meta, *logits):
        keep, logp, ent = meta
        ctx.keep = keep
        ctx.recs = [getattr(lg, "_vln_rec", None) for lg in logits]
        return logp, ent

    @staticmethod
    def backward(ctx, dlogp, dent):
        keep, recs = ctx.keep, ctx.recs
        ctx.keep = ctx.recs = None
        T = len(keep)
        B = keep[0][0].shape[0]
        gl = dlogp.contiguous() if dlogp is not None else None
        ge = dent.contiguous() if dent is not None else None
        outs, steps = [], []
        for probs, act in keep:
            dl = ops.empty_like(probs)
            outs.append(dl)
            steps.append(_lib.CatStep(probs.data_ptr(), act.data_ptr(), dl.data_ptr(), probs.shape[1]))
        lib = _lib.load()
        for i in range(0, T, _lib.CE_MAX_STEPS):
            chunk = steps[i:i + _lib.CE_MAX_STEPS]
            arr = (_lib.CatStep * len(chunk))(*chunk)
            st = lib.vln_categorical_multi_bwd(arr, len(chunk), B, None if gl is None else gl[i:].data_ptr(),
                                               None if ge is None else ge[i:].data_ptr(), _lib.raw_stream())
            if st:
                _lib.check(st, "vln_categorical_multi_bwd")
        batched = all(r is not None and r.slot is not None for r in recs)
        if batched:
            mod = recs[0].mod
            batched = bool(getattr(mod, "batch_logit_backward", False)) and all(r.mod is mod and r.B == B for r in recs)
        if batched and any(r.dhtd_ext for r in recs):      # the branch ran for another consumer (losses.RolloutCE): see there
            batched = False
        if batched:
            # the decoder's rollout-wide logit branch takes these d logits here and now (one multi-step weighted sum + one GEMM
            # for all steps); autograd gets None, so a step's backward only sees what OTHER consumers of its logits sent
            mod.logit_branch_backward(list(zip(recs, outs)))
            return (None,) * (T + 1)
        return (None, *outs)


class RolloutSampler:
    """The sampled-action branch of a WHOLE rollout (envdrop.py:186-195 every step, :235-264 afterwards):

        s = RolloutSampler()
        for t in range(T):  a_t = s.step(logits_t, cand_mask_t)      # one launch: mask, softmax, draw, log-prob, entropy
        logp, ent = s.stats()                                        # [T,B] each, ONE autograd node for all steps

    Same draws and numbers as calling `sample_action` every step; the difference is the backward: the d logits of all T
    steps come from one launch at the root of the backward and -- with an EnvDropDecoder -- go through the decoder's
    rollout-wide logit branch (one multi-step weighted sum, one GEMM) instead of three launches on every step's chain."""

    def __init__(self, seed: int = 0x5A3B1E, capacity: int = _lib.CE_MAX_STEPS, clock=None):
        """clock (runtime.DeviceClock): the draws' Philox offsets are (clock word + call index since the last tick) * 8 instead of a
        host counter -- the launch arguments repeat from iteration to iteration (whole-iteration / segmented hipGraphs), the draws
        do not."""
        self.seed, self.cap, self.clock = int(seed), int(capacity), clock
        self.logits, self.keep = [], []
        self.logp = self.ent = None

    def step(self, logits: torch.Tensor, cand_mask: Optional[torch.Tensor] = None, action: Optional[torch.Tensor] = None,
             offset: Optional[int] = None) -> torch.Tensor:
        _not_deferred(logits, "RolloutSampler.step")
        B, C = logits.shape
        dev = logits.device
        t = len(self.logits)
        if self.logp is None:
            self.logp = ops.empty(self.cap, B, dtype=torch.float32, device=dev)
            self.ent = ops.empty(self.cap, B, dtype=torch.float32, device=dev)
        if t >= self.cap or B != self.logp.shape[1]:
            raise ValueError(f"RolloutSampler.step: more than {self.cap} steps, or the batch size changed")
        base = None
        if offset is None and self.clock is not None:
            offset, base = self.clock.rel(("RolloutSampler", self.seed)) * 8, self.clock.ptr
        elif offset is None:
            _sample_calls[0] += 1
            offset = _sample_calls[0]
        lg = logits.detach()
        if not lg.is_contiguous():
            lg = lg.contiguous()
        probs = ops.empty(B, C, dtype=torch.float32, device=dev)
        act = action.contiguous() if action is not None else ops.empty(B, dtype=torch.int64, device=dev)
        st = _lib.load().vln_categorical_fwd(_p(lg), lg.stride(0), _p(_mask8(cand_mask)), _p(action if action is None else act),
                                             None if action is not None else act.data_ptr(), _p(probs), self.logp[t].data_ptr(),
                                             self.ent[t].data_ptr(), B, C, self.seed, int(offset), base, _lib.raw_stream())
        if st:
            _lib.check(st, "vln_categorical_fwd")
        self.logits.append(logits); self.keep.append((probs, act))
        return act

    def bind(self, B: int, C: int, dev, action: Optional[torch.Tensor] = None, offset: Optional[int] = None):
        """The arguments of step t's draw for a caller that issues it INSIDE its own launch (EnvDropDecoder.forward(sampler=...):
        candidate dots + mask + softmax + draw in one launch, vln_envdrop_step.s_*): -> (probs [B,C], action [B] int64, logp row
        address, entropy row address, seed, offset, device offset base | None).  `commit(logits)` then records the step."""
        t = len(self.logits)
        if self.logp is None:
            self.logp = ops.empty(self.cap, B, dtype=torch.float32, device=dev)
            self.ent = ops.empty(self.cap, B, dtype=torch.float32, device=dev)
        if t >= self.cap or B != self.logp.shape[1]:
            raise ValueError(f"RolloutSampler: more than {self.cap} steps, or the batch size changed")
        base = None
        if offset is None and self.clock is not None:
            offset, base = self.clock.rel(("RolloutSampler", self.seed)) * 8, self.clock.ptr
        elif offset is None:
            _sample_calls[0] += 1
            offset = _sample_calls[0]
        probs = ops.empty(B, C, dtype=torch.float32, device=dev)
        act = action.contiguous() if action is not None else ops.empty(B, dtype=torch.int64, device=dev)
        self._bound = (probs, act)
        return probs, act, self.logp[t].data_ptr(), self.ent[t].data_ptr(), self.seed, int(offset), base

    def commit(self, logits: torch.Tensor) -> torch.Tensor:
        probs, act = self._bound
        self._bound = None
        self.logits.append(logits); self.keep.append((probs, act))
        return act

    def stats(self):
        """(log_prob [T,B], entropy [T,B]) of the recorded steps, differentiable w.r.t. every step's logits."""
        T = len(self.logits)
        if T == 0:
            raise ValueError("RolloutSampler.stats: no steps recorded")
        out = _RolloutStats.apply((self.keep, self.logp[:T], self.ent[:T]), *self.logits)
        self.logits, self.keep, self.logp, self.ent = [], [], None, None
        return out
