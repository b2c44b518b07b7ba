"""Operator-level autograd bindings over the C ABI (used by the Follower / Self-Monitor decoders and usable on
their own).  Each Function's forward/backward is a handful of HIP launches; weight shadows (transposed /
bf16 copies) are cached per parameter version."""
from __future__ import annotations

import ctypes as C
import weakref
from typing import Optional

import torch

from . import _lib, ops

_p = ops._p


class _ShadowCache:
    """param -> {(kind, dtype): tensor}, invalidated when the parameter's version or storage changes."""

    def __init__(self):
        self._d = {}

    def get(self, w: torch.Tensor, kind: str, dtype) -> torch.Tensor:
        key = (id(w), kind, dtype)
        ver = (w._version, w.data_ptr())
        hit = self._d.get(key)
        if hit is not None and hit[0] == ver and hit[2]() is w:      # id() can be recycled: check identity too
            return hit[1]
        src = w.detach()
        if kind == "t":
            t = ops.transpose_cast(src.contiguous(), dtype)
        elif dtype == torch.float32:
            t = src.contiguous()
        else:
            t = ops.cast_copy(src.contiguous(), dtype)
        if len(self._d) > 256:
            self._d.clear()
        self._d[key] = (ver, t, weakref.ref(w))
        return t


    def ensure(self, wants, fused=()):
        """Refresh every STALE shadow of `wants` = [(parameter, kind "n" | "t", dtype)] and of `fused` = [(w_ih, w_hh, dtype,
        transposed)] (the [w_ih | w_hh] matrices of _fused_lstm_weight) in ONE launch (`vln_shadow_refresh`) instead of one
        transpose / cast launch (and one torch.cat) each: a decoder's forward names all the shadows its steps will stream (round 5:
        8-14 launches per iteration at the top of the Self-Monitor / Follower rollouts).  The later `get` / `_fused_lstm_weight`
        calls then hit.  Same bytes as the single launches (the same cast / transpose of the same source)."""
        batch = ops.ShadowBatch()
        for w, kind, dtype in wants:
            key = (id(w), kind, dtype)
            ver = (w._version, w.data_ptr())
            hit = self._d.get(key)
            if hit is not None and hit[0] == ver and hit[2]() is w:
                continue
            src = w.detach()
            if kind != "t" and dtype == torch.float32:
                t = src.contiguous()
            elif not src.is_contiguous() or src.dim() != 2 or src.dtype != torch.float32:
                continue                                    # (left to get(): shapes the grouped launch does not take)
            else:
                N, K = src.shape
                t = torch.empty((K, N) if kind == "t" else (N, K), dtype=dtype, device=src.device)
                batch.add(src, dst=None if kind == "t" else t, dst_t=t if kind == "t" else None)
            if len(self._d) > 256:
                self._d.clear()
            self._d[key] = (ver, t, weakref.ref(w))
        for w_ih, w_hh, dtype, transposed in fused:
            key = (id(w_ih), id(w_hh), dtype, transposed)
            ver = (w_ih._version, w_hh._version, w_ih.data_ptr(), w_hh.data_ptr())
            hit = _fused_cache.get(key)
            if hit is not None and hit[0] == ver and hit[2]() is w_ih and hit[3]() is w_hh:
                continue
            a, b = w_ih.detach(), w_hh.detach()
            if not (a.is_contiguous() and b.is_contiguous() and a.dtype == torch.float32 and b.dtype == torch.float32):
                continue
            N, Kx, Kh = a.shape[0], a.shape[1], b.shape[1]
            if transposed:
                t = torch.empty(Kx + Kh, N, dtype=dtype, device=a.device)
                batch.add(a, dst_t=t[:Kx]); batch.add(b, dst_t=t[Kx:])
            else:
                t = torch.empty(N, Kx + Kh, dtype=dtype, device=a.device)
                batch.add(a, dst=t[:, :Kx]); batch.add(b, dst=t[:, Kx:])
            if len(_fused_cache) > 64:
                _fused_cache.clear()
            _fused_cache[key] = (ver, t, weakref.ref(w_ih), weakref.ref(w_hh))
        if len(batch.jobs) > _lib.SHADOW_MAX_JOBS:
            jobs, keep = batch.jobs, batch.keep
            for i in range(0, len(jobs), _lib.SHADOW_MAX_JOBS):
                part = ops.ShadowBatch()
                part.jobs, part.keep = jobs[i:i + _lib.SHADOW_MAX_JOBS], keep
                part.run()
        else:
            batch.run()


SHADOWS = _ShadowCache()


def wdtype(dtype, name):
    """Streaming dtype of the weight matrix `name` of a fused node.  `dtype` is the node's compute dtype, or a pair
    (compute dtype, frozenset of matrix names that are streamed in fp32 all the same) -- the per-matrix override of the bf16 mode
    (MonitorDecoder.fp32_weights; EnvDropDecoder.fp32_weights is the same idea on its own struct)."""
    if isinstance(dtype, tuple):
        return torch.float32 if name in dtype[1] else dtype[0]
    return dtype


def base_dtype(dtype):
    return dtype[0] if isinstance(dtype, tuple) else dtype


def _lin(x, W, kind, dtype, name, bias=None):
    """x @ shadow(W)^T (+ bias) with matrix `name`'s streaming dtype under `dtype` (see wdtype): an fp32-streamed matrix of the
    bf16 mode multiplies with split operands (ops.linear_fwd(split=True) = VLN_F32S), like the C-call steps do."""
    wd = wdtype(dtype, name)
    return ops.linear_fwd(x, SHADOWS.get(W, kind, wd), bias, split=(wd == torch.float32 and base_dtype(dtype) != torch.float32))


# ---- parameter gradients of the fused nodes: returned to autograd (default) or added straight into p.grad ---------------
_GRAD_IN_PLACE = [False]
GRAD_IN_PLACE_STATS = [0, 0]            # [gradients added in place, gradients returned to autograd] (diagnostic)


def set_grad_in_place(on: bool):
    """With this on, the fused decoder nodes (BnMlpFn, MonitorCoreFn, FollowerCoreFn) ADD their Linear weight / bias
    gradients into an existing `p.grad` inside the grouped launches and hand autograd `None` for those parameters: the
    ~250 AccumulateGrad `add_` launches per Self-Monitor iteration (one per parameter per decoder step) disappear.
    Same numbers in `p.grad` after `backward()`; NOT for `torch.autograd.grad(...)` callers (they would see None), hence
    opt-in.  Parameters whose `.grad` is None (first backward after `zero_grad(set_to_none=True)`) take the normal route."""
    _GRAD_IN_PLACE[0] = bool(on)


class RolloutWgrads:
    """Parameter gradients of the per-step C calls (MonitorStepFn, FollowerStepFn, BnMlpFn) formed ONCE PER ROLLOUT.

    A per-step call that adds into `p.grad` reads and rewrites every weight gradient of its module at every decoder step: 22
    pack + 22 contraction + 22 bias launches per Self-Monitor iteration (0.7 ms of 5.05).  With this on (and
    `set_grad_in_place(True)`, every `p.grad` present) a step's backward skips those launches and records the jobs it would
    have run (`vln_param_jobs`); the operands stay where they are -- the step's saved-activation block and its backward
    scratch are SLOTS of two arenas owned by this object, so operand i of steps 0..T-1 is one segmented matrix -- and when
    autograd's backward pass ends (`queue_callback`) every group of steps is contracted in one `vln_wgrad_grouped_seg` + one
    `vln_colsum_grouped_seg`.  Same gradients up to the summation order over steps (tests/test_hip_agents.py).

    Lifetime: a slot is reused by the same step index of the NEXT rollout; BnMlpFn's output aliases its slot, like every module
    output under ops.RolloutArena it must be consumed within the iteration.  A group = (kind, rows, first weight)."""

    def __init__(self):
        self.enabled = False
        self.groups = {}
        self._queued = False
        self.stats = [0, 0]          # [steps deferred, segmented launches issued]
        self.stale_dropped = 0       # queued flushes dropped because the backward pass that queued them raised

    class _Group:
        __slots__ = ("fwd", "fwd_n", "bwd", "bwd_n", "nf", "pending", "out")

        def __init__(self):
            self.fwd = self.bwd = None
            self.fwd_n = self.bwd_n = 0
            self.nf, self.pending = 0, []
            self.out = 0             # forward steps whose backward has not deferred its jobs yet (their slots are live)

    def active(self, ctx):
        # (inside Function.forward grad mode is off: whether a backward can follow is what needs_input_grad says)
        return self.enabled and _GRAD_IN_PLACE[0] and any(ctx.needs_input_grad)

    def reset(self):
        """Forget half-finished rollouts (forwards whose backward never ran)."""
        for g in self.groups.values():
            g.nf, g.pending, g.out = 0, [], 0
        self._queued = False

    @staticmethod
    def _slot(buf, n, slot, floats, dev):
        floats = (floats + 63) & ~63
        if buf is None or n != floats or buf.shape[0] <= slot or buf.device != dev:
            cap = 8
            while cap <= slot:
                cap *= 2
            buf, n = torch.empty(cap, floats, dtype=torch.float32, device=dev), floats     # the old arena lives on in its steps' ctx
        return buf, n, buf[slot]

    def saved(self, key, floats, dev):
        """forward: (this step's saved-activation block, its slot)."""
        if self._queued:
            # A flush is still queued while a NEW forward step runs: the backward pass that queued it raised (the engine drops its
            # callbacks then) -- its deferred jobs are stale, and `_queued` would keep every later backward from queueing a flush.
            # Drop them; the failed iteration's gradients are lost with the exception the caller already saw.
            for g in self.groups.values():
                g.pending = []
            self._queued = False
            self.stale_dropped += 1
        g = self.groups.get(key)
        if g is None:
            g = self.groups[key] = RolloutWgrads._Group()
        slot = g.nf
        g.out += 1
        if slot >= 4096:
            raise _lib.VlnError("RolloutWgrads: 4096 forward steps without a backward pass; call functional.ROLLOUT_WGRADS.reset()")
        g.nf += 1
        g.fwd, g.fwd_n, t = self._slot(g.fwd, g.fwd_n, slot, floats, dev)
        return t, slot

    def scratch(self, key, slot, floats, dev):
        g = self.groups[key]
        g.bwd, g.bwd_n, t = self._slot(g.bwd, g.bwd_n, slot, floats, dev)
        return t

    def defer(self, key, slot, jobs, keep):
        g = self.groups[key]
        g.pending.append((slot, jobs, keep))
        g.out = max(0, g.out - 1)
        self.stats[0] += 1
        if not self._queued:
            self._queued = True
            torch.autograd.Variable._execution_engine.queue_callback(self.flush)

    def flush(self):
        """Issue the deferred gradients of every group (runs at the end of the backward pass; callable by hand)."""
        self._queued = False
        lib = _lib.load()
        st = _lib.raw_stream()
        for g in self.groups.values():
            pend, g.pending = sorted(g.pending, key=lambda t: t[0]), []
            if g.out == 0:           # slots restart only when no other rollout's forward steps still wait for their backward
                g.nf = 0
            i = 0
            while i < len(pend):
                base = pend[i][1]
                nw, nc = base.nw, base.nc
                ptrs = lambda pj: ([pj.w[k].dy or 0 for k in range(nw)], [pj.w[k].x or 0 for k in range(nw)], [pj.c[k].A or 0 for k in range(nc)])
                p0 = ptrs(base)
                # the longest run of consecutive slots whose operands all sit at one constant distance
                j, stride = i + 1, None
                while j < len(pend) and pend[j][0] == pend[j - 1][0] + 1 and pend[j][1].nw == nw and pend[j][1].nc == nc:
                    a, b = ptrs(pend[j - 1][1]), ptrs(pend[j][1])
                    d = tuple(tuple(y - x for x, y in zip(u, v)) for u, v in zip(a, b))
                    if any(v % 16 for u in d for v in u) or (stride is not None and d != stride):
                        break
                    stride = d
                    j += 1
                n_seg = j - i
                if stride is None:
                    stride = tuple(tuple(0 for _ in u) for u in p0)
                dev = pend[i][2][0].device
                # BnMlpFn's input BatchNorm from the first layer's weight gradient: that layer's dW / db of THIS run of steps go to
                # temporaries (not yet into the accumulated gradients), vln_bn0_grads_from_wgrad hands them on
                meta = next((k["bn0"] for k in pend[i][2] if isinstance(k, dict) and "bn0" in k), None)
                with_meta = sum(1 for q in pend[i:j] if any(isinstance(k, dict) and "bn0" in k for k in q[2]))
                if with_meta not in (0, n_seg):
                    raise _lib.VlnError("RolloutWgrads: set_bn0_grads_from_wgrad changed between the steps of one rollout")
                if meta is not None:
                    N_, K_ = meta["W"].shape
                    tmpW = ops.empty(N_, K_, dtype=torch.float32, device=dev); tmpb = ops.empty(N_, dtype=torch.float32, device=dev)
                    jw = next(k for k in range(nw) if base.w[k].dw == meta["gW"].data_ptr())
                    jc = next(k for k in range(nc) if base.c[k].out1 == meta["gb"].data_ptr())
                    base.w[jw].dw, base.w[jw].accumulate = tmpW.data_ptr(), 0
                    base.c[jc].out1, base.c[jc].accumulate = tmpb.data_ptr(), 0
                if nw:
                    dy_s = (_lib.i64 * nw)(*[v // 4 for v in stride[0]]); x_s = (_lib.i64 * nw)(*[v // 4 for v in stride[1]])
                    need = max(int(lib.vln_wgrad_grouped_ws_floats(base.w, nw, base.rows * n_seg)), 1 << 22)
                    ws = ops.workspace(dev, need)
                    _lib.check(lib.vln_wgrad_grouped_seg(base.w, dy_s, x_s, nw, base.rows, n_seg, base.precision, ws.data_ptr(), ws.numel(), st),
                               "vln_wgrad_grouped_seg")
                if nc:
                    c_s = (_lib.i64 * nc)(*[v // 4 for v in stride[2]])
                    ws = ops.workspace(dev, 1 << 22)
                    _lib.check(lib.vln_colsum_grouped_seg(base.c, c_s, nc, base.rows, n_seg, ws.data_ptr(), ws.numel(), st),
                               "vln_colsum_grouped_seg")
                if meta is not None:
                    ws = ops.workspace(dev, 1 << 22)
                    _lib.check(lib.vln_bn0_grads_from_wgrad(tmpW.data_ptr(), tmpb.data_ptr(), meta["W"].data_ptr(), meta["W"].stride(0),
                                                            meta["gamma"].data_ptr(), meta["beta"].data_ptr(), meta["gW"].data_ptr(),
                                                            meta["gb"].data_ptr(), meta["gg"].data_ptr(), meta["gbeta"].data_ptr(), N_, K_,
                                                            1, 1, 1, ws.data_ptr(), ws.numel(), st), "vln_bn0_grads_from_wgrad")
                self.stats[1] += 1
                i = j


ROLLOUT_WGRADS = RolloutWgrads()
_BN0_FROM_WGRAD = [True]


def set_bn0_grads_from_wgrad(on: bool):
    """BnMlpFn with rollout-level parameter gradients and an input that carries no gradient: the INPUT BatchNorm's d gamma / d beta
    from the first Linear layer's weight gradient (vln_bn0_grads_from_wgrad: no dz W product, no BatchNorm backward over the input
    rows) -- the default -- or by the direct path (False; A/B, and the way out when a BatchNorm weight is exactly 0)."""
    _BN0_FROM_WGRAD[0] = bool(on)


def set_rollout_wgrads(on: bool):
    """Per-step C calls (Self-Monitor step, Follower step, BN-MLP) form their parameter gradients once per rollout instead
    of once per decoder step (RolloutWgrads).  Needs `set_grad_in_place(True)`; steps whose gradients are returned to autograd
    keep the per-step launches."""
    ROLLOUT_WGRADS.enabled = bool(on)
    if not on:
        ROLLOUT_WGRADS.reset()
        ROLLOUT_WGRADS.groups.clear()            # give the arenas back


def _bn_ws(R, D, dev):
    """Scratch of the row-chunked BatchNorm form (vln_bn_fwd / vln_bn_bwd, inputs of >= 512 rows): (tensor, ptr, floats)."""
    if R < 512:
        return None, None, 0
    n = -(-R // 128) * 2 * D
    t = ops.empty(n, dtype=torch.float32, device=dev)
    return t, t.data_ptr(), n


def _gsink(p):
    """-> (tensor the launch writes, accumulate flag, value to return to autograd is None)."""
    g = p.grad
    if _GRAD_IN_PLACE[0] and g is not None and g.dtype == torch.float32 and g.shape == p.shape and g.is_contiguous():
        GRAD_IN_PLACE_STATS[0] += 1
        return g, True
    GRAD_IN_PLACE_STATS[1] += 1
    return torch.empty_like(p), False


def _gret(t, acc):
    return None if acc else t


class LinearFn(torch.autograd.Function):
    """y = act(x W^T + b) with x [M,K] fp32; act in {none,tanh,relu}.  Replaces F.linear (+ nn.Tanh/ReLU)."""

    @staticmethod
    def forward(ctx, x, weight, bias, act, dtype):
        x2 = x.reshape(-1, x.shape[-1])
        if x2.stride(-1) != 1:
            x2 = x2.contiguous()
        y = ops.linear_fwd(x2, SHADOWS.get(weight, "n", dtype), None if bias is None else bias.detach(), act)
        ctx.save_for_backward(x2, weight, y if act != ops.ACT_NONE else None)
        ctx.act, ctx.dtype, ctx.has_bias, ctx.xshape, ctx.bias = act, dtype, bias is not None, x.shape, bias
        return y.view(*x.shape[:-1], weight.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, weight, y = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1]).contiguous()
        if ctx.act == ops.ACT_TANH:
            dy2 = ops.ew(ops.EW_TANH_GRAD, dy2, y)
        elif ctx.act == ops.ACT_RELU:
            dy2 = dy2 * (y > 0).to(dy2.dtype)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = ops.linear_fwd(dy2, SHADOWS.get(weight, "t", ctx.dtype)).view(ctx.xshape)
        if ctx.needs_input_grad[1]:           # (set_grad_in_place: added into weight.grad by the launch, autograd gets None)
            g, acc = _gsink(weight)
            dw = _gret(ops.linear_wgrad(dy2, x2, out=g, accumulate=acc), acc)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            g, acc = _gsink(ctx.bias)
            db = _gret(ops.colsum(dy2, out=g, accumulate=acc), acc)
        return dx, dw, db, None, None


def linear(x, weight, bias=None, act=ops.ACT_NONE, dtype=torch.float32):
    return LinearFn.apply(x, weight, bias, act, dtype)


class AttnDotFn(torch.autograd.Function):
    """dots[b,s] = keys[b,s,:] . vec[b,:]   (torch.bmm(context, target), units.py:109/150)"""

    @staticmethod
    def forward(ctx, keys, vec):
        keys = keys.contiguous()
        vec = vec.contiguous()
        ctx.save_for_backward(keys, vec)
        return ops.attn_dot(keys, vec)

    @staticmethod
    def backward(ctx, dd):
        keys, vec = ctx.saved_tensors
        dd = dd.contiguous()
        dkeys = dvec = None
        if ctx.needs_input_grad[1]:
            dvec = ops.rows_wsum(keys, dd)
        if ctx.needs_input_grad[0]:
            dkeys = dd.unsqueeze(2) * vec.unsqueeze(1).to(dd.dtype)
        return dkeys, dvec


class SoftmaxWsumFn(torch.autograd.Function):
    """attn = softmax(mask(logits)); out[b,:] = sum_s attn[b,s] values[b,s,:]   (units.py:111-117, 152-159)"""

    @staticmethod
    def forward(ctx, values, logits, mask):
        values = values.contiguous()
        out, attn = ops.attn_softmax_wsum(values, logits.contiguous(), mask)
        ctx.save_for_backward(values, attn)
        return out, attn

    @staticmethod
    def backward(ctx, dout, dattn):
        values, attn = ctx.saved_tensors
        B, S, D = values.shape
        dalpha = None
        if dout is not None:
            dout = dout.contiguous()
            dalpha = ops.attn_dot(values, dout)
        dvalues = None
        need_dv = ctx.needs_input_grad[0] and dout is not None
        if need_dv:
            dvalues = torch.zeros(B, S, D, dtype=torch.float32, device=values.device)
            zero = torch.zeros(B, D, dtype=torch.float32, device=values.device)
        dattn_c = None if dattn is None else dattn.contiguous()
        _, dl = ops.attn_bwd(values, attn, dalpha, dattn_c, dout if need_dv else None, zero if need_dv else None,
                             dvalues, want_dl=True)
        return dvalues, dl, None


class SoftDotRowsFn(torch.autograd.Function):
    """softmax(mask(values . q)) and the weighted sum of `values` when the keys ARE the values (units.py:106-118): one
    launch forward (vln_attn_fwd_rows); backward = one launch for d(query) and d(logits) (vln_attn_bwd_rows) plus, when
    the attended tensor needs a gradient, one for d(values) = attn (x) dout + dl (x) q (vln_attn_dctx_deferred, T = 1)."""

    @staticmethod
    def forward(ctx, values, query, mask):
        values = values.contiguous()
        query = query.contiguous()
        out, attn = ops.attn_fwd_rows(values, query, mask)
        ctx.save_for_backward(values, query, attn)
        return out, attn

    @staticmethod
    def backward(ctx, dout, dattn):
        values, query, attn = ctx.saved_tensors
        B, S, D = values.shape
        if dout is None:
            dout = torch.zeros(B, D, dtype=torch.float32, device=values.device)
        dout = dout.contiguous()
        need_dv = ctx.needs_input_grad[0]
        dq, dl = ops.attn_bwd_rows(values, attn, dout, dattn, want_dl=need_dv)
        dvalues = None
        if need_dv:
            if D % 4 == 0:
                dvalues = torch.empty(B, S, D, dtype=torch.float32, device=values.device)
                ops.attn_dctx_deferred([attn.data_ptr()], [dl.data_ptr()], [dout.data_ptr()], dout.stride(0),
                                       [query.data_ptr()], query.stride(0), dvalues)
            else:
                dvalues = attn.unsqueeze(2) * dout.unsqueeze(1) + dl.unsqueeze(2) * query.unsqueeze(1)
        return dvalues, (dq if ctx.needs_input_grad[1] else None), None


def soft_dot_core(query_vec, keys, values, mask):
    """-> (weighted values [B,Dv], attn [B,S]).  keys may be `values` itself."""
    if keys is values and values.dtype in (torch.float32, torch.bfloat16):
        return SoftDotRowsFn.apply(values, query_vec, mask)
    logits = AttnDotFn.apply(keys, query_vec)
    return SoftmaxWsumFn.apply(values, logits, mask)


class LSTMCellFn(torch.autograd.Function):
    """nn.LSTMCell (policy.py:30,96): one fused-weight GEMM [x|h] [W_ih|W_hh]^T + fused 4-gate pointwise."""

    @staticmethod
    def forward(ctx, x, h, c, w_ih, w_hh, b_ih, b_hh, dtype):
        B, H = h.shape
        xc = torch.cat((x, h), 1).contiguous()
        wcat = _fused_lstm_weight(w_ih, w_hh, dtype, False)
        K = xc.shape[1]
        ws = ops.workspace(x.device, 16 * B * 4 * H)
        lib = _lib.load()
        st = _lib.raw_stream()
        # slabs -> pointwise sums them
        y = torch.empty(B, 4 * H, dtype=torch.float32, device=x.device)
        _lib.check(lib.vln_linear_fwd(xc.data_ptr(), xc.stride(0), wcat.data_ptr(), ops._dt(wcat), wcat.stride(0),
                                      y.data_ptr(), y.stride(0), B, 4 * H, K, None, 0, ws.data_ptr(), ws.numel(), st),
                   "vln_linear_fwd")
        h1, c1, act, tc, _ = ops.lstm_pointwise_fwd(y.view(1, B, 4 * H), b_ih.detach(), b_hh.detach(), c.detach().contiguous())
        ctx.save_for_backward(xc, c.detach().contiguous(), act, tc, w_ih, w_hh)
        ctx.dtype, ctx.kx = dtype, x.shape[1]
        return h1, c1

    @staticmethod
    def backward(ctx, dh1, dc1):
        xc, c0, act, tc, w_ih, w_hh = ctx.saved_tensors
        dg, dc0 = ops.lstm_pointwise_bwd(None if dh1 is None else dh1.contiguous(), None,
                                         None if dc1 is None else dc1.contiguous(), act, tc, c0)
        wcat_t = _fused_lstm_weight(w_ih, w_hh, ctx.dtype, True)
        dxc = ops.linear_fwd(dg, wcat_t)
        kx = ctx.kx
        dwi = ops.linear_wgrad(dg, xc[:, :kx])
        dwh = ops.linear_wgrad(dg, xc[:, kx:])
        db = ops.colsum(dg)
        return dxc[:, :kx], dxc[:, kx:], dc0, dwi, dwh, db, db.clone(), None


_fused_cache = {}


def _fused_lstm_weight(w_ih, w_hh, dtype, transposed):
    key = (id(w_ih), id(w_hh), dtype, transposed)
    ver = (w_ih._version, w_hh._version, w_ih.data_ptr(), w_hh.data_ptr())
    hit = _fused_cache.get(key)
    if hit is not None and hit[0] == ver and hit[2]() is w_ih and hit[3]() is w_hh:
        return hit[1]
    wc = torch.cat((w_ih.detach(), w_hh.detach()), 1).contiguous()
    t = ops.transpose_cast(wc, dtype) if transposed else (wc if dtype == torch.float32 else ops.cast_copy(wc, dtype))
    if len(_fused_cache) > 64:
        _fused_cache.clear()
    _fused_cache[key] = (ver, t, weakref.ref(w_ih), weakref.ref(w_hh))
    return t


class DropoutFn(torch.autograd.Function):
    """nn.Dropout with the kernels' Philox stream (seed, offset): mask regenerated in backward.  base: the device word the kernels
    add (x 8) to `offset` (runtime.DeviceClock: the launch arguments then repeat from iteration to iteration), or None."""

    @staticmethod
    def forward(ctx, x, p, seed, offset, base=None):
        ctx.cfg = (p, seed, offset, base)
        return _scale_dropout(x, p, seed, offset, base)

    @staticmethod
    def backward(ctx, dy):
        p, seed, offset, base = ctx.cfg
        return _scale_dropout(dy, p, seed, offset, base), None, None, None, None


def _scale_dropout(x, p, seed, offset, base=None):
    """y = x * mask(seed, offset) in ONE launch (vln_scale_dropout); the mask is a function of the flat element index."""
    xc = x.contiguous()
    if xc.dtype != torch.float32:
        xc = xc.float()
    y = ops.empty(xc.shape, dtype=torch.float32, device=xc.device)
    cols = xc.shape[-1] if xc.dim() > 0 else 1
    rows = xc.numel() // max(cols, 1)
    st = _lib.load().vln_scale_dropout(_p(xc), cols, _p(y), cols, rows, cols, seed, offset, p, base, _lib.raw_stream())
    if st:
        _lib.check(st, "vln_scale_dropout")
    return y


def dropout(x, p: float, training: bool, seed: int, offset: int, base=None):
    if not training or p <= 0.0:
        return x
    return DropoutFn.apply(x, p, seed, offset, base)


class BatchNormFn(torch.autograd.Function):
    """nn.BatchNorm1d (+ optionally the ReLU that follows it) as ONE HIP launch forward and ONE backward
    (vln_bn_fwd / vln_bn_bwd); running statistics and num_batches_tracked are updated in place like torch's."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, nbt, training, momentum, eps, relu):
        R, D = x.shape
        xc = x.detach()
        if not xc.is_contiguous():
            xc = xc.contiguous()
        dev = x.device
        y = ops.empty(R, D, dtype=torch.float32, device=dev)
        stats = ops.empty(2, D, dtype=torch.float32, device=dev) if training else None
        use_batch = bool(training or running_mean is None)
        _wst, wsp, wsn = _bn_ws(R, D, dev)
        st = _lib.load().vln_bn_fwd(_p(xc), xc.stride(0), _p(y), y.stride(0), _p(weight), _p(bias), _p(running_mean), _p(running_var),
                                    _p(nbt) if training else None, _p(stats), None if stats is None else stats.data_ptr() + 4 * D,
                                    R, D, eps, momentum, 1 if use_batch else 0, 1 if relu else 0, 0, 0, 0.0, None, wsp, wsn,
                                    _lib.raw_stream())
        if st:
            _lib.check(st, "vln_bn_fwd")
        ctx.save_for_backward(xc, y if relu else None, weight, stats, None if use_batch else running_mean, None if use_batch else running_var)
        ctx.cfg = (use_batch, eps, relu)
        return y

    @staticmethod
    def backward(ctx, dy):
        xc, y, weight, stats, rmean, rvar = ctx.saved_tensors
        use_batch, eps, relu = ctx.cfg
        R, D = xc.shape
        dy = dy.contiguous()
        dev = xc.device
        need_w = weight is not None and ctx.needs_input_grad[1]
        dx = ops.empty(R, D, dtype=torch.float32, device=dev) if ctx.needs_input_grad[0] else None
        dgb = torch.empty(2, D, dtype=torch.float32, device=dev) if need_w else None        # parameter gradients: not arena memory
        mean_p = stats.data_ptr() if use_batch else rmean.data_ptr()
        rstd_p = stats.data_ptr() + 4 * D if use_batch else rvar.data_ptr()
        _wst, wsp, wsn = _bn_ws(R, D, dev)
        st = _lib.load().vln_bn_bwd(_p(xc), xc.stride(0), _p(dy), dy.stride(0), _p(y), 0 if y is None else y.stride(0), _p(weight),
                                    mean_p, rstd_p, _p(dx), 0 if dx is None else dx.stride(0), None if dgb is None else dgb.data_ptr(),
                                    None if dgb is None else dgb.data_ptr() + 4 * D, R, D, eps, 1 if use_batch else 0,
                                    1 if relu else 0, 0, 0, 0, 0.0, None, wsp, wsn, _lib.raw_stream())
        if st:
            _lib.check(st, "vln_bn_bwd")
        return dx, (dgb[0] if need_w else None), (dgb[1] if need_w else None), None, None, None, None, None, None, None


def batch_norm(x, weight, bias, running_mean, running_var, num_batches_tracked, training, momentum=0.1, eps=1e-5, relu=False):
    return BatchNormFn.apply(x, weight, bias, running_mean, running_var, num_batches_tracked, bool(training), float(momentum),
                             float(eps), bool(relu))


class BnMlpFn(torch.autograd.Function):
    """MLPwithBN (units.py:210-242) with use_bn and relu -- BatchNorm, then per hidden layer Linear, BatchNorm, Dropout,
    ReLU -- as ONE autograd node: forward = one BatchNorm launch per layer (statistics, normalisation, dropout, ReLU and
    the zeroing of padded rows fused, vln_bn_fwd) + one GEMM per layer; backward = the mirrored chain with every weight
    gradient of the call in one grouped launch and every bias gradient in another.  `cfg` = (training, eps, momentum,
    dtype, [(p_drop, seed, offset) per hidden layer]); `row_zero` [R] uint8/bool or None zeroes output rows
    (policy.py:148-149).  tensors = bn0.w, bn0.b, then per layer W, b, bn.w, bn.b; buffers (running stats, counters)
    ride in `bufs` and are updated in place."""

    @staticmethod
    def _c_desc(x, rz, cfg, bufs, tensors):
        """vln_bn_mlp filled from the module's tensors (+ the objects that keep its pointers alive)."""
        training, eps, momentum, dtype, drops = cfg[:5]
        seg = cfg[5] if len(cfg) > 5 else None          # (R1, [segment 1's dropout offset per layer]): two batches in one call
        nl = (len(tensors) - 2) // 4
        m = _lib.BnMlp()
        x2 = seg[2] if (seg and len(seg) > 2) else None         # the second batch in its own array (read in place)
        m.R, m.D0, m.nl = x.shape[0] + (0 if x2 is None else x2.shape[0]), x.shape[1], nl
        m.R1 = seg[0] if seg else 0
        if x2 is not None:
            m.x2, m.ldx2 = x2.data_ptr(), x2.stride(0)
        wdt = wdtype(dtype, "mlp")                  # the Linear layers' streaming dtype (per-matrix fp32 override: functional.wdtype)
        base = base_dtype(dtype)
        m.wtype = ops.F32 if base == torch.float32 else (ops.F32S if wdt == torch.float32 else ops.BF16)
        m.training, m.eps, m.momentum = int(training), eps, momentum
        keep = []

        def affine(a, w, b, buf):
            a.gamma, a.beta = _p(w), _p(b)
            a.run_mean, a.run_var, a.nbt = _p(buf[0]), _p(buf[1]), _p(buf[2])
        affine(m.bn0, tensors[0], tensors[1], bufs[0])
        for i in range(nl):
            W, b, gw, gb = tensors[2 + 4 * i: 6 + 4 * i]
            wn, wt = SHADOWS.get(W, "n", wdt), SHADOWS.get(W, "t", wdt)
            keep += [wn, wt]
            l = m.layer[i]
            l.w, l.w_t, l.w_f32, l.b = wn.data_ptr(), wt.data_ptr(), W.data_ptr(), _p(b)
            affine(l.bn, gw, gb, bufs[1 + i])
            l.out = W.shape[0]
            l.p_drop, l.seed, l.offset = drops[i][:3]
            l.offset2 = seg[1][i] if seg else 0
            if len(drops[i]) > 3:
                m.offset_base_dev = drops[i][3]                     # runtime.DeviceClock word: every layer's offset is relative to it
        m.row_zero = _p(rz)
        return m, keep

    @staticmethod
    def _forward_c(ctx, x, rz, cfg, bufs, tensors):
        lib = _lib.load()
        m, keep = BnMlpFn._c_desc(x, rz, cfg, bufs, tensors)
        dev = x.device
        ctx.rw = None
        if ROLLOUT_WGRADS.active(ctx) and cfg[0]:
            key = ("bn_mlp", m.R, id(tensors[2]))
            saved, slot = ROLLOUT_WGRADS.saved(key, lib.vln_bn_mlp_saved_floats(m), dev)
            ctx.rw = (key, slot)
        else:
            saved = ops.empty(lib.vln_bn_mlp_saved_floats(m), dtype=torch.float32, device=dev)
        ws = ops.workspace(dev, lib.vln_bn_mlp_ws_floats(m))
        rc = lib.vln_bn_mlp_fwd(m, x.data_ptr(), x.stride(0), saved.data_ptr(), ws.data_ptr(), ws.numel(), _lib.raw_stream())
        if rc:
            _lib.check(rc, "vln_bn_mlp_fwd")
        nl = m.nl
        off = lib.vln_bn_mlp_out_offset(m)
        out_dim = m.layer[nl - 1].out
        ctx.cfg, ctx.bufs, ctx.nl, ctx.rz, ctx.c_call = cfg, bufs, nl, rz, True
        # the second batch is read in place again by the backward (BatchNorm's d gamma): it rides in `cfg`, not among the saved
        # tensors, so its version is checked by hand -- what autograd does for a saved tensor that was modified in place
        seg = cfg[5] if len(cfg) > 5 else None
        ctx.x2_version = seg[2]._version if (seg and len(seg) > 2) else None
        ctx.save_for_backward(x, saved, *tensors)
        # the output is the last block of `saved` (which the backward reads): the caller gets an alias, the ctx keeps the base
        out = saved[off:off + m.R * out_dim].view(m.R, out_dim).detach()
        if m.R1 > 0:                                   # two batches: one output (and one incoming gradient) per segment
            return out[:m.R1], out[m.R1:]
        return out

    @staticmethod
    def _backward_c(ctx, *dys):
        lib = _lib.load()
        x, saved = ctx.saved_tensors[:2]
        tensors = list(ctx.saved_tensors[2:])
        cfg = ctx.cfg
        dtype = cfg[3]
        if ctx.x2_version is not None and cfg[5][2]._version != ctx.x2_version:
            raise RuntimeError("MLPwithBN.forward_pair: the second batch (read in place, not copied) was modified in place between "
                               "the forward and the backward; keep it unchanged until the backward has run, or set "
                               "`inputs_in_place = False` on the module (one concatenated copy per call)")
        m, keep = BnMlpFn._c_desc(x, ctx.rz, cfg, ctx.bufs, tensors)
        dev = x.device
        if m.R1 > 0:                                   # the two segments' gradients -> the rows of one [R, out] operand (one launch)
            od = m.layer[m.nl - 1].out
            g1 = dys[0] if dys[0] is not None else torch.zeros(m.R1, od, dtype=torch.float32, device=dev)
            g2 = dys[1] if dys[1] is not None else torch.zeros(m.R - m.R1, od, dtype=torch.float32, device=dev)
            if g1.is_contiguous() and g2.is_contiguous() and g1.data_ptr() + 4 * m.R1 * od == g2.data_ptr() and \
                    g1.untyped_storage().data_ptr() == g2.untyped_storage().data_ptr():
                dy = torch.as_strided(g1, (m.R, od), (od, 1))      # the two gradients already ARE the rows of one array (monitor_step)
            else:
                dy = torch.cat([g1.reshape(m.R1, od), g2.reshape(m.R - m.R1, od)], 0)
        else:
            dy = dys[0].contiguous()
        g = _lib.BnMlpGrads()
        grads = [None] * len(tensors)

        def sink(i):
            t, acc = _gsink(tensors[i])
            grads[i] = _gret(t, acc)
            return t, acc
        (g0, a0), (b0, ab0) = sink(0), sink(1)
        if a0 != ab0:                                         # one accumulate flag per BatchNorm: both in place or neither
            g0, b0 = torch.empty_like(tensors[0]), torch.empty_like(tensors[1]); grads[0], grads[1] = g0, b0; a0 = False
        g.g_gamma0, g.g_beta0, g.acc0 = g0.data_ptr(), b0.data_ptr(), int(a0)
        keepg = [g0, b0]
        for i in range(ctx.nl):
            W, b, gw, gb = tensors[2 + 4 * i: 6 + 4 * i]
            gl = g.layer[i]
            t, acc = sink(2 + 4 * i); gl.g_w, gl.acc_w = t.data_ptr(), int(acc); keepg.append(t)
            if b is not None:
                t, acc = sink(3 + 4 * i); gl.g_b, gl.acc_b = t.data_ptr(), int(acc); keepg.append(t)
            (tg, ag), (tb, ab) = sink(4 + 4 * i), sink(5 + 4 * i)
            if ag != ab:
                tg, tb = torch.empty_like(gw), torch.empty_like(gb); grads[4 + 4 * i], grads[5 + 4 * i] = tg, tb; ag = False
            gl.g_gamma, gl.g_beta, gl.acc_bn = tg.data_ptr(), tb.data_ptr(), int(ag)
            keepg += [tg, tb]
        g.precision = ops.wgrad_precision(base_dtype(dtype) != torch.float32)
        pj = None
        in_place = all(g.layer[i].acc_w and (tensors[3 + 4 * i] is None or g.layer[i].acc_b) for i in range(ctx.nl))
        bn0_meta = None
        if ctx.rw is not None and ROLLOUT_WGRADS.enabled and in_place:
            scratch = ROLLOUT_WGRADS.scratch(ctx.rw[0], ctx.rw[1], lib.vln_bn_mlp_bwd_scratch_floats(m), dev)
            pj = _lib.ParamJobs()
            g.defer = C.pointer(pj)
            W0, b0_ = tensors[2], tensors[3]
            if _BN0_FROM_WGRAD[0] and not ctx.needs_input_grad[0] and cfg[0] and a0 and b0_ is not None and W0.is_contiguous() and \
                    W0.dtype == torch.float32:
                # the input carries no gradient: its BatchNorm's d gamma / d beta come from the first layer's rollout-level weight
                # gradient (RolloutWgrads.flush -> vln_bn0_grads_from_wgrad); this call skips dz W and the BatchNorm backward
                g.bn0_from_wgrad = 1
                bn0_meta = {"bn0": dict(W=W0.detach(), gamma=tensors[0].detach(), beta=tensors[1].detach(), gW=keepg[2], gb=keepg[3],
                                        gg=g0, gbeta=b0)}
        else:
            scratch = ops.empty(lib.vln_bn_mlp_bwd_scratch_floats(m), dtype=torch.float32, device=dev)
        g.scratch, g.scratch_floats = scratch.data_ptr(), scratch.numel()
        dx = ops.empty(x.shape[0], x.shape[1], dtype=torch.float32, device=dev) if ctx.needs_input_grad[0] else None
        ws = ops.workspace(dev, lib.vln_bn_mlp_ws_floats(m))
        rc = lib.vln_bn_mlp_bwd(m, x.data_ptr(), x.stride(0), saved.data_ptr(), dy.data_ptr(), dy.stride(0), _p(dx),
                                0 if dx is None else dx.stride(0), g, ws.data_ptr(), ws.numel(), _lib.raw_stream())
        if rc:
            _lib.check(rc, "vln_bn_mlp_bwd")
        if pj is not None:
            ROLLOUT_WGRADS.defer(ctx.rw[0], ctx.rw[1], pj, (saved, scratch, keep, keepg) + ((bn0_meta,) if bn0_meta else ()))
        return (dx, None, None, None) + tuple(grads)

    @staticmethod
    def forward(ctx, x, row_zero, cfg, bufs, *tensors):
        training, eps, momentum, dtype, drops = cfg[:5]
        if len(cfg) > 5 and cfg[5] is not None and not (_BN_MLP_C_CALL[0] and (len(tensors) - 2) // 4 <= _lib.BN_MLP_MAX_LAYERS):
            raise _lib.VlnError("MLPwithBN: the two-segment form needs the C-call form of the BN-MLP")
        lib = _lib.load()
        st_ = _lib.raw_stream()
        x = x.detach()
        if not x.is_contiguous():
            x = x.contiguous()
        R = x.shape[0]
        dev = x.device
        nl = (len(tensors) - 2) // 4
        rz = None
        if row_zero is not None:
            rz = row_zero.contiguous()
            rz = rz.view(torch.uint8) if rz.dtype == torch.bool else rz.to(torch.uint8)
        ctx.c_call = False
        if any(len(dr) > 3 for dr in drops) and not (_BN_MLP_C_CALL[0] and nl <= _lib.BN_MLP_MAX_LAYERS):
            raise _lib.VlnError("MLPwithBN: a DeviceClock needs the C-call form of the BN-MLP")
        if _BN_MLP_C_CALL[0] and nl <= _lib.BN_MLP_MAX_LAYERS and all(tensors[3 + 4 * i] is not None for i in range(nl)):
            return BnMlpFn._forward_c(ctx, x, rz, cfg, bufs, tensors)

        def bn(inp, w, b, buf, relu, drop, rzp):
            D = inp.shape[1]
            out = ops.empty(R, D, dtype=torch.float32, device=dev)
            stats = ops.empty(2, D, dtype=torch.float32, device=dev) if training else None
            rm, rv, nbt = buf
            p_, seed_, off_ = drop
            _wst, wsp, wsn = _bn_ws(R, D, dev)
            rc = lib.vln_bn_fwd(_p(inp), inp.stride(0), _p(out), out.stride(0), _p(w), _p(b), _p(rm), _p(rv), _p(nbt) if training else None,
                                _p(stats), None if stats is None else stats.data_ptr() + 4 * D, R, D, eps, momentum,
                                1 if training else 0, 1 if relu else 0, seed_, off_, p_ if training else 0.0, _p(rzp), wsp, wsn, st_)
            if rc:
                _lib.check(rc, "vln_bn_fwd")
            return out, stats

        saved = []
        y, s0 = bn(x, tensors[0], tensors[1], bufs[0], False, (0.0, 0, 0), None)
        saved += [x, s0, y]
        for i in range(nl):
            W, b, gw, gb = tensors[2 + 4 * i: 6 + 4 * i]
            # forward of an fp32-streamed layer in bf16 mode: the fp32-GRADE six-product form (a ReLU decision follows: csrc/bn_mlp.hip),
            # the three-product split only in the backward
            wd_ = wdtype(dtype, "mlp")
            z = ops.linear_fwd(y, SHADOWS.get(W, "n", wd_), None if b is None else b.detach(),
                               split="x6" if (wd_ == torch.float32 and base_dtype(dtype) != torch.float32) else False)
            last = i == nl - 1
            y, si = bn(z, gw, gb, bufs[1 + i], True, drops[i], rz if last else None)
            saved += [z, si, y]
        ctx.cfg, ctx.bufs, ctx.nl, ctx.rz = cfg, bufs, nl, rz
        ctx.save_for_backward(*[t for t in saved if t is not None], *tensors)
        ctx.layout = [t is not None for t in saved]
        return y

    @staticmethod
    def backward(ctx, *dys):
        if ctx.c_call:
            return BnMlpFn._backward_c(ctx, *dys)
        dy = dys[0]
        training, eps, momentum, dtype, drops = ctx.cfg[:5]
        lib = _lib.load()
        st_ = _lib.raw_stream()
        nl, rz, bufs = ctx.nl, ctx.rz, ctx.bufs
        it = iter(ctx.saved_tensors)
        saved = [next(it) if present else None for present in ctx.layout]
        tensors = list(it)
        x, s0, y0 = saved[0], saved[1], saved[2]
        R = x.shape[0]
        dev = x.device
        grads = [None] * len(tensors)
        wb = ops.WgradBatch(base_dtype(dtype) != torch.float32, never_plain=True)     # behind a BatchNorm: split operands (csrc/bn_mlp.hip)
        cb = ops.ColsumBatch()

        def bn_bwd(inp, dyy, yy, w, stats, buf, relu, drop, rzp, want_dx, gi):
            D = inp.shape[1]
            dx = ops.empty(R, D, dtype=torch.float32, device=dev) if want_dx else None
            # d gamma / d beta: straight into existing .grad tensors when grad-in-place is on (the kernel accumulates), else
            # fresh tensors for autograd's AccumulateGrad (two `add_` launches per BatchNorm per step)
            (gg, acc_g), (gbt, acc_b) = _gsink(tensors[gi]), _gsink(tensors[gi + 1])
            if acc_g and acc_b:
                pg, pb, acc = gg.data_ptr(), gbt.data_ptr(), 1
            else:
                dgb = torch.empty(2, D, dtype=torch.float32, device=dev)
                pg, pb, acc = dgb.data_ptr(), dgb.data_ptr() + 4 * D, 0
            p_, seed_, off_ = drop
            mean_p = stats.data_ptr() if training else buf[0].data_ptr()
            rstd_p = stats.data_ptr() + 4 * D if training else buf[1].data_ptr()
            _wst, wsp, wsn = _bn_ws(R, D, dev)
            rc = lib.vln_bn_bwd(_p(inp), inp.stride(0), _p(dyy), dyy.stride(0), _p(yy), 0 if yy is None else yy.stride(0), _p(w), mean_p,
                                rstd_p, _p(dx), 0 if dx is None else dx.stride(0), pg, pb, R, D, eps,
                                1 if training else 0, 1 if relu else 0, acc, seed_, off_, p_ if training else 0.0, _p(rzp), wsp, wsn, st_)
            if rc:
                _lib.check(rc, "vln_bn_bwd")
            if not acc:
                grads[gi], grads[gi + 1] = dgb[0], dgb[1]
            return dx

        g = dy.contiguous()
        for i in range(nl - 1, -1, -1):
            W, b, gw, gb = tensors[2 + 4 * i: 6 + 4 * i]
            z, si, y = saved[3 + 3 * i], saved[4 + 3 * i], saved[5 + 3 * i]
            y_prev = saved[2 + 3 * i]                    # input of this layer's Linear (y0 for i = 0)
            dz = bn_bwd(z, g, y, gw, si, bufs[1 + i], True, drops[i], rz if i == nl - 1 else None, True, 4 + 4 * i)
            dW, accW = _gsink(W)
            wb.add(dz, y_prev, dW, accW)
            grads[2 + 4 * i] = _gret(dW, accW)
            if b is not None:
                db, accb = _gsink(b)
                cb.add(dz, db, None, accb)
                grads[3 + 4 * i] = _gret(db, accb)
            g = _lin(dz, W, "t", dtype, "mlp")
        # Mt is the same for every product of the call: one grouped launch each for weights and biases
        wb.run(); cb.run()
        dx = bn_bwd(x, g, None, tensors[0], s0, bufs[0], False, (0.0, 0, 0), None, ctx.needs_input_grad[0], 0)
        return (dx, None, None, None) + tuple(grads)


_BN_MLP_C_CALL = [True]


def set_bn_mlp_c_call(on: bool):
    """A/B: BnMlpFn as one C call each way (vln_bn_mlp_fwd / bwd, the default) or its launches driven from Python."""
    _BN_MLP_C_CALL[0] = bool(on)


def bn_mlp(x, row_zero, training, eps, momentum, dtype, drops, bufs, tensors, seg=None):
    """seg = (R1, [dropout offset of segment 1 per layer]): rows [0, R1) and [R1, R) are two independent batches (vln_bn_mlp.R1);
    `row_zero` then covers segment 1's rows and the result is a pair (one output per segment).  seg = (R1, offsets, x2): `x` holds
    segment 0's R1 rows only and segment 1 is the contiguous-row matrix `x2`, read where it is (no concatenated copy; neither may
    require a gradient)."""
    cfg = (bool(training), float(eps), float(momentum), dtype, tuple(drops)) + (((int(seg[0]), tuple(seg[1])) + tuple(seg[2:3]),) if seg else ())
    return BnMlpFn.apply(x, row_zero, cfg, tuple(bufs), *tensors)


def _add_n(out, srcs):
    """out = sum of up to four [rows, cols] matrices (row strides free)."""
    rows, cols = out.shape
    a = [(None, 0)] * 4
    for i, t in enumerate(srcs):
        a[i] = (t.data_ptr(), t.stride(0))
    rc = _lib.load().vln_add_n(out.data_ptr(), out.stride(0), rows, cols, a[0][0], a[0][1], a[1][0], a[1][1], a[2][0], a[2][1],
                               a[3][0], a[3][1], 0, _lib.raw_stream())
    if rc:
        _lib.check(rc, "vln_add_n")
    return out


class MonitorCoreFn(torch.autograd.Function):
    """Everything of MonitorDecoder.forward (policy.py:132-166) after the BN-MLP, as ONE autograd node: positional encoding
    + dropout, the two attentions, the LSTM cell, the action logits and the progress monitor -- forward ~20 launches,
    backward ~25 (hand-derived chain; all six weight gradients in one grouped launch, all bias gradients in another)
    instead of ~35 operator nodes each way.  cfg = (training, dtype, p_pe, (seed_pe, off_pe), p_drop, seed, off_h1, off_mem).
    params = W_tin, W_vh, b_vh, W_ih, W_hh, b_ih, b_hh, W_a, b_a, W_m, b_m, W_c, b_c."""

    @staticmethod
    def forward(ctx, cfg, pe, ctx_mask, cand_mask, prev_rep, cand_rep, h0, c0, ctxt, *params):
        training, dtype, p_pe, (seed_pe, off_pe), p_drop, seed, off_h1, off_mem = cfg
        W_tin, W_vh, b_vh, W_ih, W_hh, b_ih, b_hh, W_a, b_a, W_m, b_m, W_c, b_c = params
        lib = _lib.load()
        st_ = _lib.raw_stream()
        f32 = torch.float32
        prev_rep, cand_rep = prev_rep.detach().contiguous(), cand_rep.detach().contiguous()
        h0, c0, ctxt = h0.detach().contiguous(), c0.detach().contiguous(), ctxt.detach().contiguous()
        B, C, M = cand_rep.shape
        H = h0.shape[1]
        L = ctxt.shape[1]
        dev = h0.device
        XK = 2 * M + 2 * H
        pp = p_pe if training else 0.0
        pd = p_drop if training else 0.0
        # positioned context (fresh dropout mask per step, units.py:205-207)
        pctx = ops.empty(B, L, H, dtype=f32, device=dev)
        rc = lib.vln_pe_dropout(_p(ctxt), _p(pe), _p(pctx), B, L, H, seed_pe, off_pe, pp, st_)
        if rc:
            _lib.check(rc, "vln_pe_dropout")
        xcat = ops.empty(B, XK, dtype=f32, device=dev)          # [prev_rep | moves | words | h0]: the LSTM input row
        tq = _lin(h0, W_tin, "n", dtype, "w_tin")
        _, word_w = ops.attn_fwd_rows(pctx, tq, ctx_mask, out=xcat[:, 2 * M:2 * M + H])
        vq = _lin(h0, W_vh, "n", dtype, "w_vh", b_vh.detach())
        _, move_w = ops.attn_fwd_rows(cand_rep, vq, cand_mask, out=xcat[:, M:2 * M])
        xcat[:, :M].copy_(prev_rep)
        xcat[:, 2 * M + H:].copy_(h0)
        # the gate product's split-K slabs go straight to the pointwise launch (as in the one-call step, csrc/monitor.hip)
        gates = ops.linear_fwd_slabs(xcat, _fused_lstm_weight(W_ih, W_hh, wdtype(dtype, "w_cat"), False), split=isinstance(dtype, tuple) and "w_cat" in dtype[1])
        h1, c1, act, tc, hd = ops.lstm_pointwise_fwd(gates, b_ih.detach(), b_hh.detach(), c0, seed, off_h1, pd, True)
        tcat = ops.empty(B, 2 * H, dtype=f32, device=dev)        # [words | drop(h1)]
        tcat[:, :H].copy_(xcat[:, 2 * M:2 * M + H])
        tcat[:, H:].copy_(hd)
        aq = _lin(tcat, W_a, "n", dtype, "w_a", b_a.detach())
        logit = ops.attn_dot(cand_rep, aq)
        hm = ops.empty(B, H + M, dtype=f32, device=dev)          # [h0 | moves]
        hm[:, :H].copy_(h0)
        hm[:, H:].copy_(xcat[:, M:2 * M])
        mg = _lin(hm, W_m, "n", dtype, "w_m", b_m.detach())
        mem = ops.empty(B, H, dtype=f32, device=dev)
        prog = ops.empty(B, dtype=f32, device=dev)
        wc = W_c.detach().reshape(-1)
        rc = lib.vln_monitor_head_fwd(_p(mg), _p(c1), _p(word_w), _p(wc), _p(b_c.detach()), _p(mem), _p(prog), B, L, H, seed, off_mem,
                                      pd, st_)
        if rc:
            _lib.check(rc, "vln_monitor_head_fwd")
        ctx.cfg = cfg
        ctx.save_for_backward(pctx, tq, vq, xcat, tcat, aq, hm, mg, mem, prog, word_w, move_w, act, tc, c0, c1, cand_rep, h0, *params)
        ctx.set_materialize_grads(False)
        return logit, prog, h1, c1, word_w, move_w

    @staticmethod
    def backward(ctx, dlogit, dprog, dh1, dc1, dww_ext, dmw_ext):
        training, dtype, p_pe, (seed_pe, off_pe), p_drop, seed, off_h1, off_mem = ctx.cfg
        (pctx, tq, vq, xcat, tcat, aq, hm, mg, mem, prog, word_w, move_w, act, tc, c0, c1, cand_rep, h0,
         W_tin, W_vh, b_vh, W_ih, W_hh, b_ih, b_hh, W_a, b_a, W_m, b_m, W_c, b_c) = ctx.saved_tensors
        lib = _lib.load()
        st_ = _lib.raw_stream()
        f32 = torch.float32
        B, C, M = cand_rep.shape
        H, L = h0.shape[1], pctx.shape[1]
        dev = h0.device
        pp = p_pe if training else 0.0
        pd = p_drop if training else 0.0
        cz = lambda t: None if t is None else t.contiguous()
        dlogit, dprog, dh1, dc1, dww_ext, dmw_ext = cz(dlogit), cz(dprog), cz(dh1), cz(dc1), cz(dww_ext), cz(dmw_ext)
        E = lambda *sh: ops.empty(*sh, dtype=f32, device=dev)
        # progress head (policy.py:126-130)
        dmg, dc1_t, dww, Z, dpre = E(B, H), E(B, H), E(B, L), E(B, L + H), E(B, 1)
        wc = W_c.detach().reshape(-1)
        rc = lib.vln_monitor_head_bwd(_p(mg), _p(c1), _p(word_w), _p(wc), _p(mem), _p(prog), _p(dprog), _p(dc1), _p(dww_ext), _p(dmg),
                                      _p(dc1_t), _p(dww), _p(Z), _p(dpre), B, L, H, seed, off_mem, pd, st_)
        if rc:
            _lib.check(rc, "vln_monitor_head_bwd")
        dhm = _lin(dmg, W_m, "t", dtype, "w_m")                          # [B, H+M] -> h0 | moves
        # action logits (policy.py:108-117): logit = cand_rep . aq
        if dlogit is None:
            dlogit = torch.zeros(B, C, dtype=f32, device=dev)
        daq = ops.rows_wsum(cand_rep, dlogit)
        dtcat = _lin(daq, W_a, "t", dtype, "w_a")                        # [B, 2H] -> words | drop(h1)
        # LSTM cell
        dg, dc0 = ops.lstm_pointwise_bwd(dh1, dtcat[:, H:].contiguous(), dc1_t, act, tc, c0, seed, off_h1, pd)
        dxcat = ops.linear_fwd(dg, _fused_lstm_weight(W_ih, W_hh, wdtype(dtype, "w_cat"), True), split=isinstance(dtype, tuple) and "w_cat" in dtype[1])          # [B, 2M+2H] -> prev | moves | words | h0
        dmoves = _add_n(E(B, M), [dhm[:, H:], dxcat[:, M:2 * M]])
        dwords = _add_n(E(B, H), [dtcat[:, :H], dxcat[:, 2 * M:2 * M + H]])
        # visual attention over the projected candidates; d cand_rep = move_w (x) dmoves + dl_v (x) vq + dlogit (x) aq
        dvq, dl_v = ops.attn_bwd_rows(cand_rep, move_w, dmoves, dmw_ext, want_dl=True)
        dcand = None
        if ctx.needs_input_grad[5]:
            dcand = torch.empty(B, C, M, dtype=f32, device=dev)
            ops.attn_dctx_deferred([move_w.data_ptr(), dlogit.data_ptr()], [dl_v.data_ptr(), None], [dmoves.data_ptr(), aq.data_ptr()], M,
                                   [vq.data_ptr(), None], M, dcand)
        dh0_v = _lin(dvq, W_vh, "t", dtype, "w_vh")
        # text attention over dropout(ctx + pe): d ctx = (word_w (x) dwords + dl_t (x) tq) * this step's mask
        dtq, dl_t = ops.attn_bwd_rows(pctx, word_w, dwords, dww, want_dl=True)
        dctx = None
        if ctx.needs_input_grad[8]:
            dctx = torch.empty(B, L, H, dtype=f32, device=dev)
            ops.attn_dctx_deferred([word_w.data_ptr()], [dl_t.data_ptr()], [dwords.data_ptr()], H, [tq.data_ptr()], H, dctx,
                                   drop=[(seed_pe, off_pe, pp)])
        dh0_t = _lin(dtq, W_tin, "t", dtype, "w_tin")
        dh0 = _add_n(E(B, H), [dhm[:, :H], dxcat[:, 2 * M + H:], dh0_v, dh0_t])
        # parameter gradients: six products over the same B rows -> one grouped launch; biases -> another
        sk = [_gsink(w) for w in (W_tin, W_vh, W_ih, W_hh, W_a, W_m)]
        gW = [t for t, _ in sk]
        wb = ops.WgradBatch(base_dtype(dtype) != f32)
        wb.add(dtq, h0, gW[0], sk[0][1]); wb.add(dvq, h0, gW[1], sk[1][1])
        wb.add(dg, xcat[:, :2 * M + H], gW[2], sk[2][1]); wb.add(dg, xcat[:, 2 * M + H:], gW[3], sk[3][1])
        wb.add(daq, tcat, gW[4], sk[4][1]); wb.add(dmg, hm, gW[5], sk[5][1])
        wb.run()
        (gb_vh, a_vh), (gb_a, a_a), (gb_m, a_m) = _gsink(b_vh), _gsink(b_a), _gsink(b_m)
        (gb_ih, a_ih), (gb_hh, a_hh) = _gsink(b_ih), _gsink(b_hh)
        gWc = torch.empty(L + H, dtype=f32, device=dev)
        cb = ops.ColsumBatch()
        cb.add(dvq, gb_vh, None, a_vh); cb.add(daq, gb_a, None, a_a); cb.add(dmg, gb_m, None, a_m); cb.add(Z, gWc)
        if a_ih == a_hh:
            cb.add(dg, gb_ih, gb_hh, a_ih)
        else:
            cb.add(dg, gb_ih, None, a_ih); cb.add(dg, gb_hh, None, a_hh)
        cb.run()
        gbc = dpre.sum(0)
        return (None, None, None, None, dxcat[:, :M], dcand, dh0, dc0, dctx,
                _gret(gW[0], sk[0][1]), _gret(gW[1], sk[1][1]), _gret(gb_vh, a_vh), _gret(gW[2], sk[2][1]), _gret(gW[3], sk[3][1]),
                _gret(gb_ih, a_ih), _gret(gb_hh, a_hh), _gret(gW[4], sk[4][1]), _gret(gb_a, a_a), _gret(gW[5], sk[5][1]), _gret(gb_m, a_m),
                gWc.view_as(W_c), gbc.view_as(b_c))


class FollowerCoreFn(torch.autograd.Function):
    """AttnDecoderLSTM.forward (policy.py:37-60) + ActionScoring (units.py:163-185) as ONE autograd node: the v-projected
    visual attention, dropout, the LSTM cell, the text attention with its tanh output layer and the candidate scores --
    no torch glue (concatenations are row blocks of one buffer, `context * target` is folded into the score query:
    logit = context . (target (.) w_out) + b_out), every weight gradient of the step in one grouped launch over the B
    rows, every bias gradient in another.  The two projections over many rows (36 views, C candidates) never form their
    [B*S, dot] gradient: dW_v = tq^T (sum_v dl_v img_v), dW_act = q^T (sum_c dlogit_c cand_c).
    cfg = (training, dtype, p_drop, seed, off)   (dropout sites off, off + 1 as on the operator path).
    params = W_h, b_h, W_v, b_v, W_ih, W_hh, b_ih, b_hh, W_tin, W_tout, W_act, b_act, W_hid, b_hid, w_out, b_out."""

    @staticmethod
    def forward(ctx, cfg, ctx_mask, img, a_prev, cands, h0, c0, ctxt, *params):
        training, dtype, p_drop, seed, off = cfg
        W_h, b_h, W_v, b_v, W_ih, W_hh, b_ih, b_hh, W_tin, W_tout, W_act, b_act, W_hid, b_hid, w_out, b_out = params
        f32 = torch.float32
        img, cands = img.detach().contiguous(), cands.detach().contiguous()
        a_prev, h0, c0, ctxt = a_prev.detach().contiguous(), h0.detach().contiguous(), c0.detach().contiguous(), ctxt.detach().contiguous()
        B, V, F = img.shape
        C, A = cands.shape[1], cands.shape[2]
        H = h0.shape[1]
        D = W_h.shape[0]
        dev = h0.device
        pd = p_drop if training else 0.0
        E = lambda *sh: ops.empty(*sh, dtype=f32, device=dev)
        # (1) panorama attention: logits_v = (W_v img_v + b_v) . tq, tq = W_h h0 + b_h, taken as img_v . (W_v^T tq) -- b_v . tq is one
        # constant per episode under the softmax, the [B * V, D] keys are never formed (csrc/follower.hip); weighted sum over the views
        tq = ops.linear_fwd(h0, SHADOWS.get(W_h, "n", dtype), b_h.detach())
        keys = ops.linear_fwd(tq, SHADOWS.get(W_v, "t", dtype))  # [B, F]: the projected query (the saved slot keeps its old name)
        vlog = ops.attn_dot(img, keys)
        xcat = E(B, A + F + H)                                   # [a_prev | pano | h0]: the LSTM input row
        _, view_w = ops.attn_softmax_wsum(img, vlog, None, out=xcat[:, A:A + F])
        xcat[:, :A].copy_(a_prev)
        xcat[:, A + F:].copy_(h0)
        if pd > 0:                                               # dropout over cat(a_prev, pano) (policy.py:49), in place
            st = _lib.load().vln_scale_dropout(xcat.data_ptr(), xcat.stride(0), xcat.data_ptr(), xcat.stride(0), B, A + F, seed, off, pd,
                                               None, _lib.raw_stream())
            if st:
                _lib.check(st, "vln_scale_dropout")
        # (2) LSTM cell
        gates = ops.linear_fwd_slabs(xcat, _fused_lstm_weight(W_ih, W_hh, dtype, False))       # slabs -> the pointwise launch (csrc/follower.hip)
        h1, c1, act, tc, hd = ops.lstm_pointwise_fwd(gates, b_ih.detach(), b_hh.detach(), c0, seed, off + 1, pd, True)
        # (3) text attention + tanh(W_out [wc ; drop(h1)])
        tq2 = ops.linear_fwd(hd, SHADOWS.get(W_tin, "n", dtype))
        tcat = E(B, 2 * H)
        _, word_w = ops.attn_fwd_rows(ctxt, tq2, ctx_mask, out=tcat[:, :H])
        tcat[:, H:].copy_(hd)
        grounded = ops.linear_fwd(tcat, SHADOWS.get(W_tout, "n", dtype), None, ops.ACT_TANH)
        # (4) candidate scores
        target = ops.linear_fwd(grounded, SHADOWS.get(W_hid, "n", dtype), b_hid.detach())
        wo = w_out.detach().reshape(-1)
        q = ops.ew(ops.EW_MUL, target, wo, bcast=True)
        context = ops.linear_fwd(cands.view(B * C, A), SHADOWS.get(W_act, "n", dtype), b_act.detach())
        raw = ops.attn_dot(context.view(B, C, D), q)
        logit = ops.ew(ops.EW_ADD_SCALAR, raw, b_out.detach(), bcast=True)
        ctx.cfg = cfg
        ctx.save_for_backward(img, cands, h0, c0, ctxt, tq, keys, view_w, xcat, act, tc, hd, tq2, tcat, word_w, grounded, target, q,
                              context, *params)
        ctx.set_materialize_grads(False)
        return logit, h1, c1, word_w, view_w

    @staticmethod
    def backward(ctx, dlogit, dh1, dc1, dww_ext, dvw_ext):
        training, dtype, p_drop, seed, off = ctx.cfg
        (img, cands, h0, c0, ctxt, tq, keys, view_w, xcat, act, tc, hd, tq2, tcat, word_w, grounded, target, q, context,
         W_h, b_h, W_v, b_v, W_ih, W_hh, b_ih, b_hh, W_tin, W_tout, W_act, b_act, W_hid, b_hid, w_out, b_out) = ctx.saved_tensors
        f32 = torch.float32
        B, V, F = img.shape
        C, A = cands.shape[1], cands.shape[2]
        H, D = h0.shape[1], W_h.shape[0]
        dev = h0.device
        pd = p_drop if training else 0.0
        cz = lambda t: None if t is None else t.contiguous()
        dlogit, dh1, dc1, dww_ext, dvw_ext = cz(dlogit), cz(dh1), cz(dc1), cz(dww_ext), cz(dvw_ext)
        E = lambda *sh: ops.empty(*sh, dtype=f32, device=dev)
        if dlogit is None:
            dlogit = torch.zeros(B, C, dtype=f32, device=dev)
        wo = w_out.detach().reshape(-1)
        # (4) scores: logit = context . q + b_out, q = target (.) w_out, context = W_act cands + b_act
        dq = ops.rows_wsum(context.view(B, C, D), dlogit)
        dtarget = ops.ew(ops.EW_MUL, dq, wo, bcast=True)
        Zo = ops.ew(ops.EW_MUL, dq, target)                                      # colsum -> d w_out
        rc = ops.rows_wsum(cands, dlogit)                                        # [B, A]: sum_c dlogit_c cand_c -> d W_act = q^T rc
        qs = ops.ew(ops.EW_MUL_ROWSUM, q, dlogit, nb=C)                          # colsum -> d b_act
        sl = ops.ew(ops.EW_MUL_ROWSUM, None, dlogit, nb=C)                       # [B,1] row sums; colsum -> d b_out
        dgr = ops.linear_fwd(dtarget, SHADOWS.get(W_hid, "t", dtype))
        # (3) grounded = tanh(W_out tcat)
        dz = ops.ew(ops.EW_TANH_GRAD, dgr, grounded)
        dtcat = ops.linear_fwd(dz, SHADOWS.get(W_tout, "t", dtype))              # [B, 2H] -> wc | drop(h1)
        need_dctx = ctx.needs_input_grad[7]
        dtq2, dl_t = ops.attn_bwd_rows(ctxt, word_w, dtcat[:, :H], dww_ext, want_dl=need_dctx)
        dctx = None
        if need_dctx:
            dctx = torch.empty_like(ctxt)
            ops.attn_dctx_deferred([word_w.data_ptr()], [dl_t.data_ptr()], [dtcat.data_ptr()], dtcat.stride(0), [tq2.data_ptr()], H, dctx)
        dhd = _add_n(E(B, H), [dtcat[:, H:], ops.linear_fwd(dtq2, SHADOWS.get(W_tin, "t", dtype))])
        # (2) LSTM cell
        dg, dc0 = ops.lstm_pointwise_bwd(dh1, dhd, dc1, act, tc, c0, seed, off + 1, pd)
        dxcat = ops.linear_fwd(dg, _fused_lstm_weight(W_ih, W_hh, dtype, True))  # [B, A+F+H] -> a_prev | pano | h0
        if pd > 0:
            st = _lib.load().vln_scale_dropout(dxcat.data_ptr(), dxcat.stride(0), dxcat.data_ptr(), dxcat.stride(0), B, A + F, seed, off, pd,
                                               None, _lib.raw_stream())
            if st:
                _lib.check(st, "vln_scale_dropout")
        # (1) panorama attention: pano = sum_v alpha_v img_v, alpha = softmax(keys . tq)
        dpano = dxcat[:, A:A + F]
        dalpha = ops.attn_dot(img, dpano)
        _, dl_v = ops.attn_bwd(img, view_w, dalpha, dvw_ext, None, None, None, want_dl=True)
        rv = ops.rows_wsum(img, dl_v)                                            # [B, F]: sum_v dl_v img_v = d(W_v^T tq) -> d W_v = tq^T rv
        dtq = ops.linear_fwd(rv, SHADOWS.get(W_v, "n", dtype))                   # d tq = rv W_v^T
        tqs = torch.zeros(B, D, dtype=f32, device=dev)                           # d b_v is exactly 0 (b_v . tq shifts an episode's logits alike)
        dh0 = _add_n(E(B, H), [dxcat[:, A + F:], ops.linear_fwd(dtq, SHADOWS.get(W_h, "t", dtype))])
        # parameter gradients: eight products over the same B rows -> one grouped launch; the biases -> another
        sk = [_gsink(w) for w in (W_h, W_v, W_ih, W_hh, W_tin, W_tout, W_act, W_hid)]
        gW = [t for t, _ in sk]
        wb = ops.WgradBatch(dtype != f32)
        wb.add(dtq, h0, gW[0], sk[0][1]); wb.add(tq, rv, gW[1], sk[1][1])
        wb.add(dg, xcat[:, :A + F], gW[2], sk[2][1]); wb.add(dg, xcat[:, A + F:], gW[3], sk[3][1])
        wb.add(dtq2, hd, gW[4], sk[4][1]); wb.add(dz, tcat, gW[5], sk[5][1])
        wb.add(q, rc, gW[6], sk[6][1]); wb.add(dtarget, grounded, gW[7], sk[7][1])
        wb.run()
        bk = [_gsink(b) for b in (b_h, b_v, b_ih, b_hh, b_act, b_hid)]
        gb_h, gb_v, gb_ih, gb_hh, gb_act, gb_hid = (t for t, _ in bk)
        gwo = torch.empty(D, dtype=f32, device=dev)
        gbo = torch.empty(1, dtype=f32, device=dev)
        cb = ops.ColsumBatch()
        cb.add(dtq, gb_h, None, bk[0][1]); cb.add(tqs, gb_v, None, bk[1][1]); cb.add(qs, gb_act, None, bk[4][1])
        cb.add(dtarget, gb_hid, None, bk[5][1]); cb.add(Zo, gwo)
        if bk[2][1] == bk[3][1]:
            cb.add(dg, gb_ih, gb_hh, bk[2][1])
        else:
            cb.add(dg, gb_ih, None, bk[2][1]); cb.add(dg, gb_hh, None, bk[3][1])
        cb.add(sl, gbo)
        cb.run()
        da = dxcat[:, :A] if ctx.needs_input_grad[3] else None
        return (None, None, None, da, None, dh0, dc0, dctx,
                _gret(gW[0], sk[0][1]), _gret(gb_h, bk[0][1]), _gret(gW[1], sk[1][1]), _gret(gb_v, bk[1][1]), _gret(gW[2], sk[2][1]),
                _gret(gW[3], sk[3][1]), _gret(gb_ih, bk[2][1]), _gret(gb_hh, bk[3][1]), _gret(gW[4], sk[4][1]), _gret(gW[5], sk[5][1]),
                _gret(gW[6], sk[6][1]), _gret(gb_act, bk[4][1]), _gret(gW[7], sk[7][1]), _gret(gb_hid, bk[5][1]),
                gwo.view_as(w_out), gbo.view_as(b_out))
