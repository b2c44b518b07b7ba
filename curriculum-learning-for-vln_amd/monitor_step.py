"""The Self-Monitor decoder step after its BN-MLP (reference policy.py:132-166) as ONE C call each way
(`vln_monitor_step_fwd` / `vln_monitor_step_bwd`, csrc/monitor.hip): the ~20 forward and ~25 backward launches that
`functional.MonitorCoreFn` drives from Python, one ctypes call each, are issued by the library here.  Same numbers (it is the
same launch sequence); at B = 128 the agent is bound by the GPU instead of by the Python of its steps.

`MonitorStepFn.apply` takes MonitorCoreFn's arguments.  Parameter gradients follow `functional.set_grad_in_place`: added into an
existing contiguous `p.grad` inside the grouped launches (autograd gets None), or returned to autograd.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib, ops
from .functional import ROLLOUT_WGRADS, SHADOWS, _fused_lstm_weight, _gret, _gsink, base_dtype, wdtype

_p = ops._p

# ---- the rollout's context gradient in ONE buffer (round 5) ----------------------------------------------------------------------
# Every decoder step of a rollout attends the SAME encoder context, so autograd received T [B,L,H] gradients per rollout (21 MB each
# at B 128 / L 80 / H 512) and summed them with T - 1 element-wise adds.  With the context passed through `gated_ctx` the steps'
# backward calls hand autograd None and only REPORT their term (`vln_dctx_term`: the addresses of alpha, dl, g, q and the step's
# pe-dropout site); the identity node `runtime.CtxGate` -- whose backward runs once every step that consumed its output has run --
# forms all T terms with ONE launch and returns the buffer.  (`DEFER_DCTX[0] = False`: the steps add their terms to the rollout's
# buffer in place, one launch each -- the C steps' `dctx_accumulate`.)
_CTX_ENTRIES = {}


def gated_ctx(ctx_t):
    """-> (the tensor to hand to the step node, its runtime.CtxEntry or None when no gradient is wanted)."""
    from .runtime import CtxEntry, CtxGate
    if not (torch.is_grad_enabled() and ctx_t.requires_grad):
        return ctx_t, None
    k = id(ctx_t)
    e = _CTX_ENTRIES.get(k)
    if e is None or e.ref() is not ctx_t:
        if len(_CTX_ENTRIES) > 16:
            for kk in [kk for kk, x in _CTX_ENTRIES.items() if x.ref() is None]:
                del _CTX_ENTRIES[kk]
        e = _CTX_ENTRIES[k] = CtxEntry(ctx_t)
    if e.gated is None:
        e.gated = CtxGate.apply(e, ctx_t)
        e.dctx = None
        import weakref
        e.gated._vln_ctx_entry = weakref.ref(e)          # how the step nodes find the rollout's accumulator
        for a in ("_vln_lp", "_vln_born"):               # attributes the encoder hangs on its output (bf16 stream copy, arena stamp)
            if hasattr(ctx_t, a):
                setattr(e.gated, a, getattr(ctx_t, a))
    return e.gated, e


DEFER_DCTX = [True]        # A/B: False = every step adds its own term to the rollout's buffer (one launch per step)


def _dctx_target(ctx, entry_ref, want, B, L, H, dev):
    """-> (buffer the step's backward writes or None, accumulate flag, what the node returns to autograd for the context, the
    rollout's CtxEntry when the step's term is only REPORTED (vln_dctx_term) and formed once per rollout by runtime.CtxGate)."""
    if not want:
        return None, 0, None, None
    e = entry_ref() if entry_ref is not None else None
    if e is None:
        t = torch.empty(B, L, H, dtype=torch.float32, device=dev)
        return t, 0, t, None
    if DEFER_DCTX[0] and (e.kctx is None or e.kctx is False):
        return None, 0, None, e
    first = e.dctx is None
    if first:
        e.dctx = ops.empty(B, L, H, dtype=torch.float32, device=dev)
    return e.dctx, 0 if first else 1, None, None


def _report_term(e, term, shape, drop, keep):
    """One step's (alpha, dl, g, q) addresses onto the rollout's list (runtime.CtxGate forms them in one launch)."""
    if e.tdesc is None:
        e.tdesc = [int(term.ldg), int(term.ldq), []]
    elif e.tdesc[0] != term.ldg or e.tdesc[1] != term.ldq:
        raise _lib.VlnError("deferred context gradient: the steps of one rollout disagree on their layouts")
    e.shape = shape
    e.tdesc[2].append(drop)
    e.terms.append((term.alpha, term.dl, term.g, term.q, keep[0], keep))


def _m8(mask):
    if mask.dtype == torch.bool and mask.is_contiguous():
        return mask.view(torch.uint8)
    return mask.to(torch.uint8).contiguous()


def _r64(n):
    return (n + 63) & ~63


class MonitorStepFn(torch.autograd.Function):
    """cfg = (training, dtype, p_pe, (seed_pe, off_pe), p_drop, seed, off_h1, off_mem);
    params = W_tin, W_vh, b_vh, W_ih, W_hh, b_ih, b_hh, W_a, b_a, W_m, b_m, W_c, b_c."""

    @staticmethod
    def forward(ctx, cfg, pe, ctx_mask, cand_mask, prev_rep, cand_rep, h0, c0, ctxt, *params):
        ctx_arg = ctxt
        training, dtype, p_pe, (seed_pe, off_pe), p_drop, seed, off_h1, off_mem = cfg[:8]
        W_tin, W_vh, b_vh, W_ih, W_hh, b_ih, b_hh, W_a, b_a, W_m, b_m, W_c, b_c = params
        lib = _lib.load()
        f32 = torch.float32
        prev_rep, cand_rep = prev_rep.detach().contiguous(), cand_rep.detach().contiguous()
        h0, c0, ctxt = h0.detach().contiguous(), c0.detach().contiguous(), ctxt.detach().contiguous()
        B, Cn, M = cand_rep.shape
        H, L = h0.shape[1], ctxt.shape[1]
        dev = h0.device
        XK = 2 * M + 2 * H
        base = base_dtype(dtype)                     # `dtype` may carry the per-matrix fp32 override (functional.wdtype)
        wt = ops.F32 if base == f32 else ops.BF16
        d = _lib.MonitorDims(B, L, Cn, H, M, wt)
        w = _lib.MonitorWeights()
        hold = []                                    # streamed weight copies: kept alive until the launches are queued
        w.f32_mask = 0
        for bit, name in enumerate(("w_tin", "w_vh", "w_cat", "w_a", "w_m")):
            if base != f32 and wdtype(dtype, name) == f32:
                w.f32_mask |= 1 << bit

        def sh(W, kind, name):
            t = SHADOWS.get(W, kind, wdtype(dtype, name))
            hold.append(t)
            return t.data_ptr()

        w.w_tin, w.w_tin_t = sh(W_tin, "n", "w_tin"), sh(W_tin, "t", "w_tin")
        w.w_vh, w.w_vh_t, w.b_vh = sh(W_vh, "n", "w_vh"), sh(W_vh, "t", "w_vh"), b_vh.data_ptr()
        dcat = wdtype(dtype, "w_cat")
        wc_n, wc_t = _fused_lstm_weight(W_ih, W_hh, dcat, False), _fused_lstm_weight(W_ih, W_hh, dcat, True)
        hold += [wc_n, wc_t]
        w.w_cat, w.w_cat_t, w.b_ih, w.b_hh = wc_n.data_ptr(), wc_t.data_ptr(), b_ih.data_ptr(), b_hh.data_ptr()
        w.w_a, w.w_a_t, w.b_a = sh(W_a, "n", "w_a"), sh(W_a, "t", "w_a"), b_a.data_ptr()
        w.w_m, w.w_m_t, w.b_m = sh(W_m, "n", "w_m"), sh(W_m, "t", "w_m"), b_m.data_ptr()
        wcf = W_c.detach().reshape(-1)
        pe_c = pe if pe.is_contiguous() else pe.contiguous()
        hold += [wcf, pe_c]
        w.w_c, w.b_c, w.pe = wcf.data_ptr(), b_c.data_ptr(), pe_c.data_ptr()
        # outputs (returned) and the step's saved activations (one flat allocation)
        logit = ops.empty(B, Cn, dtype=f32, device=dev); prog = ops.empty(B, dtype=f32, device=dev)
        h1 = ops.empty(B, H, dtype=f32, device=dev); c1 = ops.empty(B, H, dtype=f32, device=dev)
        word_w = ops.empty(B, L, dtype=f32, device=dev); move_w = ops.empty(B, Cn, dtype=f32, device=dev)
        sizes = (("pctx", B * L * H), ("tq", B * H), ("vq", B * M), ("xcat", B * XK), ("tcat", B * 2 * H), ("aq", B * M),
                 ("hm", B * (H + M)), ("mg", B * H), ("mem", B * H), ("act", B * 4 * H), ("tanh_c1", B * H),
                 ("gates", B * 4 * H), ("dots", B * max(L, Cn)))
        ctx.rw = None
        if ROLLOUT_WGRADS.active(ctx) and training:          # the saved block as a slot of the rollout's arena (functional.RolloutWgrads)
            key = ("monitor", B, id(W_ih))
            flat, slot = ROLLOUT_WGRADS.saved(key, sum(_r64(n) for _, n in sizes), dev)
            ctx.rw = (key, slot)
        else:
            flat = ops.empty(sum(_r64(n) for _, n in sizes), dtype=f32, device=dev)
        io = _lib.MonitorStep()
        q = flat.data_ptr()
        for name, n in sizes:
            setattr(io, name, q)
            q += 4 * _r64(n)
        m_ctx, m_cand = _m8(ctx_mask), _m8(cand_mask)
        io.prev_rep, io.cand_rep, io.h0, io.c0, io.ctx = prev_rep.data_ptr(), cand_rep.data_ptr(), h0.data_ptr(), c0.data_ptr(), ctxt.data_ptr()
        io.ctx_mask, io.cand_mask = m_ctx.data_ptr(), m_cand.data_ptr()
        io.logit, io.prog, io.h1, io.c1 = logit.data_ptr(), prog.data_ptr(), h1.data_ptr(), c1.data_ptr()
        io.word_w, io.move_w = word_w.data_ptr(), move_w.data_ptr()
        ws = ops.workspace(dev, int(lib.vln_monitor_ws_floats(C.byref(d))))       # room for the slabs the consumers sum themselves
        io.ws, io.ws_floats = ws.data_ptr(), ws.numel()
        io.seed_pe, io.off_pe, io.p_pe = seed_pe, off_pe, (p_pe if training else 0.0)
        io.seed, io.off_h1, io.off_mem, io.p_drop = seed, off_h1, off_mem, (p_drop if training else 0.0)
        io.offset_base_dev = cfg[8] if len(cfg) > 8 else None        # runtime.DeviceClock word (offsets relative to it)
        st = lib.vln_monitor_step_fwd(C.byref(d), C.byref(w), C.byref(io), _lib.raw_stream())
        if st:
            _lib.check(st, "vln_monitor_step_fwd")
        ctx.cfg = cfg
        ctx.dentry = getattr(ctx_arg, "_vln_ctx_entry", None)          # weakref to the rollout's CtxEntry (gated_ctx), or None
        # the backward reads c1, prog and the two attention maps: their memory is kept alive HERE, and the caller gets fresh
        # aliases -- the returned objects carry this node as grad_fn, holding THEM on ctx would be a reference cycle
        ctx.pack = (d, w, io, hold, (m_ctx, m_cand), (prog, c1, word_w, move_w))
        ctx.save_for_backward(flat, prev_rep, cand_rep, h0, c0, ctxt, *params)
        ctx.set_materialize_grads(False)
        return logit, prog.detach(), h1, c1.detach(), word_w.detach(), move_w.detach()

    @staticmethod
    def backward(ctx, dlogit, dprog, dh1, dc1, dww_ext, dmw_ext):
        d, w, io, hold, masks, outs = ctx.pack
        flat, prev_rep, cand_rep, h0, c0, ctxt, *params = ctx.saved_tensors
        W_tin, W_vh, b_vh, W_ih, W_hh, b_ih, b_hh, W_a, b_a, W_m, b_m, W_c, b_c = params
        dtype = ctx.cfg[1]
        lib = _lib.load()
        f32 = torch.float32
        B, Cn, M = cand_rep.shape
        H, L = h0.shape[1], ctxt.shape[1]
        dev = h0.device
        cz = lambda t: None if t is None else t.contiguous()
        ups = [cz(t) for t in (dlogit, dprog, dh1, dc1, dww_ext, dmw_ext)]
        g = _lib.MonitorGrads()
        g.dlogit, g.dprog, g.dh1, g.dc1, g.dww_ext, g.dmw_ext = (_p(t) for t in ups)
        # d prev_rep and d cand_rep as the rows of ONE array: the two-batch BN-MLP in front of the step takes them as its single
        # [B + B*C, M] incoming gradient without a concatenating copy (functional.BnMlpFn._backward_c)
        if ctx.needs_input_grad[5]:
            both = ops.empty(B * (Cn + 1), M, dtype=f32, device=dev)
            dprev, dcand = both[:B], both[B:].view(B, Cn, M)
        else:
            dprev, dcand = ops.empty(B, M, dtype=f32, device=dev), None
        dh0 = ops.empty(B, H, dtype=f32, device=dev); dc0 = ops.empty(B, H, dtype=f32, device=dev)
        dctx_buf, dctx_acc, dctx, dentry = _dctx_target(ctx, ctx.dentry, ctx.needs_input_grad[8], B, L, H, dev)
        g.dprev_rep, g.dcand_rep, g.dh0, g.dc0, g.dctx = dprev.data_ptr(), _p(dcand), dh0.data_ptr(), dc0.data_ptr(), _p(dctx_buf)
        g.dctx_accumulate = dctx_acc
        term = None
        if dentry is not None:
            term = _lib.DctxTerm()
            g.dctx_term = C.pointer(term)
        sinks = [_gsink(p) for p in params]
        names = ("g_tin", "g_vh", "g_bvh", "g_ih", "g_hh", "g_bih", "g_bhh", "g_a", "g_ba", "g_m", "g_bm", "g_wc", "g_bc")
        for i, (n, (t, acc)) in enumerate(zip(names, sinks)):
            setattr(g, n, t.data_ptr())
            g.acc[i] = 1 if acc else 0
        g.precision = ops.wgrad_precision(base_dtype(dtype) != f32)
        ns = int(lib.vln_monitor_bwd_scratch_floats(C.byref(d)))
        pj = None
        if ctx.rw is not None and ROLLOUT_WGRADS.enabled and all(acc for _, acc in sinks):
            scratch = ROLLOUT_WGRADS.scratch(ctx.rw[0], ctx.rw[1], ns, dev)
            pj = _lib.ParamJobs()
            g.defer = C.pointer(pj)
        else:
            scratch = ops.empty(ns, dtype=f32, device=dev)
        g.scratch, g.scratch_floats = scratch.data_ptr(), ns
        ws = ops.workspace(dev, int(lib.vln_monitor_ws_floats(C.byref(d))))    # (the forward's pointer may belong to another stream's workspace)
        io.ws, io.ws_floats = ws.data_ptr(), ws.numel()
        st = lib.vln_monitor_step_bwd(C.byref(d), C.byref(w), C.byref(io), C.byref(g), _lib.raw_stream())
        if st:
            _lib.check(st, "vln_monitor_step_bwd")
        if pj is not None:
            ROLLOUT_WGRADS.defer(ctx.rw[0], ctx.rw[1], pj, (flat, scratch, hold, [t for t, _ in sinks]))
        if term is not None:
            cfg = ctx.cfg
            drop = (int(term.seed), int(term.offset), float(term.p)) + ((cfg[8],) if len(cfg) > 8 and cfg[8] is not None else ())
            _report_term(dentry, term, (B, L, H), drop if term.p > 0 else None, (scratch, flat, outs))
        ctx.pack = None
        return (None, None, None, None, dprev, dcand, dh0, dc0, dctx) + tuple(_gret(t, acc) for t, acc in sinks)


class FollowerStepFn(torch.autograd.Function):
    """AttnDecoderLSTM.forward + ActionScoring (policy.py:37-60, units.py:163-185) as ONE C call each way
    (`vln_follower_step_fwd/bwd`, csrc/follower.hip); arguments of functional.FollowerCoreFn:
    cfg = (training, dtype, p_drop, seed, off); params = W_h, b_h, W_v, b_v, W_ih, W_hh, b_ih, b_hh, W_tin, W_tout, W_act, b_act,
    W_hid, b_hid, w_out, b_out."""

    @staticmethod
    def forward(ctx, cfg, ctx_mask, img, a_prev, cands, h0, c0, ctxt, *params):
        ctx_arg = ctxt
        training, dtype, p_drop, seed, off = cfg[:5]
        W_h, b_h, W_v, b_v, W_ih, W_hh, b_ih, b_hh, W_tin, W_tout, W_act, b_act, W_hid, b_hid, w_out, b_out = params
        lib = _lib.load()
        f32 = torch.float32
        img, cands = img.detach().contiguous(), cands.detach().contiguous()
        a_prev, h0, c0, ctxt = a_prev.detach().contiguous(), h0.detach().contiguous(), c0.detach().contiguous(), ctxt.detach().contiguous()
        B, V, F = img.shape
        Cn, A = cands.shape[1], cands.shape[2]
        H, L, D = h0.shape[1], ctxt.shape[1], W_h.shape[0]
        dev = h0.device
        XK = A + F + H
        wt = ops.F32 if dtype == f32 else ops.BF16
        d = _lib.FollowerDims(B, L, V, Cn, H, F, A, D, wt)
        w = _lib.FollowerWeights()
        hold = []

        def sh(W, kind):
            t = SHADOWS.get(W, kind, dtype)
            hold.append(t)
            return t.data_ptr()

        w.w_h, w.w_h_t, w.b_h = sh(W_h, "n"), sh(W_h, "t"), b_h.data_ptr()
        w.w_v, w.w_v_t, w.b_v = sh(W_v, "n"), sh(W_v, "t"), b_v.data_ptr()
        wc_n, wc_t = _fused_lstm_weight(W_ih, W_hh, dtype, False), _fused_lstm_weight(W_ih, W_hh, dtype, True)
        wo = w_out.detach().reshape(-1)
        hold += [wc_n, wc_t, wo]
        w.w_cat, w.w_cat_t, w.b_ih, w.b_hh = wc_n.data_ptr(), wc_t.data_ptr(), b_ih.data_ptr(), b_hh.data_ptr()
        w.w_tin, w.w_tin_t = sh(W_tin, "n"), sh(W_tin, "t")
        w.w_tout, w.w_tout_t = sh(W_tout, "n"), sh(W_tout, "t")
        w.w_act, w.b_act = sh(W_act, "n"), b_act.data_ptr()
        w.w_hid, w.w_hid_t, w.b_hid = sh(W_hid, "n"), sh(W_hid, "t"), b_hid.data_ptr()
        w.w_out, w.b_out = wo.data_ptr(), b_out.data_ptr()
        logit = ops.empty(B, Cn, dtype=f32, device=dev)
        h1 = ops.empty(B, H, dtype=f32, device=dev); c1 = ops.empty(B, H, dtype=f32, device=dev)
        word_w = ops.empty(B, L, dtype=f32, device=dev); view_w = ops.empty(B, V, dtype=f32, device=dev)
        # ("keys": the projected query W_v^T tq [B, F] since ABI 16 -- the [B * V, D] keys are never formed)
        sizes = (("tq", B * D), ("keys", B * F), ("vlog", 0), ("xcat", B * XK), ("act", B * 4 * H), ("tanh_c1", B * H),
                 ("tq2", B * H), ("tcat", B * 2 * H), ("grounded", B * H), ("target", B * D), ("q", B * D), ("context", B * Cn * D),
                 ("gates", B * 4 * H), ("dots", B * max(L, V, Cn)))
        ctx.rw = None
        if ROLLOUT_WGRADS.active(ctx) and training:
            key = ("follower", B, id(W_ih))
            flat, slot = ROLLOUT_WGRADS.saved(key, sum(_r64(n) for _, n in sizes), dev)
            ctx.rw = (key, slot)
        else:
            flat = ops.empty(sum(_r64(n) for _, n in sizes), dtype=f32, device=dev)
        io = _lib.FollowerStep()
        q = flat.data_ptr()
        for name, n in sizes:
            setattr(io, name, q)
            q += 4 * _r64(n)
        m_ctx = _m8(ctx_mask) if ctx_mask is not None else None
        io.img, io.a_prev, io.cands, io.h0, io.c0, io.ctx = (img.data_ptr(), a_prev.data_ptr(), cands.data_ptr(), h0.data_ptr(),
                                                           c0.data_ptr(), ctxt.data_ptr())
        io.ctx_mask = _p(m_ctx)
        io.logit, io.h1, io.c1, io.word_w, io.view_w = logit.data_ptr(), h1.data_ptr(), c1.data_ptr(), word_w.data_ptr(), view_w.data_ptr()
        ws = ops.workspace(dev, 1 << 22)
        io.ws, io.ws_floats = ws.data_ptr(), ws.numel()
        io.seed, io.off, io.p_drop = seed, off, (p_drop if training else 0.0)
        io.offset_base_dev = cfg[5] if len(cfg) > 5 else None
        sync = cfg[6] if len(cfg) > 6 else None
        if sync is not None:
            io.attn_sync, io.attn_sync_bytes = sync.data_ptr(), sync.numel() * 4
            hold.append(sync)
        pre = cfg[7] if len(cfg) > 7 else None           # W_act cands + b_act formed up front (AttnDecoderLSTM.project_candidates)
        if pre is not None:
            if tuple(pre.shape) != (B * Cn, D) or not pre.is_contiguous() or pre.dtype != f32:
                raise _lib.VlnError(f"FollowerStepFn: the projected candidates must be a contiguous fp32 [{B * Cn}, {D}] matrix")
            io.context, io.context_ready = pre.data_ptr(), 1
            hold.append(pre)
        st = lib.vln_follower_step_fwd(C.byref(d), C.byref(w), C.byref(io), _lib.raw_stream())
        if st:
            _lib.check(st, "vln_follower_step_fwd")
        ctx.cfg = cfg
        ctx.dentry = getattr(ctx_arg, "_vln_ctx_entry", None)
        # the backward reads both attention maps: memory kept alive here, fresh aliases returned (no ctx <-> output cycle)
        ctx.pack = (d, w, io, hold, m_ctx, (word_w, view_w))
        ctx.save_for_backward(flat, img, cands, a_prev, h0, c0, ctxt, *params)
        ctx.set_materialize_grads(False)
        return logit, h1, c1, word_w.detach(), view_w.detach()

    @staticmethod
    def backward(ctx, dlogit, dh1, dc1, dww_ext, dvw_ext):
        d, w, io, hold, m_ctx, _alive = ctx.pack
        flat, img, cands, a_prev, h0, c0, ctxt, *params = ctx.saved_tensors
        dtype = ctx.cfg[1]
        lib = _lib.load()
        f32 = torch.float32
        B, A = a_prev.shape
        H, L = h0.shape[1], ctxt.shape[1]
        dev = h0.device
        cz = lambda t: None if t is None else t.contiguous()
        ups = [cz(t) for t in (dlogit, dh1, dc1, dww_ext, dvw_ext)]
        g = _lib.FollowerGrads()
        g.dlogit, g.dh1, g.dc1, g.dww_ext, g.dvw_ext = (_p(t) for t in ups)
        da = ops.empty(B, A, dtype=f32, device=dev) if ctx.needs_input_grad[3] else None
        dh0 = ops.empty(B, H, dtype=f32, device=dev); dc0 = ops.empty(B, H, dtype=f32, device=dev)
        dctx_buf, dctx_acc, dctx, dentry = _dctx_target(ctx, ctx.dentry, ctx.needs_input_grad[7], B, L, H, dev)
        g.da_prev, g.dh0, g.dc0, g.dctx = _p(da), dh0.data_ptr(), dc0.data_ptr(), _p(dctx_buf)
        g.dctx_accumulate = dctx_acc
        term = None
        if dentry is not None:
            term = _lib.DctxTerm()
            g.dctx_term = C.pointer(term)
        sinks = [_gsink(p) for p in params]
        names = ("g_wh", "g_bh", "g_wv", "g_bv", "g_ih", "g_hh", "g_bih", "g_bhh", "g_tin", "g_tout", "g_wact", "g_bact", "g_whid", "g_bhid",
                 "g_wout", "g_bout")
        for i, (n, (t, acc)) in enumerate(zip(names, sinks)):
            setattr(g, n, t.data_ptr())
            g.acc[i] = 1 if acc else 0
        g.precision = ops.wgrad_precision(dtype != f32)
        ns = int(lib.vln_follower_bwd_scratch_floats(C.byref(d)))
        pj = None
        if ctx.rw is not None and ROLLOUT_WGRADS.enabled and all(acc for _, acc in sinks):
            scratch = ROLLOUT_WGRADS.scratch(ctx.rw[0], ctx.rw[1], ns, dev)
            pj = _lib.ParamJobs()
            g.defer = C.pointer(pj)
        else:
            scratch = ops.empty(ns, dtype=f32, device=dev)
        g.scratch, g.scratch_floats = scratch.data_ptr(), ns
        ws = ops.workspace(dev, 1 << 22)
        io.ws, io.ws_floats = ws.data_ptr(), ws.numel()
        st = lib.vln_follower_step_bwd(C.byref(d), C.byref(w), C.byref(io), C.byref(g), _lib.raw_stream())
        if st:
            _lib.check(st, "vln_follower_step_bwd")
        if pj is not None:
            ROLLOUT_WGRADS.defer(ctx.rw[0], ctx.rw[1], pj, (flat, scratch, hold, [t for t, _ in sinks]))
        if term is not None:
            _report_term(dentry, term, (B, L, H), None, (scratch, flat, _alive))
        ctx.pack = None
        return (None, None, None, da, None, dh0, dc0, dctx) + tuple(_gret(t, acc) for t, acc in sinks)
