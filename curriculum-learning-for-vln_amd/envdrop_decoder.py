"""Drop-in `EnvDropDecoder` and `Critic` (reference: src/model/policy.py:173-267).

Same constructor arguments, `forward` signature / return tuple and `state_dict`
keys as the reference modules; the math runs in the gfx950 HIP kernels through
the C ABI (`vln_envdrop_step_fwd/bwd`), one C call per decoder step and one per
step backward.  Weight gradients are deferred (see runtime.py).
"""
from __future__ import annotations

import contextlib
import ctypes as C
import weakref
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from . import _lib, ops
from .runtime import GatedModuleMixin


class _SoftDotParams(nn.Module):
    """Parameter holder with the reference's SoftDotAttention key names (units.py:83-96)."""

    def __init__(self, query_dim, context_only=False, context_dim=None):
        super().__init__()
        ctx_dim = query_dim if context_dim is None else context_dim
        self.context_only = context_only
        self.linear_in = nn.Linear(query_dim, ctx_dim, bias=False)
        if not context_only:
            self.linear_out = nn.Linear(query_dim + ctx_dim, query_dim, bias=False)


class _StepRec:
    __slots__ = ("io", "dims", "slot", "keep", "ctx_owner", "entry", "B", "L", "C")


class _EnvDropStepFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, rec, h_tilde_prev, c0, ctx_t, *gated):
        lib = _lib.load()
        _lib.check(lib.vln_envdrop_step_fwd(C.byref(rec.dims), C.byref(mod._wstruct), C.byref(rec.io),
                                            torch.cuda.current_stream().cuda_stream), "vln_envdrop_step_fwd")
        ctx.mod, ctx.rec = mod, rec
        ctx.set_materialize_grads(False)
        k = rec.keep
        return k["logit"], k["h1"], k["c1"], k["h_tilde"]

    @staticmethod
    def backward(ctx, dlogit, dh1, dc1, dht):
        mod, rec = ctx.mod, ctx.rec
        lib = _lib.load()
        dev = rec.keep["h1"].device
        B, H = rec.dims.B, rec.dims.H
        g = _lib.EnvDropGrads()
        hold = []
        for name, t in (("dlogit", dlogit), ("dh1", dh1), ("dc1", dc1), ("dh_tilde", dht)):
            if t is not None:
                t = t.contiguous()
                hold.append(t)
                setattr(g, name, t.data_ptr())
        dhtp = torch.empty(B, H, dtype=torch.float32, device=dev)
        dc0 = torch.empty(B, H, dtype=torch.float32, device=dev)
        g.dh_tilde_prev, g.dc0 = dhtp.data_ptr(), dc0.data_ptr()
        e = rec.entry
        if ctx.needs_input_grad[4]:
            if e.dctx is None:
                e.dctx = torch.zeros(rec.B, rec.L, H, dtype=torch.float32, device=dev)
            g.dctx = e.dctx.data_ptr()
        s = rec.slot
        g.s_dtc, g.s_dz, g.s_dtt = s.view("dtc").data_ptr(), s.view("dz").data_ptr(), s.view("dtt").data_ptr()
        g.s_dgates, g.s_dtv, g.s_de = s.view("dgates").data_ptr(), s.view("dtv").data_ptr(), s.view("de").data_ptr()
        ws = ops.workspace(dev, rec.io.ws_floats)
        rec.io.ws = ws.data_ptr()
        _lib.check(lib.vln_envdrop_step_bwd(C.byref(rec.dims), C.byref(mod._wstruct), C.byref(rec.io), C.byref(g),
                                            torch.cuda.current_stream().cuda_stream), "vln_envdrop_step_bwd")
        s.done = True
        ctx.rec = None
        return (None, None, dhtp, dc0, None) + (None,) * (len(ctx.needs_input_grad) - 5)


class EnvDropDecoder(nn.Module, GatedModuleMixin):
    """policy.py:173-246.  `compute_dtype=torch.bfloat16` streams bf16 weight shadows / features / context
    with fp32 accumulation (BASELINE config 1); fp32 is bit-for-bit fp32 math on the f32 MFMA."""

    def __init__(self, hidden_size, drop_ratio, feat_drop_ratio, action_embed_size: int = 64,
                 angle_feat_size: int = 128, feature_size: int = 2048 + 128, compute_dtype=torch.float32):
        super().__init__()
        self.feature_size = feature_size
        self.action_embed_size = action_embed_size
        self.angle_feat_size = angle_feat_size
        self.hidden_size = hidden_size
        self.drop_ratio = float(drop_ratio)
        self.feat_drop_ratio = float(feat_drop_ratio)
        # parameter holders: same names/shapes/default init as the reference (policy.py:186-197)
        self.act_embed = nn.Sequential(nn.Linear(angle_feat_size, action_embed_size), nn.Tanh())
        self.drop = nn.Dropout(p=drop_ratio)
        self.env_drop = nn.Dropout(p=feat_drop_ratio)
        self.lstm = nn.LSTMCell(action_embed_size + feature_size, hidden_size)
        self.text_attn = _SoftDotParams(hidden_size)
        self.visual_attn = _SoftDotParams(hidden_size, context_dim=feature_size, context_only=True)
        self.cand_attn = nn.Linear(hidden_size, feature_size, bias=False)
        self._init_gating()
        self.compute_dtype = compute_dtype
        self._wstruct = _lib.EnvDropWeights()
        # Deferred weight-gradient GEMMs on a side stream (only when every grad lands in place).  OFF by default:
        # measured on MI355X it LOSES 20 % when the consumer that follows is the persistent encoder BPTT kernel,
        # whose 128 co-resident workgroups then compete for CUs with the side stream's GEMM workgroups.
        self.overlap_wgrads = False
        self._side_stream = None

    # ---- gating hooks ----------------------------------------------------------------------------------
    def _gated_params(self) -> List[torch.Tensor]:
        return [self.act_embed[0].weight, self.act_embed[0].bias, self.visual_attn.linear_in.weight,
                self.lstm.weight_ih, self.lstm.weight_hh, self.lstm.bias_ih, self.lstm.bias_hh,
                self.text_attn.linear_in.weight, self.text_attn.linear_out.weight, self.cand_attn.weight]

    def _stash_sites(self) -> Dict[str, int]:
        H, F, AE, ANG = self.hidden_size, self.feature_size, self.action_embed_size, self.angle_feat_size
        return {"a": ANG, "hq": H, "xcat": AE + F + H, "tcat": 2 * H, "htd": H,
                "de": AE, "dtv": F, "dgates": 4 * H, "dtt": H, "dz": H, "dtc": F}

    def _refresh_shadows(self):
        dt = self.compute_dtype
        t = self._shadow.t
        H, F, AE = self.hidden_size, self.feature_size, self.action_embed_size
        XK = AE + F + H
        dev = self.lstm.weight_ih.device

        def both(name, w):
            wf = w.detach()
            t[name] = wf if dt == torch.float32 else ops.cast_copy(wf, dt, t.get(name))
            t[name + "_t"] = ops.transpose_cast(wf, dt, t.get(name + "_t"))

        both("w_vin", self.visual_attn.linear_in.weight)
        both("w_tin", self.text_attn.linear_in.weight)
        both("w_tout", self.text_attn.linear_out.weight)
        both("w_c", self.cand_attn.weight)
        # fused LSTM weight [W_ih | W_hh] and its transpose
        wc = t.get("w_cat")
        if wc is None or wc.dtype != dt or wc.device != dev:
            wc = torch.empty(4 * H, XK, dtype=dt, device=dev)
            t["w_cat_t"] = torch.empty(XK, 4 * H, dtype=dt, device=dev)
        ops.cast_copy(self.lstm.weight_ih.detach(), dt, wc[:, :AE + F])
        ops.cast_copy(self.lstm.weight_hh.detach(), dt, wc[:, AE + F:])
        ops.transpose_cast(self.lstm.weight_ih.detach(), dt, t["w_cat_t"][:AE + F])
        ops.transpose_cast(self.lstm.weight_hh.detach(), dt, t["w_cat_t"][AE + F:])
        t["w_cat"] = wc
        w = self._wstruct
        w.act_w, w.act_b = self.act_embed[0].weight.data_ptr(), self.act_embed[0].bias.data_ptr()
        w.b_ih, w.b_hh = self.lstm.bias_ih.data_ptr(), self.lstm.bias_hh.data_ptr()
        for k in ("w_vin", "w_cat", "w_tin", "w_tout", "w_c"):
            setattr(w, k, t[k].data_ptr())
            setattr(w, k + "_t", t[k + "_t"].data_ptr())

    def _deferred_wgrads(self):
        """dW for every gated parameter from the stash: one contraction over (steps x batch) per weight."""
        H, F, AE = self.hidden_size, self.feature_size, self.action_embed_size
        P = self._gated_params()
        runs = list(self._stash.done_runs())
        self._gate_consumed()
        if not runs:         # no decoder step took part in this backward
            return [None] * len(P)
        # Where a parameter already has a (contiguous) .grad -- e.g. a dp.GradBucket view -- the contraction
        # accumulates straight into it and autograd gets None: no temporary, no extra add pass.
        tgt, ret, acc0 = [], [], []
        for p in P:
            if p.grad is not None and p.grad.is_contiguous() and p.grad.dtype == torch.float32:
                tgt.append(p.grad); ret.append(None); acc0.append(True)
            else:
                t = torch.empty_like(p)
                tgt.append(t); ret.append(t); acc0.append(False)
        (g_aw, g_ab, g_vin, g_ih, g_hh, g_bih, g_bhh, g_tin, g_tout, g_c) = tgt
        # When every gradient lands in place nothing downstream of this node reads the results before backward()
        # returns, so the contractions go to a side stream and overlap the encoder's BPTT (128 workgroups of
        # latency-bound recurrence leave half the chip idle); the main stream re-joins when backward finishes.
        side = None
        if self.overlap_wgrads and all(acc0):
            main = torch.cuda.current_stream()
            side = self._side_stream
            if side is None or side.device != main.device:
                side = self._side_stream = torch.cuda.Stream(main.device)
            side.wait_stream(main)
            torch.autograd.Variable._execution_engine.queue_callback(lambda: main.wait_stream(side))
        with torch.cuda.stream(side) if side is not None else contextlib.nullcontext():
            self._issue_wgrads(runs, acc0, tgt)
        return ret

    def _issue_wgrads(self, runs, acc0, tgt):
        H, F, AE = self.hidden_size, self.feature_size, self.action_embed_size
        (g_aw, g_ab, g_vin, g_ih, g_hh, g_bih, g_bhh, g_tin, g_tout, g_c) = tgt
        for i, (bufs, r0, r1) in enumerate(runs):
            sl = slice(r0, r1)
            a = [x or i > 0 for x in acc0]      # the first run overwrites fresh buffers, everything else accumulates
            ops.linear_wgrad(bufs["de"][sl], bufs["a"][sl], g_aw, a[0])
            ops.colsum(bufs["de"][sl], g_ab, a[1])
            ops.linear_wgrad(bufs["dtv"][sl], bufs["hq"][sl], g_vin, a[2])
            ops.linear_wgrad(bufs["dgates"][sl], bufs["xcat"][sl][:, :AE + F], g_ih, a[3])
            ops.linear_wgrad(bufs["dgates"][sl], bufs["xcat"][sl][:, AE + F:], g_hh, a[4])
            ops.colsum(bufs["dgates"][sl], g_bih, a[5])
            ops.colsum(bufs["dgates"][sl], g_bhh, a[6])   # d b_hh == d b_ih (same pre-activation)
            ops.linear_wgrad(bufs["dtt"][sl], bufs["tcat"][sl][:, H:], g_tin, a[7])
            ops.linear_wgrad(bufs["dz"][sl], bufs["tcat"][sl], g_tout, a[8])
            ops.linear_wgrad(bufs["dtc"][sl], bufs["htd"][sl], g_c, a[9])

    # ---- forward -----------------------------------------------------------------------------------------
    def forward(self, a_t_prev, img_feature, cand_feature, h_tilde_prev, h_0, c_0, ctx, ctx_mask=None,
                already_dropfeat=False, img_lp=None, cand_lp=None):
        """Same contract as policy.py:208-246 (h_0 is unused there too).  img_feature / cand_feature are
        overwritten in place by the feature dropout, like the reference.  Extension (optional): `img_lp` / `cand_lp`
        = bf16 copies of the two feature tensors already produced by `DeviceFeatureStore.gather_*` (together with
        `already_dropfeat=True`): the decoder then skips its own dropout/copy pass over the features."""
        if not img_feature.is_cuda:
            raise _lib.VlnError("EnvDropDecoder: tensors must be on the GPU; there is no CPU fallback")
        B, V, F = img_feature.shape
        Cn = cand_feature.shape[1]
        L = ctx.shape[1]
        H, AE, ANG = self.hidden_size, self.action_embed_size, self.angle_feat_size
        dev = img_feature.device
        need_grad = torch.is_grad_enabled() and (
            h_tilde_prev.requires_grad or c_0.requires_grad or ctx.requires_grad or self.lstm.weight_ih.requires_grad)
        gated = self._ensure_current(need_grad)
        ctx_in, entry = self._gated_ctx(ctx, need_grad)
        dt = self.compute_dtype
        lp = dt != torch.float32

        rec = _StepRec()
        rec.B, rec.L, rec.C = B, L, Cn
        rec.ctx_owner, rec.entry = ctx, entry
        d = _lib.EnvDropDims(B, L, V, Cn, H, F - ANG, ANG, AE, ops.BF16 if lp else ops.F32, ops.BF16 if lp else ops.F32)
        rec.dims = d
        img = img_feature if img_feature.is_contiguous() else img_feature.contiguous()
        cand = cand_feature if cand_feature.is_contiguous() else cand_feature.contiguous()
        a = a_t_prev.contiguous()
        htp = h_tilde_prev.detach().contiguous()
        c0 = c_0.detach().contiguous()
        ctxc = ctx.detach()
        if not ctxc.is_contiguous():
            ctxc = ctxc.contiguous()
        f32 = dict(dtype=torch.float32, device=dev)
        keep = {"img": img, "cand": cand, "htp": htp, "c0": c0, "ctx": ctxc,
                "logit": torch.empty(B, Cn, **f32), "h1": torch.empty(B, H, **f32), "c1": torch.empty(B, H, **f32),
                "h_tilde": torch.empty(B, H, **f32)}
        # per-step saved activations: one flat allocation carved into views
        sizes = (("e", B * AE), ("alpha_v", B * V), ("gate_act", B * 4 * H), ("tanh_c1", B * H), ("tt", B * H),
                 ("alpha_t", B * L))
        flat = torch.empty(sum(n for _, n in sizes), **f32)
        off = 0
        for name, n in sizes:
            keep[name] = flat[off:off + n]
            off += n
        if need_grad:
            slot = self._stash.take(B, entry.ref)
            rec.slot = slot
            xa, hq, xcat, tcat, htd = (slot.view(k) for k in ("a", "hq", "xcat", "tcat", "htd"))
            xa.copy_(a)
            a_ptr = xa.data_ptr()
        else:
            rec.slot = None
            hq, xcat, tcat, htd = (torch.empty(B, w, **f32) for w in (H, AE + F + H, 2 * H, H))
            keep["tmp"] = (hq, xcat, tcat, htd)
            a_ptr = a.data_ptr()
            keep["a"] = a
        io = _lib.EnvDropStep()
        io.a_prev, io.img, io.cand = a_ptr, img.data_ptr(), cand.data_ptr()
        if lp:
            ready = already_dropfeat and img_lp is not None and cand_lp is not None and img_lp.dtype == dt and \
                cand_lp.dtype == dt and img_lp.is_contiguous() and cand_lp.is_contiguous()
            keep["img_lp"] = img_lp if ready else torch.empty(B, V, F, dtype=dt, device=dev)
            keep["cand_lp"] = cand_lp if ready else torch.empty(B, Cn, F, dtype=dt, device=dev)
            io.lp_ready = int(bool(ready))
            keep["ctx_lp"] = self._ctx_lp(entry, ctx, dt)
            io.img_lp, io.cand_lp, io.ctx_lp = keep["img_lp"].data_ptr(), keep["cand_lp"].data_ptr(), keep["ctx_lp"].data_ptr()
        io.h_tilde_prev, io.c0, io.ctx = htp.data_ptr(), c0.data_ptr(), ctxc.data_ptr()
        if ctx_mask is not None:
            m8 = ctx_mask.contiguous().view(torch.uint8) if ctx_mask.dtype == torch.bool else ctx_mask.to(torch.uint8).contiguous()
            keep["mask"] = m8
            io.ctx_mask = m8.data_ptr()
        io.logit, io.h1, io.c1, io.h_tilde = (keep[k].data_ptr() for k in ("logit", "h1", "c1", "h_tilde"))
        io.e, io.xcat, io.hq = keep["e"].data_ptr(), xcat.data_ptr(), hq.data_ptr()
        io.alpha_v, io.gate_act, io.tanh_c1 = keep["alpha_v"].data_ptr(), keep["gate_act"].data_ptr(), keep["tanh_c1"].data_ptr()
        io.tcat, io.tt, io.alpha_t, io.htd = tcat.data_ptr(), keep["tt"].data_ptr(), keep["alpha_t"].data_ptr(), htd.data_ptr()
        io.seed, io.offset = self.dropout_seed, self._next_offset()
        io.p_drop = self.drop_ratio if self.training else 0.0
        io.p_feat = self.feat_drop_ratio if self.training else 0.0
        io.already_dropfeat = int(bool(already_dropfeat))
        nws = _lib.load().vln_envdrop_ws_floats(C.byref(d))
        ws = ops.workspace(dev, nws)
        io.ws, io.ws_floats = ws.data_ptr(), nws
        rec.io, rec.keep = io, keep

        if need_grad:
            logit, h1, c1, h_tilde = _EnvDropStepFn.apply(self, rec, h_tilde_prev, c_0, ctx_in, *gated)
        else:
            _lib.check(_lib.load().vln_envdrop_step_fwd(C.byref(d), C.byref(self._wstruct), C.byref(io),
                                                        torch.cuda.current_stream().cuda_stream), "vln_envdrop_step_fwd")
            logit, h1, c1, h_tilde = keep["logit"], keep["h1"], keep["c1"], keep["h_tilde"]
        if img is not img_feature:
            img_feature.copy_(img)
        if cand is not cand_feature:
            cand_feature.copy_(cand)
        return logit, (h1, c1), h_tilde


class Critic(nn.Module):
    """policy.py:249-267: Linear -> ReLU -> Dropout -> Linear -> squeeze, on the HIP linear kernels."""

    def __init__(self, hidden_size, drop_ratio):
        super().__init__()
        self.hidden_size = hidden_size
        self.drop_ratio = drop_ratio
        self.state2value = nn.Sequential(nn.Linear(hidden_size, hidden_size), nn.ReLU(), nn.Dropout(drop_ratio),
                                         nn.Linear(hidden_size, 1))
        self.dropout_seed = 0xC417
        self._calls = 0

    def forward(self, state):
        self._calls += 1
        p = self.drop_ratio if self.training else 0.0
        l0, l3 = self.state2value[0], self.state2value[3]
        return _CriticFn.apply(state, l0.weight, l0.bias, l3.weight, l3.bias, p, self.dropout_seed, self._calls).squeeze()


class _CriticFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w0, b0, w3, b3, p, seed, offset):
        x = x.contiguous()
        z = ops.linear_fwd(x, w0.detach(), b0.detach(), ops.ACT_RELU)
        n = z.numel()
        m = ops.dropout_mask(n, seed, offset, p, x.device).view_as(z) if p > 0 else None
        zd = z * m if m is not None else z
        v = ops.linear_fwd(zd, w3.detach(), b3.detach())
        ctx.save_for_backward(x, w0, w3, z, zd)
        ctx.m = m
        return v

    @staticmethod
    def backward(ctx, dv):
        x, w0, w3, z, zd = ctx.saved_tensors
        dv = dv.contiguous()
        dzd = ops.linear_fwd(dv, w3.detach().t().contiguous())            # [B,1] x [H,1]^T
        if ctx.m is not None:
            dzd = dzd * ctx.m
        dz = dzd * (z > 0).to(dzd.dtype)
        dw3 = ops.linear_wgrad(dv, zd)
        db3 = dv.sum(0)
        dx = ops.linear_fwd(dz, ops.transpose_cast(w0.detach()))
        dw0 = ops.linear_wgrad(dz, x)
        db0 = ops.colsum(dz)
        return dx, dw0, db0, dw3, db3, None, None, None
