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


class _StepPlan:
    """What a decoder step needs again when the same argument block comes back (see EnvDropDecoder.forward)."""
    __slots__ = ("io", "dims", "nws", "keep", "i0", "n_alloc", "xcat_ptr", "ctx_lp_ptr", "mask_ptr", "first_ptr", "kctx_ptr")

    def __init__(self, io, dims, nws, keep, i0, n_alloc, xcat_ptr, ctx_lp_ptr, mask_ptr, first_ptr, gathered=False, kctx_ptr=0):
        self.io = _lib.EnvDropStep.from_buffer_copy(io)
        own = ("logit", "h1", "c1", "h_tilde", "flat", "img_lp", "cand_lp")      # the module's buffers only: the caller's
        if gathered:                                                              # tensors are taken afresh every call
            own = own + ("img", "cand")
        self.dims, self.nws, self.i0, self.n_alloc = dims, nws, i0, n_alloc
        self.keep = {k: keep[k] for k in own if k in keep}
        self.xcat_ptr, self.ctx_lp_ptr, self.mask_ptr, self.first_ptr = xcat_ptr, ctx_lp_ptr, mask_ptr, first_ptr
        self.kctx_ptr = kctx_ptr


class _StepRec:
    __slots__ = ("io", "dims", "slot", "keep", "ctx_owner", "entry", "B", "L", "C", "H", "mod", "dhtd_ext", "dhtd_keep", "flushed")


_NONES = (None,) * 64


def _tp(t):
    return 0 if t is None else t.data_ptr()


class _EnvDropStepFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, rec, h_tilde_prev, c0, ctx_t, gate_token):
        st = _lib.load().vln_envdrop_step_fwd(C.byref(rec.dims), C.byref(mod._wstruct), C.byref(rec.io), _lib.raw_stream())
        if st:
            _lib.check(st, "vln_envdrop_step_fwd")
        if rec.io.chain & 1:
            # the step left its last stage PENDING in the library: what that stage writes (h_tilde, the htd stash row) stays
            # alive until the next chained step or flush has replaced this reference, whatever happens to the rollout
            object.__setattr__(mod, "_pend_keep", rec.keep)
        ctx.mod, ctx.rec = mod, rec
        ctx.set_materialize_grads(False)
        k = rec.keep
        # fresh aliases: the returned objects get this node as grad_fn, and rec (reachable from the node) must not hold
        # them, or every step that never runs backward leaks until the cyclic GC passes
        return k["logit"].detach(), k["h1"].detach(), k["c1"].detach(), k["h_tilde"].detach()

    @staticmethod
    def backward(ctx, dlogit, dh1, dc1, dht):
        mod, rec = ctx.mod, ctx.rec
        dev = rec.keep["h1"].device
        B, H = rec.B, rec.H
        s = rec.slot
        want_ctx = ctx.needs_input_grad[4]
        # backward plans (arena mode): same idea as the forward step plans -- the gradient block of this step two
        # iterations ago is reused when every address in it repeats
        arena = ops.current_arena() if mod.step_graphs else None
        bkey = None
        if arena is not None:
            bkey = (arena.g, arena.i, 0 if dlogit is None else dlogit.data_ptr(), 0 if dh1 is None else dh1.data_ptr(),
                    0 if dc1 is None else dc1.data_ptr(), 0 if dht is None else dht.data_ptr(), want_ctx, s.ptr("dtc"), B, rec.L)
            bp = mod._bplans.get(bkey)
            if (bp is not None and arena.reserve(bp[1], bp[2], bp[3]) and (dlogit is None or dlogit.is_contiguous())
                    and (dh1 is None or dh1.is_contiguous()) and (dc1 is None or dc1.is_contiguous())
                    and (dht is None or dht.is_contiguous())):
                arena.i = bp[1] + bp[2]
                g = _lib.EnvDropGrads.from_buffer_copy(bp[0])
                g.dhtd_ext = rec.dhtd_ext          # the rollout-wide logit branch (EnvDropDecoder.logit_branch_backward), if it ran
                dhtp, dc0, t = bp[4], bp[5], bp[6]
                if want_ctx:
                    io0 = rec.io
                    rec.entry.terms.append((io0.alpha_t, g.s_dl, g.s_dtcat, io0.tcat + 4 * H if io0.kctx else io0.tt, t, rec.keep["flat"]))
                    rec.entry.shape = (B, rec.L, H)
                io = rec.io
                io.ws = mod._step_ws(dev, io.ws_floats, io.chain != 0)
                st = _lib.load().vln_envdrop_step_bwd(C.byref(rec.dims), C.byref(mod._wstruct), C.byref(io), C.byref(g), _lib.raw_stream())
                if st:
                    _lib.check(st, "vln_envdrop_step_bwd")
                s.done = True
                rec.dhtd_keep = None
                ctx.rec = None
                if io.chain & 2:
                    object.__setattr__(mod, "_pend_keep_b", (dhtp, rec.keep))      # what the pending backward stage writes / reads
                return None, None, dhtp, dc0, None, None
        arena_i0 = arena.i if arena is not None else 0
        g = _lib.EnvDropGrads()
        hold = []          # keeps the contiguous copies alive until the launch is queued
        if dlogit is not None:
            if not dlogit.is_contiguous():
                dlogit = dlogit.contiguous(); hold.append(dlogit)
            g.dlogit = dlogit.data_ptr()
        if dh1 is not None:
            if not dh1.is_contiguous():
                dh1 = dh1.contiguous(); hold.append(dh1)
            g.dh1 = dh1.data_ptr()
        if dc1 is not None:
            if not dc1.is_contiguous():
                dc1 = dc1.contiguous(); hold.append(dc1)
            g.dc1 = dc1.data_ptr()
        if dht is not None:
            if not dht.is_contiguous():
                dht = dht.contiguous(); hold.append(dht)
            g.dh_tilde = dht.data_ptr()
        dhtp = ops.empty(B, H, dtype=torch.float32, device=dev)
        dc0 = ops.empty(B, H, dtype=torch.float32, device=dev)
        g.dh_tilde_prev, g.dc0 = dhtp.data_ptr(), dc0.data_ptr()
        t = None
        if want_ctx:
            # context gradient deferred: this step leaves d(text logits) and d(weighted ctx) behind; CtxGate forms
            # dctx once per rollout from all steps (one write of [B,L,H] instead of T read-modify-write sweeps)
            L = rec.L
            n_dl = (B * L + 3) & ~3                      # keeps the [B,2H] block 16-byte aligned
            t = ops.empty(n_dl + B * 2 * H, dtype=torch.float32, device=dev)
            q = t.data_ptr()
            g.s_dl, g.s_dtcat = q, q + 4 * n_dl
            io0 = rec.io
            # (the q operand of the deferred dctx: the step's query -- or, on the projected context, its drop(h_1) rows)
            rec.entry.terms.append((io0.alpha_t, q, q + 4 * n_dl, io0.tcat + 4 * H if io0.kctx else io0.tt, t, rec.keep["flat"]))
            rec.entry.shape = (B, L, H)
        g.s_dtc, g.s_dz, g.s_dtt = s.ptr("dtc"), s.ptr("dz"), s.ptr("dtt")
        g.s_dgates, g.s_dtv, g.s_de = s.ptr("dgates"), s.ptr("dtv"), s.ptr("de")
        g.dhtd_ext = rec.dhtd_ext
        if bkey is not None and not hold:
            if len(mod._bplans) > 256:
                mod._bplans.clear()
            mod._bplans[bkey] = (_lib.EnvDropGrads.from_buffer_copy(g), arena_i0, arena.i - arena_i0, dhtp.data_ptr(), dhtp, dc0, t)
        io = rec.io
        io.ws = mod._step_ws(dev, io.ws_floats, io.chain != 0)
        st = _lib.load().vln_envdrop_step_bwd(C.byref(rec.dims), C.byref(mod._wstruct), C.byref(io), C.byref(g), _lib.raw_stream())
        if st:
            _lib.check(st, "vln_envdrop_step_bwd")
        s.done = True
        rec.dhtd_keep = None
        ctx.rec = None
        if io.chain & 2:
            object.__setattr__(mod, "_pend_keep_b", (dhtp, rec.keep))
        return None, None, dhtp, dc0, None, None


class EnvDropDecoder(nn.Module, GatedModuleMixin):
    """policy.py:173-246.  `compute_dtype=torch.bfloat16` streams bf16 weight shadows / features / context
    with fp32 accumulation (BASELINE config 1); fp32 is bit-for-bit fp32 math on the f32 MFMA."""
    # Class-wide default of `fp32_weights`.  Round 4: the two attention QUERY projections (visual_attn.linear_in 512 x 2176,
    # text_attn.linear_in 512 x 512) are streamed in fp32 by default -- their 2^-9 bf16 rounding sits in front of a softmax and
    # alone put d h_tilde / d visual_attn.linear_in at 1.4e-2 of the fp32 reference; with them in fp32 EVERY logit and gradient
    # of BASELINE config 1 is within north_star's 1e-2 (tests/test_hip_modules.py::test_envdrop_full_size_bf16).  `frozenset()`
    # = every matrix bf16 (round 3's default; bench.py secondary `all_bf16_weights_ms_per_step`).
    default_step_graphs = True       # decoder steps as hipGraphs by default (see __init__)
    IN_STEP_SAMPLER_MAX_C = 64       # candidates per row the sampled branch takes (cand_logits_sample, vln_categorical_fwd: one wavefront)
    default_fp32_weights = frozenset({"w_vin", "w_tin"})

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
        self.batch_logit_backward = True      # losses.RolloutCE hands all d logits of a rollout over at once (logit_branch_backward)
        # Opt-in, TEACHER FORCING ONLY: forward() returns `logit` tensors that are filled later -- for all steps of the
        # rollout at once (one GEMM over steps x batch + one multi-step dot launch, logit_branch_forward) when
        # losses.RolloutCE evaluates the loss.  Nothing may read the logits before that (a sampled / greedy rollout does).
        self.defer_logits = False
        # Opt-in, with defer_logits (TEACHER FORCING ONLY): consecutive steps are CHAINED -- a step leaves the tanh + dropout
        # epilogue of its last product pending and the next step's first launch finishes it (the same in the backward with the
        # act-embedding / h_tilde_prev stage): one dependent launch less per step and direction (vln_envdrop_step.chain).  Nothing
        # but the next step (and the rollout-level logits / loss / weight gradients, which flush) may read a step's h_tilde.
        self.chain_steps = False
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
        self._dims_cache = {}
        self._plans = {}                 # step plans of the arena mode (see forward)
        self._bplans = {}                # ... and of the step backward
        self.plan_hits = 0
        self.grads_ready_hook = None     # optional callable, see _deferred_wgrads
        # True (bf16 mode, gradients accumulated in place, one stash run): the rollout's weight / bias gradient launches are
        # posted as a gradient ride (ops.GradRide, csrc/wgrad_ride.h) and travel as passengers of the encoder's BPTT launch instead
        # of standing in front of it (57-69 us of the dependent chain at B = 64).  Single-GPU schedules only: a data-parallel
        # caller wants the decoder's gradients final BEFORE the BPTT (grads_ready_hook).
        self.ride_wgrads = False
        # Replay each decoder step (forward: 9-13 launches, backward: up to 15) as ONE hipGraph.  A graph is keyed by the step's
        # argument block, i.e. by device addresses; it pays when the caller's tensors come back at the SAME addresses every
        # iteration: guaranteed under ops.RolloutArena, and in practice also with PyTorch's caching allocator once a training loop
        # has settled (round 6, the reference's unchanged caller: 458 replays against 46 captures over 36 iterations, host-bound
        # iteration 2.16 -> 1.95 ms; a chain that never repeats pauses its own capturing: csrc/graph_cache.h).  Results are
        # identical either way (tests/test_hip_modules.py::test_per_step_graphs_without_an_arena_equal_plain_launches).
        self.step_graphs = self.default_step_graphs
        # The step's two attentions on FOUR workgroups per episode (csrc/attention_split.h) instead of one: 256 workgroups at
        # B = 64.  Needs a zero-initialised exchange buffer that lives as long as the module (allocated on first use) and
        # B * 4 <= the device's CU count (the library checks; larger batches take the one-workgroup kernels).
        self.split_attention = True
        self._attn_sync = None
        # The text attention on the PROJECTED context (round 5): K = ctx W_in is formed once per rollout (one product over the
        # B * L context rows) and every step scores K . drop(h_1) -- units.py:106-109 re-associated -- so the per-step query
        # product W_in drop(h_1) and its transpose in the backward leave the step, and the LSTM cell's pointwise stage runs inside
        # the text-attention launch (vln_envdrop_step.kctx, csrc/attention_textk.h): two dependent launches less per step and
        # direction.  Taken when the four-workgroup attention covers the shape (vln_attn_textk_ok), else the step runs as before.
        self.project_context = True
        self.last_projected = False
        # Opt-in for SAMPLED rollouts (whose forward cannot be chained: every step's logits are read): the BACKWARD of consecutive
        # steps is chained like `chain_steps` chains it -- step t's act-embedding / h_tilde_prev stage rides in the first launch of
        # step t - 1's backward (vln_envdrop_step.chain == 2) -- for every step whose h_tilde_prev IS the previous call's h_tilde.
        # The caller promises that nothing but the next step consumes a step's h_tilde (its logits and h_1 / c_1 are free).
        self.chain_backward = False
        self.__dict__["_last_ht"] = None
        self.__dict__["_sampler_arg"] = None
        # bf16 compute: weight matrices that are streamed in fp32 all the same (names: w_vin, w_cat, w_tin, w_tout, w_c).  Set it
        # before the first forward (the shadows are rebuilt when a parameter changes).
        self.fp32_weights = frozenset(type(self).default_fp32_weights)

    def _attn_sync_buf(self, dev, B):
        w = self._attn_sync
        if w is None or w[0].device != dev or w[1] < B:
            n = int(_lib.load().vln_attn_sync_bytes(B))
            w = self._attn_sync = (torch.zeros((n + 3) // 4, dtype=torch.int32, device=dev), B)
        return w[0]

    def _gate_opened(self):
        # A gate opens when the previous rollouts' weight gradients have been issued (they flush what was pending) -- or when an
        # iteration was ABANDONED (a forward or backward pass that raised): a chained step's pending stage or a posted gradient
        # ride of that iteration would otherwise be issued by the next call, on buffers of a dead rollout.
        if self.chain_steps or self.chain_backward or self.ride_wgrads:
            lib = _lib.load()
            lib.vln_envdrop_drop_pending(_lib.raw_stream())
            if self.ride_wgrads:
                lib.vln_wgrad_ride_drop(_lib.raw_stream())

    def _step_ws(self, dev, floats, chained):
        """The step's split-K scratch.  Chained steps leave slabs PENDING in it between two step calls, so they get a buffer of
        their own: the shared per-stream `ops.workspace` is also the scratch of every other product, weight-gradient batch and
        gradient ride issued in between (and is reallocated when a bigger request comes)."""
        if not chained:
            return ops.workspace(dev, floats).data_ptr()
        w = self.__dict__.get("_chain_ws")
        if w is None or w.device != dev or w.numel() < floats:
            if w is not None:          # growing: nothing may still be pending in the old buffer
                _lib.check(_lib.load().vln_envdrop_flush(_lib.raw_stream()), "vln_envdrop_flush")
            w = torch.empty(int(floats), dtype=torch.float32, device=dev)
            object.__setattr__(self, "_chain_ws", w)
        return w.data_ptr()

    def _ctx_k(self, entry, ctx, lp):
        """The rollout's projected context K = ctx W_in ([B,L,H] fp32), formed on first use; False when the folded text attention
        does not cover this shape / device (the steps then project their queries themselves)."""
        k = entry.kctx
        if k is not None:
            return k
        B, L, H = ctx.shape
        k = False
        if self.project_context and self.split_attention and H == self.hidden_size:
            sy = self._attn_sync_buf(ctx.device, B)
            if _lib.load().vln_attn_textk_ok(ops.BF16 if lp else ops.F32, B, L, H, sy.data_ptr(), sy.numel() * 4):
                src = ctx.detach()
                if not src.is_contiguous():
                    src = src.contiguous()
                w_t = self._shadow.t["w_tin_t"]                  # W_in^T [H(query), H(ctx)]: K = ctx @ W_in
                k = ops.linear_fwd(src.view(B * L, H), w_t, split=lp).view(B, L, H)
                entry.k_w, entry.k_split = self._shadow.t["w_tin"], lp
                entry.k_hd = self._hd_view
        entry.kctx = k
        self.last_projected = k is not False          # (tests: which copy of ctx the latest rollout's logits were taken on)
        return k

    def _hd_view(self, address: int, rows: int):
        """[rows, H] view (row stride 2H) of the `tcat` stash whose first element lies at `address` (a step's drop(h_1) block), or
        None when those rows are not inside one chunk of the stash."""
        H = self.hidden_size
        for ch in (self._stash.chunks if self._stash is not None else ()):
            t = ch.bufs["tcat"]
            off = address - t.data_ptr()
            if 0 <= off and off + ((rows - 1) * 2 * H + H) * 4 <= t.numel() * 4 and off % 4 == 0:
                e0 = off // 4
                return t.view(-1).as_strided((rows, H), (2 * H, 1), e0)
        return None

    def scores_on_projected_context(self, ctx) -> bool:
        """Whether the rollout on `ctx` (after its first step) takes its text-attention logits on K = ctx W_in formed from the
        fp32 context (True) or on the streamed copy of ctx (False) -- what a bit-level restatement of the bf16 mode has to know."""
        e = self._ctx_entries.get(id(ctx))
        return bool(e is not None and e.ref() is ctx and e.kctx is not None and e.kctx is not False)

    # ---- gating hooks ----------------------------------------------------------------------------------
    def _gated_params(self) -> List[torch.Tensor]:
        # Called every step: read the Parameters through the owning submodules' `_parameters` dicts (resolved once)
        # instead of ten nn.Module.__getattr__ chains; re-assigned parameters are still seen.
        slots = self.__dict__.get("_pslots")
        if slots is None:
            a, l = self.act_embed[0]._parameters, self.lstm._parameters
            slots = ((a, "weight"), (a, "bias"), (self.visual_attn.linear_in._parameters, "weight"),
                     (l, "weight_ih"), (l, "weight_hh"), (l, "bias_ih"), (l, "bias_hh"),
                     (self.text_attn.linear_in._parameters, "weight"), (self.text_attn.linear_out._parameters, "weight"),
                     (self.cand_attn._parameters, "weight"))
            object.__setattr__(self, "_pslots", slots)
        return [d[k] for d, k in slots]

    def __setattr__(self, name, value):
        if isinstance(value, nn.Module):          # a replaced submodule invalidates the cached parameter slots
            self.__dict__.pop("_pslots", None)
        super().__setattr__(name, value)

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

        ck = (dt, self.fp32_weights, tuple(p.data_ptr() for p in self._gated_params()))
        c = self.__dict__.get("_sb_handle")
        if c is not None and c[0] == ck:          # same addresses, new values (an optimizer step): replay the recorded jobs
            ops.ShadowBatch.replay(c[1])
            return
        sb = ops.ShadowBatch()          # every cast / transpose below goes out as ONE launch

        f32 = self.fp32_weights

        def buf(name, shape, d=None):
            d = d or dt
            x = t.get(name)
            if x is None or x.dtype != d or x.device != dev or x.shape != shape:
                x = t[name] = torch.empty(shape, dtype=d, device=dev)
            return x

        def both(name, w):
            wf = w.detach()
            N, K = wf.shape
            if dt == torch.float32 or name in f32:
                t[name] = wf
                sb.add(wf, None, buf(name + "_t", (K, N), torch.float32))
            else:
                sb.add(wf, buf(name, (N, K)), buf(name + "_t", (K, N)))

        both("w_vin", self.visual_attn.linear_in.weight)
        both("w_tin", self.text_attn.linear_in.weight)
        both("w_tout", self.text_attn.linear_out.weight)
        both("w_c", self.cand_attn.weight)
        # fused LSTM weight [W_ih | W_hh] and its transpose
        dc = torch.float32 if "w_cat" in f32 else dt
        wc, wct = buf("w_cat", (4 * H, XK), dc), buf("w_cat_t", (XK, 4 * H), dc)
        sb.add(self.lstm.weight_ih.detach(), wc[:, :AE + F], wct[:AE + F])
        sb.add(self.lstm.weight_hh.detach(), wc[:, AE + F:], wct[AE + F:])
        handle = sb.run()
        w = self._wstruct
        w.act_w, w.act_b = self.act_embed[0].weight.data_ptr(), self.act_embed[0].bias.data_ptr()
        w.b_ih, w.b_hh = self.lstm.bias_ih.data_ptr(), self.lstm.bias_hh.data_ptr()
        w.f32_mask = 0
        for i, k in enumerate(("w_vin", "w_cat", "w_tin", "w_tout", "w_c")):
            setattr(w, k, t[k].data_ptr())
            setattr(w, k + "_t", t[k + "_t"].data_ptr())
            if dt != torch.float32 and k in f32:
                w.f32_mask |= 1 << i
        object.__setattr__(self, "_sb_handle", (ck, handle))

    def logit_branch_forward(self, recs):
        """`defer_logits`: the candidate logits of every recorded step, logit_t = cand_t . (W_c drop(h_tilde_t))
        (policy.py:199-206,243-244), formed now: ONE GEMM over (steps x batch) rows of the `htd` stash per contiguous run
        and ONE multi-step dot launch writing into the tensors forward() already returned."""
        lib = _lib.load()
        if self.chain_steps:          # the last step's pending epilogue (its drop(h_tilde) row is read below)
            _lib.check(lib.vln_envdrop_flush(_lib.raw_stream()), "vln_envdrop_flush")
        lp = self.compute_dtype != torch.float32
        F = self.feature_size
        B = recs[0].B
        runs = []
        for rec in recs:
            sl = rec.slot
            if runs and runs[-1][0] is sl.chunk and runs[-1][2] == sl.r0:
                runs[-1][2] = sl.r0 + sl.rows
                runs[-1][3].append(rec)
            else:
                runs.append([sl.chunk, sl.r0, sl.r0 + sl.rows, [rec]])
        w_c = self._shadow.t["w_c"]
        steps, hold = [], []
        for chunk, r0, r1, rs in runs:
            q = ops.linear_fwd(chunk.bufs["htd"][r0:r1], w_c)                       # [(steps x B), F]
            hold.append(q)
            base = q.data_ptr()
            for rec in rs:
                cand = rec.keep["cand_lp"] if lp else rec.keep["cand"]
                steps.append(_lib.DotStep(cand.data_ptr(), base + (rec.slot.r0 - r0) * F * 4, rec.keep["logit"].data_ptr(), rec.C))
                rec.flushed = True
        ctype = ops.BF16 if lp else ops.F32
        for i in range(0, len(steps), _lib.CE_MAX_STEPS):
            chunk = steps[i:i + _lib.CE_MAX_STEPS]
            arr = (_lib.DotStep * len(chunk))(*chunk)
            st = lib.vln_attn_dot_multi(arr, len(chunk), ctype, B, F, F, _lib.raw_stream())
            if st:
                _lib.check(st, "vln_attn_dot_multi")

    def logit_branch_backward(self, pairs, ce=None):
        """The candidate-logit branch of the backward, logit_t = cand_t . (W_c drop(h_tilde_t)) (policy.py:199-206,243-244), for
        ALL steps of a rollout at once: it depends on the d logits only -- which `losses.RolloutCE` produces for every step
        in one launch at the root of the backward -- and on no other step's backward, so 2 T skinny launches on the
        dependent chain (rows_wsum + an M = B GEMM per step) become ONE multi-step weighted sum, written straight into the
        steps' `dtc` stash rows (the dY operand of d cand_attn.weight), and ONE GEMM over (steps x batch) rows per contiguous
        stash run.  pairs: [(step record, d logits [B, C_t])] in rollout order.  Each step's backward then finds its [B,H] block
        (`vln_envdrop_grads.dhtd_ext`) and skips the branch.  With `ce` = (per-step (probs, target) list, d loss scalar tensor,
        scale, ignore_index) the d logits are never materialised: the weighted-sum launch forms them from the loss's saved
        probabilities on the fly (pairs then carry None in their place)."""
        lib = _lib.load()
        lp = self.compute_dtype != torch.float32
        F, H = self.feature_size, self.hidden_size
        B = pairs[0][0].B
        steps, runs = [], []
        ce_scale, ce_dloss, ce_ignore = 1.0, None, -1
        if ce is not None:
            ce_pt, dloss, ce_scale, ce_ignore = ce
            ce_dloss = dloss.data_ptr()
        for i, (rec, dl) in enumerate(pairs):
            cand = rec.keep["cand_lp"] if lp else rec.keep["cand"]
            if dl is not None:
                steps.append(_lib.WsumStep(cand.data_ptr(), dl.data_ptr(), rec.slot.ptr("dtc"), rec.C, None, None))
            else:
                steps.append(_lib.WsumStep(cand.data_ptr(), None, rec.slot.ptr("dtc"), rec.C, ce_pt[i][0].data_ptr(), ce_pt[i][1].data_ptr()))
            sl = rec.slot
            if runs and runs[-1][0] is sl.chunk and runs[-1][2] == sl.r0:
                runs[-1][2] = sl.r0 + sl.rows
                runs[-1][3].append(rec)
            else:
                runs.append([sl.chunk, sl.r0, sl.r0 + sl.rows, [rec]])
        ctype = ops.BF16 if lp else ops.F32
        for i in range(0, len(steps), _lib.CE_MAX_STEPS):
            chunk = steps[i:i + _lib.CE_MAX_STEPS]
            arr = (_lib.WsumStep * len(chunk))(*chunk)
            st = lib.vln_rows_wsum_multi(arr, len(chunk), ctype, B, F, F, ce_scale, ce_dloss, ce_ignore, _lib.raw_stream())
            if st:
                _lib.check(st, "vln_rows_wsum_multi")
        w_c_t = self._shadow.t["w_c_t"]
        for chunk, r0, r1, recs in runs:
            dh = ops.linear_fwd(chunk.bufs["dtc"][r0:r1], w_c_t)                    # [(steps x B), H]
            base = dh.data_ptr()
            for rec in recs:
                rec.dhtd_ext = base + (rec.slot.r0 - r0) * H * 4
                rec.dhtd_keep = dh

    def _deferred_wgrads(self):
        """dW for every gated parameter from the stash: one contraction over (steps x batch) per weight."""
        H, F, AE = self.hidden_size, self.feature_size, self.action_embed_size
        # the first step's pending prep backward (its act-embedding gradient rows are read below): chained steps, chain_backward
        _lib.check(_lib.load().vln_envdrop_flush(_lib.raw_stream()), "vln_envdrop_flush")
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
        if (self.ride_wgrads and side is None and self.grads_ready_hook is None and all(acc0) and len(runs) == 1
                and self.compute_dtype != torch.float32):
            # every gradient lands in place and nothing reads it before backward() returns: the launches are POSTED as a gradient
            # ride -- the encoder's BPTT launch (the next vln_lstm_seq_bwd on this stream) carries them on its idle CUs; the
            # callback issues them if no recurrence picked them up by the end of the backward pass
            with ops.GradRide.collect():
                self._issue_wgrads(runs, acc0, tgt)
            torch.autograd.Variable._execution_engine.queue_callback(ops.GradRide.flush)
            return ret
        with torch.cuda.stream(side) if side is not None else contextlib.nullcontext():
            self._issue_wgrads(runs, acc0, tgt)
        # Every decoder gradient is now final in its .grad (bucket view): a data-parallel caller starts their all-reduce
        # here so it overlaps the encoder's BPTT (dp.BucketReducer / FusedRMSprop.start_allreduce).
        if self.grads_ready_hook is not None and side is None and all(acc0):
            self.grads_ready_hook()
        return ret

    def _issue_wgrads(self, runs, acc0, tgt):
        H, F, AE = self.hidden_size, self.feature_size, self.action_embed_size
        (g_aw, g_ab, g_vin, g_ih, g_hh, g_bih, g_bhh, g_tin, g_tout, g_c) = tgt
        sb = self.compute_dtype != torch.float32       # bf16 mode: split-bf16 contractions (fp32 accumulation)
        for i, (bufs, r0, r1) in enumerate(runs):
            sl = slice(r0, r1)
            a = [x or i > 0 for x in acc0]      # the first run overwrites fresh buffers, everything else accumulates
            wb = ops.WgradBatch(sb)             # the seven products of a run: ONE launch in bf16 mode
            wb.add(bufs["dgates"][sl], bufs["xcat"][sl][:, :AE + F], g_ih, a[3])
            wb.add(bufs["dtv"][sl], bufs["hq"][sl], g_vin, a[2])
            wb.add(bufs["dtc"][sl], bufs["htd"][sl], g_c, a[9])
            wb.add(bufs["dgates"][sl], bufs["xcat"][sl][:, AE + F:], g_hh, a[4])
            wb.add(bufs["dz"][sl], bufs["tcat"][sl], g_tout, a[8])
            wb.add(bufs["dtt"][sl], bufs["tcat"][sl][:, H:], g_tin, a[7])
            wb.add(bufs["de"][sl], bufs["a"][sl], g_aw, a[0])
            wb.run()
            cb = ops.ColsumBatch()              # the three bias gradients: one launch (b_ih and b_hh share their sum)
            cb.add(bufs["de"][sl], g_ab, None, a[1])
            if a[5] == a[6]:
                cb.add(bufs["dgates"][sl], g_bih, g_bhh, a[5])
            else:
                cb.add(bufs["dgates"][sl], g_bih, None, a[5])
                cb.add(bufs["dgates"][sl], g_bhh, None, a[6])
            cb.run()

    def _forward_planned(self, plan, arena, entry, need_grad, gated, ctx_in, a_t_prev, img_feature, cand_feature,
                         h_tilde_prev, c_0, ctx, ctx_mask, img_lp, cand_lp, gather=None, smp=None, sbind=None):
        """The fast path of forward(): returns None (and leaves no trace) when anything the plan relies on moved."""
        if not arena.reserve(plan.i0, plan.n_alloc, plan.first_ptr):
            return None
        B, H = a_t_prev.shape[0], self.hidden_size
        lp = self.compute_dtype != torch.float32
        slot = None
        ok = True
        ctx_lp = entry.lp
        if lp:
            ok = ctx_lp is not None and ctx_lp.data_ptr() == plan.ctx_lp_ptr
        kctx = entry.kctx
        ok = ok and (kctx.data_ptr() if (kctx is not None and kctx is not False) else 0) == plan.kctx_ptr
        m8 = None
        if ok and ctx_mask is not None:
            m8 = entry.mask8 if entry.mask_src is ctx_mask else None
            if m8 is None and ctx_mask.dtype == torch.bool and ctx_mask.is_contiguous():
                m8 = ctx_mask.view(torch.uint8)
                entry.mask_src, entry.mask8 = ctx_mask, m8
            ok = m8 is not None and m8.data_ptr() == plan.mask_ptr
        if ok and need_grad:
            slot = self._stash.take(B, entry.ref)
            ok = slot.ptr("xcat") == plan.xcat_ptr
        if not ok:                    # fall back to the full path from a clean state
            arena.i = plan.i0
            if slot is not None:
                self._stash.untake(slot)
            return None
        arena.i = plan.i0 + plan.n_alloc
        self.plan_hits += 1
        io = _lib.EnvDropStep.from_buffer_copy(plan.io)
        io.offset = self._next_offset()
        if io.offset_base_dev:
            io.offset -= self._base_value
            io.offset_base_dev = self._base_ptr
        io.ws = self._step_ws(a_t_prev.device, plan.nws, io.chain != 0)
        keep = dict(plan.keep)
        keep["a"], keep["ctx"] = a_t_prev, ctx
        if gather is None:
            keep["img"], keep["cand"] = img_feature, cand_feature
        else:
            keep["gather"] = gather
        if lp:
            keep["ctx_lp"] = ctx_lp
            if img_lp is not None:
                keep["img_lp"], keep["cand_lp"] = img_lp, cand_lp
        if plan.kctx_ptr:
            keep["kctx"] = kctx
        if m8 is not None:
            keep["mask"] = m8
        if sbind is not None:
            self._fill_sampler(io, smp, sbind, keep)
        rec = _StepRec()
        rec.B, rec.L, rec.C, rec.H = B, ctx.shape[1], plan.dims.C, H
        rec.ctx_owner, rec.entry, rec.dims, rec.slot, rec.io, rec.keep = ctx, entry, plan.dims, slot, io, keep
        rec.mod, rec.dhtd_ext, rec.dhtd_keep, rec.flushed = self, None, None, False
        if need_grad:
            keep["htp"], keep["c0"] = h_tilde_prev.detach(), c_0.detach()
            logit, h1, c1, h_tilde = _EnvDropStepFn.apply(self, rec, h_tilde_prev, c_0, ctx_in, gated)
            logit._vln_rec = rec                  # lets losses.RolloutCE batch the logit branch of the backward
        else:
            keep["htp"], keep["c0"] = h_tilde_prev, c_0
            st = _lib.load().vln_envdrop_step_fwd(C.byref(plan.dims), C.byref(self._wstruct), C.byref(io), _lib.raw_stream())
            if st:
                _lib.check(st, "vln_envdrop_step_fwd")
            logit, h1, c1, h_tilde = keep["logit"], keep["h1"], keep["c1"], keep["h_tilde"]
        return logit, (h1, c1), h_tilde

    # ---- forward -----------------------------------------------------------------------------------------
    def forward(self, a_t_prev, img_feature, cand_feature, h_tilde_prev, h_0, c_0, ctx, ctx_mask=None,
                already_dropfeat=False, img_lp=None, cand_lp=None, gather=None, sampler=None):
        """Same contract as policy.py:208-246 (h_0 is unused there too).  img_feature / cand_feature are
        overwritten in place by the feature dropout, like the reference.  Extension (optional): `img_lp` / `cand_lp`
        = bf16 copies of the two feature tensors already produced by `DeviceFeatureStore.gather_*` (together with
        `already_dropfeat=True`): the decoder then skips its own dropout/copy pass over the features.
        Extension (optional): `gather=(store, rows, view_index, cand_rows, cand_views, cand_heading, cand_elevation)` with
        `img_feature = cand_feature = None`: the step reads its features from the HBM-resident table of a
        staging.DeviceFeatureStore itself (index vectors as for `store.gather_step`), applies the feature dropout of
        policy.py:226-231 on the way (its own Philox sites, exactly as on caller-given tensors) and does so in the SAME launch as
        its act-embedding / state prep -- one dependent launch less per step than `store.gather_step` + forward.
        Extension (optional): `sampler=(losses.RolloutSampler, cand_mask, action | None, action_host_address | 0)`: the sampled-action
        branch of envdrop.py:173,186-195 on this step's logits runs INSIDE the step's last launch (candidate dots + mask + softmax +
        draw + log-prob + entropy; vln_envdrop_step.s_*) and the step is recorded in the sampler as `sampler.step(logit, cand_mask,
        action)` would record it; the drawn action is `sampler.keep[-1][1]` and -- with `action_host_address`, the device-visible
        address of B int64 words of PINNED host memory -- also lands there without a copy launch (the host polls it)."""
        if sampler is not None:
            cm = sampler[1]
            Cn_ = cm.shape[1] if cm is not None else (gather[3].shape[1] if gather is not None else cand_feature.shape[1])
            if Cn_ > self.IN_STEP_SAMPLER_MAX_C:
                # both forms of the sampled branch (this step's in-launch draw and losses.RolloutSampler.step / vln_categorical_fwd) hold a
                # row's candidates in one wavefront; R2R's panoramas have at most ~15 navigable candidates.  Say so here, before any launch
                # of the step has gone out, instead of failing in the step's last launch.
                raise ValueError(f"EnvDropDecoder: the sampled-action branch takes at most {self.IN_STEP_SAMPLER_MAX_C} candidates per row, "
                                 f"got a batch padded to {Cn_}")
        self.__dict__["_sampler_arg"] = sampler          # (plain dict stores: nn.Module.__setattr__ costs microseconds per step)
        if gather is not None:
            if img_feature is not None or cand_feature is not None or already_dropfeat:
                raise TypeError("EnvDropDecoder: with gather=(store, ...) pass img_feature=None, cand_feature=None")
            if not a_t_prev.is_cuda:
                raise _lib.VlnError("EnvDropDecoder: tensors must be on the GPU; there is no CPU fallback")
            out = self._forward(a_t_prev, None, None, h_tilde_prev, c_0, ctx, ctx_mask, False, None, None, gather)
            if ops._arena is not None:
                ops.stamp(out[0], out[1][0], out[1][1], out[2])
            return out
        if not img_feature.is_cuda:
            raise _lib.VlnError("EnvDropDecoder: tensors must be on the GPU; there is no CPU fallback")
        if img_feature.dtype != torch.float32 or cand_feature.dtype != torch.float32:
            # bf16 feature tensors (DeviceFeatureStore.gather_*(want_f32=False)): they ARE the stream copies
            if not (already_dropfeat and img_feature.dtype == self.compute_dtype == cand_feature.dtype):
                raise TypeError("EnvDropDecoder: non-fp32 features need compute_dtype of that type and already_dropfeat=True")
            img_lp, cand_lp = img_feature, cand_feature
        out = self._forward(a_t_prev, img_feature, cand_feature, h_tilde_prev, c_0, ctx, ctx_mask, already_dropfeat, img_lp, cand_lp)
        if ops._arena is not None:         # generation stamps: a later module call refuses these once their memory is reused
            ops.stamp(out[0], out[1][0], out[1][1], out[2])
        return out

    def _forward(self, a_t_prev, img_feature, cand_feature, h_tilde_prev, c_0, ctx, ctx_mask, already_dropfeat, img_lp, cand_lp,
                 gather=None):
        if ops._arena is not None:
            ops.check_live(h_tilde_prev, "EnvDropDecoder(h_tilde_prev)"); ops.check_live(c_0, "EnvDropDecoder(c_0)")
            ops.check_live(ctx, "EnvDropDecoder(ctx)")
        if gather is not None:
            store, g_rows, g_vidx, g_crows, g_cviews, g_chead, g_celev = gather
            B, V, F = g_rows.shape[0], store.V, store.IMG + store.ANG
            Cn = g_crows.shape[1]
            dev = a_t_prev.device
            if F != self.feature_size or store.ANG != self.angle_feat_size:
                raise ValueError("EnvDropDecoder: the feature store's row layout does not match the decoder's feature_size / angle_feat_size")
        else:
            B, V, F = img_feature.shape
            Cn = cand_feature.shape[1]
            dev = img_feature.device
        L = ctx.shape[1]
        H, AE, ANG = self.hidden_size, self.action_embed_size, self.angle_feat_size
        P = self._gated_params()
        need_grad = torch.is_grad_enabled() and (
            h_tilde_prev.requires_grad or c_0.requires_grad or ctx.requires_grad or P[3].requires_grad)
        gated = self._ensure_current(need_grad, P)
        ctx_in, entry = self._gated_ctx(ctx, need_grad)
        dt = self.compute_dtype
        lp = dt != torch.float32
        # the in-step sampler's buffers come first (the same place in the arena's order on the planned and on the full path)
        smp = self.__dict__.get("_sampler_arg")
        if smp is not None:
            self.__dict__["_sampler_arg"] = None
        sbind = None
        if smp is not None:
            if self.defer_logits:
                raise ValueError("EnvDropDecoder: sampler= needs the step's logits (defer_logits = False)")
            sbind = smp[0].bind(B, Cn, dev, smp[2])
        # backward-only chaining (chain_backward): this step follows the previous call's step
        follows = bool(self.chain_backward and need_grad and not self.defer_logits and self._last_ht is not None
                       and self._last_ht() is h_tilde_prev)

        # Step plans (arena mode): with address-stable buffers, the n-th step of an iteration sees the argument block of
        # the n-th step two iterations earlier.  The filled C struct, the output tensors and the saved-activation
        # block are then reused as they are (only the dropout offset and the per-rollout objects change) instead of
        # being rebuilt: ~40 us of Python per step, on a path where the host, not the GPU, sets the pace at B = 64.
        arena = ops.current_arena() if self.step_graphs else None
        pkey = None
        if arena is not None:
            if gather is not None:      # the index vectors stand in for the feature tensors
                fk = (g_rows.data_ptr(), g_crows.data_ptr(), g_vidx.data_ptr(), g_cviews.data_ptr(), g_chead.data_ptr(),
                      g_celev.data_ptr(), store.table.data_ptr())
            else:
                fk = (img_feature.data_ptr(), cand_feature.data_ptr())
            pkey = (arena.g, arena.i, a_t_prev.data_ptr(), fk, h_tilde_prev.data_ptr(),
                    c_0.data_ptr(), ctx.data_ptr(), 0 if ctx_mask is None else ctx_mask.data_ptr(), B, V, F, Cn, L, need_grad,
                    self.training, bool(already_dropfeat), 0 if img_lp is None else img_lp.data_ptr(),
                    0 if cand_lp is None else cand_lp.data_ptr(), dt, bool(self.defer_logits), bool(self.chain_steps), bool(self.project_context), follows,
                    None if smp is None else (0 if smp[1] is None else smp[1].data_ptr(), int(smp[3] or 0), smp[2] is not None))
        ctx_lp = None
        if lp:                         # once per rollout; BEFORE the step's own buffers so the arena order is the same on
            ctx_lp = entry.lp          # the planned and on the full path
            if ctx_lp is None:
                ctx_lp = self._ctx_lp(entry, ctx, dt)
        kctx = entry.kctx              # the projected context, likewise once per rollout
        if kctx is None:
            kctx = self._ctx_k(entry, ctx, lp)
        if pkey is not None:
            plan = self._plans.get(pkey)
            if plan is not None:
                out = self._forward_planned(plan, arena, entry, need_grad, gated, ctx_in, a_t_prev, img_feature, cand_feature,
                                            h_tilde_prev, c_0, ctx, ctx_mask, img_lp, cand_lp, gather, smp, sbind)
                if out is not None:
                    return self._step_done(out, smp)
        arena_i0 = arena.i if arena is not None else 0

        rec = _StepRec()
        rec.B, rec.L, rec.C, rec.H = B, L, Cn, H
        rec.ctx_owner, rec.entry = ctx, entry
        rec.mod, rec.dhtd_ext, rec.dhtd_keep, rec.flushed = self, None, None, False
        # the dims block and its scratch size depend on the shapes only: built once per shape
        dk = (B, L, V, Cn, lp)
        cached = self._dims_cache.get(dk)
        if cached is None:
            d = _lib.EnvDropDims(B, L, V, Cn, H, F - ANG, ANG, AE, ops.BF16 if lp else ops.F32, ops.BF16 if lp else ops.F32)
            nws = _lib.load().vln_envdrop_ws_floats(C.byref(d))
            if nws < 0:
                _lib.check(int(nws), "vln_envdrop_ws_floats")
            if len(self._dims_cache) > 64:
                self._dims_cache.clear()
            cached = self._dims_cache[dk] = (d, nws)
        d, nws = cached
        rec.dims = d
        if gather is not None:          # the step's own feature buffers: fp32 rows (fp32 compute) or only the bf16 stream rows
            img = None if lp else ops.empty(B, V, F, dtype=torch.float32, device=dev)
            cand = None if lp else ops.empty(B, Cn, F, dtype=torch.float32, device=dev)
        else:
            img = img_feature if img_feature.is_contiguous() else img_feature.contiguous()
            cand = cand_feature if cand_feature.is_contiguous() else cand_feature.contiguous()
        a = a_t_prev if a_t_prev.is_contiguous() else a_t_prev.contiguous()
        htp = h_tilde_prev.detach()
        if not htp.is_contiguous():
            htp = htp.contiguous()
        c0 = c_0.detach()
        if not c0.is_contiguous():
            c0 = c0.contiguous()
        ctxc = ctx.detach()
        if not ctxc.is_contiguous():
            ctxc = ctxc.contiguous()
        logit = ops.empty(B, Cn, dtype=torch.float32, device=dev)
        h1 = ops.empty(B, H, dtype=torch.float32, device=dev)
        c1 = ops.empty(B, H, dtype=torch.float32, device=dev)
        h_tilde = ops.empty(B, H, dtype=torch.float32, device=dev)
        keep = {"img": img, "cand": cand, "a": a, "htp": htp, "c0": c0, "ctx": ctxc,
                "logit": logit, "h1": h1, "c1": c1, "h_tilde": h_tilde}
        io = _lib.EnvDropStep()
        # per-step saved activations: ONE flat allocation, addressed by offset (no tensor views on this path)
        n_e, n_av, n_g, n_h, n_at = B * AE, B * V, B * 4 * H, B * H, B * L
        if need_grad:
            slot = self._stash.take(B, entry.ref)
            rec.slot = slot
            flat = ops.empty(4 + n_e + n_av + n_g + 2 * n_h + n_at, dtype=torch.float32, device=dev)
            io.a_stash, io.hq, io.xcat, io.tcat, io.htd = (slot.ptr("a"), slot.ptr("hq"), slot.ptr("xcat"), slot.ptr("tcat"),
                                                           slot.ptr("htd"))
        else:
            rec.slot = None
            XK = AE + F + H
            flat = ops.empty(4 + n_e + n_av + n_g + 2 * n_h + n_at + B * (H + XK + 2 * H + H), dtype=torch.float32, device=dev)
            q = flat.data_ptr() + 4 * (4 + n_e + n_av + n_g + 2 * n_h + n_at)
            io.hq, io.xcat, io.tcat, io.htd = q, q + 4 * B * H, q + 4 * B * (H + XK), q + 4 * B * (H + XK + 2 * H)
        keep["flat"] = flat
        q = flat.data_ptr()
        # offsets relative to a device word: per-step graphs (arena) and runtime.DeviceClock (whole-iteration graphs)
        base_mode = (self.step_graphs or self.__dict__.get("clock") is not None) and need_grad
        if self.step_graphs and not base_mode:     # the step's dropout offset lives on the device: launch arguments repeat
            io.offset_dev = q
        q += 16
        io.e = q; q += 4 * n_e
        io.alpha_v = q; q += 4 * n_av
        io.gate_act = q; q += 4 * n_g
        io.tanh_c1 = q; q += 4 * n_h
        io.tt = q; q += 4 * n_h
        io.alpha_t = q
        io.a_prev, io.img, io.cand = a.data_ptr(), _tp(img), _tp(cand)
        if gather is not None:
            for t_, nm in ((g_rows, "rows"), (g_vidx, "view_index"), (g_crows, "cand_rows"), (g_cviews, "cand_views"),
                           (g_chead, "cand_heading"), (g_celev, "cand_elevation")):
                if not (t_.is_cuda and t_.is_contiguous()):
                    raise ValueError(f"EnvDropDecoder gather: {nm} must be a contiguous device tensor")
            if g_rows.dtype != torch.int64 or g_crows.dtype != torch.int64 or g_vidx.dtype != torch.int32 or g_cviews.dtype != torch.int32:
                raise TypeError("EnvDropDecoder gather: rows int64, view indices int32")
            io.g_table, io.g_angle_table = store.table.data_ptr(), store.angle_table.data_ptr()
            io.g_rows, io.g_vidx, io.g_crows, io.g_cviews = g_rows.data_ptr(), g_vidx.data_ptr(), g_crows.data_ptr(), g_cviews.data_ptr()
            io.g_chead, io.g_celev = g_chead.data_ptr(), g_celev.data_ptr()
            io.g_ttype = ops._dt(store.table)
            keep["gather"] = gather
        if lp:
            ready = already_dropfeat and img_lp is not None and cand_lp is not None and img_lp.dtype == dt and \
                cand_lp.dtype == dt and img_lp.is_contiguous() and cand_lp.is_contiguous()
            if not ready:
                img_lp = ops.empty(B, V, F, dtype=dt, device=dev)
                cand_lp = ops.empty(B, Cn, F, dtype=dt, device=dev)
            else:
                io.lp_ready = 1
            keep["img_lp"], keep["cand_lp"], keep["ctx_lp"] = img_lp, cand_lp, ctx_lp
            io.img_lp, io.cand_lp, io.ctx_lp = img_lp.data_ptr(), cand_lp.data_ptr(), ctx_lp.data_ptr()
        if self.split_attention:
            sy = self._attn_sync_buf(dev, B)
            io.attn_sync, io.attn_sync_bytes = sy.data_ptr(), sy.numel() * 4
        io.h_tilde_prev, io.c0, io.ctx = htp.data_ptr(), c0.data_ptr(), ctxc.data_ptr()
        if kctx is not False:
            io.kctx = kctx.data_ptr()
            keep["kctx"] = kctx
        if ctx_mask is not None:
            m8 = entry.mask8 if entry.mask_src is ctx_mask else None
            if m8 is None:
                if ctx_mask.dtype == torch.bool and ctx_mask.is_contiguous():
                    m8 = ctx_mask.view(torch.uint8)          # shares storage: remembered for the rollout's later steps
                    entry.mask_src, entry.mask8 = ctx_mask, m8
                else:
                    m8 = ctx_mask.to(torch.uint8).contiguous()
            keep["mask"] = m8
            io.ctx_mask = m8.data_ptr()
        io.logit, io.h1, io.c1, io.h_tilde = logit.data_ptr(), h1.data_ptr(), c1.data_ptr(), h_tilde.data_ptr()
        io.seed, io.offset = self.dropout_seed, self._next_offset()
        if base_mode:                  # offsets relative to the gate's base word: no per-step device write (runtime._rebase_offsets)
            io.offset -= self._base_value
            io.offset_base_dev = self._base_ptr
        if self.training:
            io.p_drop, io.p_feat = self.drop_ratio, self.feat_drop_ratio
        if already_dropfeat:
            io.already_dropfeat = 1
        if self.defer_logits and need_grad:
            io.defer_logits = 1
            if self.chain_steps:
                io.chain = 3
        elif follows:
            io.chain = 2
        if sbind is not None:
            self._fill_sampler(io, smp, sbind, keep)
        io.ws, io.ws_floats = self._step_ws(dev, nws, io.chain != 0), nws
        rec.io, rec.keep = io, keep
        if (pkey is not None and (gather is not None or (img is img_feature and cand is cand_feature)) and a is a_t_prev and htp.is_contiguous()
                and h_tilde_prev.is_contiguous() and c_0.is_contiguous() and ctx.is_contiguous()):
            if len(self._plans) > 256:
                self._plans.clear()
            first = img if (gather is not None and not lp) else logit          # the first buffer the step took from the arena
            self._plans[pkey] = _StepPlan(io, d, nws, keep, arena_i0, arena.i - arena_i0,
                                          slot.ptr("xcat") if need_grad else 0, keep["ctx_lp"].data_ptr() if lp else 0,
                                          io.ctx_mask, first.data_ptr(), gathered=gather is not None, kctx_ptr=io.kctx or 0)

        if need_grad:
            logit, h1, c1, h_tilde = _EnvDropStepFn.apply(self, rec, h_tilde_prev, c_0, ctx_in, gated)
            logit._vln_rec = rec                      # lets losses.RolloutCE batch the logit branch of the backward
        else:
            st = _lib.load().vln_envdrop_step_fwd(C.byref(d), C.byref(self._wstruct), C.byref(io), _lib.raw_stream())
            if st:
                _lib.check(st, "vln_envdrop_step_fwd")
        if gather is None:
            if img is not img_feature:
                img_feature.copy_(img)
            if cand is not cand_feature:
                cand_feature.copy_(cand)
        return self._step_done((logit, (h1, c1), h_tilde), smp)

    @staticmethod
    def _fill_sampler(io, smp, sbind, keep):
        probs, act, logp_p, ent_p, seed, off, base = sbind
        cm = smp[1]
        if cm is not None:
            cm = cm.view(torch.uint8) if (cm.dtype == torch.bool and cm.is_contiguous()) else cm.to(torch.uint8).contiguous()
            keep["s_mask"] = cm
        io.s_cand_mask = _tp(cm)
        io.s_action_in = act.data_ptr() if smp[2] is not None else None
        io.s_action_out = None if smp[2] is not None else act.data_ptr()
        io.s_action_host = int(smp[3] or 0) or None
        io.s_probs, io.s_logp, io.s_ent = probs.data_ptr(), logp_p, ent_p
        io.s_seed, io.s_offset, io.s_offset_base_dev = seed, off, base
        keep["s_probs"], keep["s_act"] = probs, act

    def _step_done(self, out, smp):
        """Bookkeeping after a step was issued: the in-step sampler records it; the step's h_tilde is remembered (chain_backward)."""
        if smp is not None:
            smp[0].commit(out[0])
        if self.chain_backward:
            self.__dict__["_last_ht"] = weakref.ref(out[2])
        elif self._last_ht is not None:
            self.__dict__["_last_ht"] = None
        return out


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
        # Tests only: a list that receives every call's hidden pre-activations' SIGNS (bool [rows, H], ReLU on / off) -- the
        # decisions a restatement has to share to be compared beyond them (a ReLU is a discontinuity, like a dropout mask)
        self.relu_record = None

    def forward(self, state):
        ops.check_live(state, "Critic(state)")
        p = self.drop_ratio if self.training else 0.0
        l0, l3 = self.state2value[0], self.state2value[3]
        clock = self.__dict__.get("clock")
        if clock is not None:          # runtime.DeviceClock: the mask's offset is (device word + call index since the tick) * 8
            r = clock.rel(id(self))
            self._calls = clock.value(r) * 8         # the offset `ops.dropout_mask` reproduces the mask with (tests)
            off = (r * 8, clock.ptr)
        else:
            self._calls += 1
            off = (self._calls, None)
        return _CriticFn.apply(state, l0.weight, l0.bias, l3.weight, l3.bias, p, self.dropout_seed, off, self.relu_record).squeeze()


class _CriticFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w0, b0, w3, b3, p, seed, off, record=None):
        x = x.contiguous()
        z = ops.linear_fwd(x, w0.detach(), b0.detach(), ops.ACT_RELU)
        if record is not None:
            record.append(z > 0)
        # the mask is regenerated by the kernels from (seed, offset) in the backward: z * mask and dzd * mask are one launch each
        zd = ops.scale_dropout(z, seed, off[0], p, off[1]) if p > 0 else z
        v = ops.linear_fwd(zd, w3.detach(), b3.detach())
        ctx.save_for_backward(x, w0, w3, z, zd)
        ctx.drop = (seed, off, p)
        return v

    @staticmethod
    def backward(ctx, dv):
        x, w0, w3, z, zd = ctx.saved_tensors
        seed, off, p = ctx.drop
        dv = dv.contiguous()
        dzd = ops.linear_fwd(dv, w3.detach().t().contiguous())            # [B,1] x [H,1]^T
        if p > 0:
            dzd = ops.scale_dropout(dzd, seed, off[0], p, off[1])
        dz = dzd * (z > 0).to(dzd.dtype)
        dw3 = ops.linear_wgrad(dv, zd)
        db3 = dv.sum(0)
        dx = ops.linear_fwd(dz, ops.transpose_cast(w0.detach()))
        dw0 = ops.linear_wgrad(dz, x)
        db0 = ops.colsum(dz)
        return dx, dw0, db0, dw3, db3, None, None, None, None
