"""Drop-in `EncoderLSTM` (reference: src/model/units.py:12-74).

Constructor, `forward(inputs, lengths, already_sorted=True) -> (ctx, decoder_init, c_t)` and the
`state_dict` keys (`embedding.weight`, `lstm.weight_ih_l{k}[_reverse]`, ..., `enc2dec.{weight,bias}`)
are the reference's; `nn.Embedding` / `nn.LSTM` / `nn.Linear` objects are kept only as parameter
holders (default init identical to the reference) -- their forward is never called.

Algorithm (see csrc/encoder.hip): time-major internal layout, one input-projection GEMM per layer for all
time steps, L fused recurrence launches per layer (both directions each), one BPTT launch per step, and
ONE weight-gradient contraction over (L x B) per weight.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from . import _lib, ops
from .runtime import ShadowSet

_p = ops._p


def _stream():
    return _lib.raw_stream()


class _Off:
    """Where a call's dropout offsets and launch sequence come from.  Host form: `v` = the module's call counter, the site
    offset is v * 8 + k, the recurrence counts its launches on the host.  Clock form (runtime.DeviceClock): `v` = the call's
    small index since the last tick, `base` = the clock's device word (the kernels form (word + v) * 8 + k), `seq` = the
    launch's index into the device-resident launch sequence (forward 2v, backward 2v + 1)."""
    __slots__ = ("v", "base", "seq")

    def __init__(self, v, base=None, seq=-1):
        self.v, self.base, self.seq = int(v), base, int(seq)

    def site(self, k):
        return self.v * 8 + k


class _EncoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, tokens, lens32, p_drop, offset, *params):
        lib = _lib.load()
        lib.vln_posted_drop()          # (see backward)
        ride, mod._ride = mod.__dict__.get("_ride"), None
        sh = mod._shadow.t
        B, L = tokens.shape
        dev = tokens.device
        Hd, dirs, nl = mod.hidden_size, mod.num_directions, mod.num_layers
        E = mod.embed_size
        seed = mod.dropout_seed
        f32 = dict(dtype=torch.float32, device=dev)
        wtype = ops.BF16 if mod.compute_dtype == torch.bfloat16 else ops.F32
        p_emb = 0.0 if mod.use_glove else p_drop
        x = ops.empty(L * B, E, **f32)
        _lib.check(lib.vln_embed_fwd(_p(tokens), _p(mod.embedding.weight), _p(x), B, L, E, seed, offset.site(0),
                                     p_emb, offset.base, _stream()), "vln_embed_fwd")
        saved = []
        p_inter = p_drop if nl > 1 else 0.0
        for k in range(nl):
            sync_p, sync_n = mod._sync_ws(dev, B, Hd, dirs)
            w_ih_k = sh[f"w_ih{k}"]
            # round 6: the input projection inside the persistent recurrence launch where the library takes it (Hd 256 / 512, 256 input
            # features, the granule-protocol launch): no projection GEMM, no [L*B, dirs*4Hd] intermediate written and re-read
            inproj = bool(mod.inproj and x.is_contiguous() and w_ih_k.is_contiguous() and
                          lib.vln_lstm_inproj_ok(B, L, Hd, dirs, x.shape[1], sync_p, sync_n))
            xproj = None if inproj else ops.linear_fwd(x, w_ih_k, sh[f"bsum{k}"])
            hprev = ops.empty(dirs, L, B, Hd, **f32)
            cprev = ops.empty(dirs, L, B, Hd, **f32)
            y = ops.empty(L * B, dirs * Hd, **f32)
            act = ops.empty(L * B, dirs * 4 * Hd, **f32)
            tanh_c = ops.empty(L * B, dirs * Hd, **f32)
            hcat = ops.empty(B, dirs * Hd, **f32)
            ccat = ops.empty(B, dirs * Hd, **f32)
            ride_p = C.byref(ride.struct) if (ride is not None and k == nl - 1) else None
            if inproj:
                _lib.check(lib.vln_lstm_seq_fwd_x(_p(x), x.shape[1], _p(w_ih_k), _p(sh[f"bsum{k}"]), _p(sh[f"w_hh{k}"]), wtype, _p(lens32),
                                                  _p(hprev), _p(cprev), _p(y), _p(act), _p(tanh_c), _p(hcat), _p(ccat), B, L, Hd, dirs, None, None,
                                                  sync_p, sync_n, offset.seq, ride_p, _stream()), "vln_lstm_seq_fwd_x")
            else:
                _lib.check(lib.vln_lstm_seq_fwd(_p(xproj), _p(sh[f"w_hh{k}"]), wtype, _p(lens32), _p(hprev), _p(cprev),
                                                _p(y), _p(act), _p(tanh_c), _p(hcat), _p(ccat), B, L, Hd, dirs, None, None,
                                                sync_p, sync_n, offset.seq, ride_p, _stream()), "vln_lstm_seq_fwd")
            saved.append((x, hprev, cprev, act, tanh_c))
            if k < nl - 1:
                if p_inter > 0:
                    xn = ops.empty_like(y)
                    _lib.check(lib.vln_scale_dropout(_p(y), y.stride(0), _p(xn), xn.stride(0), L * B, dirs * Hd, seed,
                                                     offset.site(2 + k), p_inter, offset.base, _stream()), "vln_scale_dropout")
                    x = xn
                else:
                    x = y
        H = dirs * Hd
        ctx_out = ops.empty(B, L, H, **f32)
        # bf16 mode: the stream copy of the context the decoders' attention reads comes out of the same pass
        ctx_lp = ops.empty(B, L, H, dtype=torch.bfloat16, device=dev) if wtype == ops.BF16 else None
        if mod.layout_with_bridge:
            # the context's layout change (HBM-bound) and the bridge product tanh(enc2dec(h_t)) (32 workgroups of latencies) do not
            # depend on each other: the copy is POSTED and rides in the product's launch (vln_layout_post)
            _lib.check(lib.vln_layout_post(0, _p(y), _p(ctx_out), _p(ctx_lp), B, L, H, seed, offset.site(1), p_drop, offset.base),
                       "vln_layout_post")
        else:
            _lib.check(lib.vln_tm_to_bm(_p(y), _p(ctx_out), _p(ctx_lp), B, L, H, seed, offset.site(1), p_drop, offset.base, _stream()),
                       "vln_tm_to_bm")
        mod._last_ctx_lp = ctx_lp
        dec_init = ops.linear_fwd(hcat, sh["w_e2d"], mod.enc2dec.bias.detach(), ops.ACT_TANH)
        if mod.layout_with_bridge:
            _lib.check(lib.vln_layout_post_flush(_stream()), "vln_layout_post_flush")      # (its own launch if the product took another kernel)
        # dec_init is an OUTPUT: keeping the returned object on ctx would close a reference cycle through its grad_fn
        # (tensor -> node -> ctx -> tensor) that only the cyclic GC can free -- ~85 MB of activations per iteration
        ctx.mod, ctx.saved, ctx.misc = mod, saved, (tokens, lens32, p_drop, offset, hcat, dec_init.detach(), wtype)
        ctx.set_materialize_grads(False)
        return ctx_out, dec_init, ccat

    @staticmethod
    def backward(ctx, dctx, ddec, dct):
        lib = _lib.load()
        lib.vln_posted_drop()          # whatever an aborted earlier call left posted (between its post and its flush) is forgotten
        mod = ctx.mod
        sh = mod._shadow.t
        tokens, lens32, p_drop, offset, hcat, dec_init, wtype = ctx.misc
        B, L = tokens.shape
        dev = tokens.device
        Hd, dirs, nl = mod.hidden_size, mod.num_directions, mod.num_layers
        H = dirs * Hd
        seed = mod.dropout_seed
        f32 = dict(dtype=torch.float32, device=dev)
        p_emb = 0.0 if mod.use_glove else p_drop
        p_inter = p_drop if nl > 1 else 0.0
        grads = {}
        mod._params_cached()
        pmap = mod._pmap

        sb = mod.compute_dtype != torch.float32     # bf16 mode: weight gradients on the split-bf16 MFMA form

        def wgrad(dy_, x_, out=None, accumulate=False):
            return ops.linear_wgrad(dy_, x_, out, accumulate, sb)

        def put(name, fn, *args):
            """Form one gradient: straight into an existing contiguous .grad (e.g. a dp.GradBucket view, autograd
            then gets None) or into a fresh tensor handed to autograd."""
            p = pmap[name]
            if not p.requires_grad:
                return
            if fn is ops.linear_wgrad:
                fn = wgrad
            g = p.grad
            if g is not None and g.is_contiguous() and g.dtype == torch.float32:
                fn(*args, g, True)
            else:
                grads[name] = fn(*args)

        # decoder_init = tanh(enc2dec(h_t))                                       units.py:69
        if ddec is not None:
            dpre = ops.ew(ops.EW_TANH_GRAD, ddec if ddec.stride(-1) == 1 else ddec.contiguous(), dec_init)      # one launch
            # d enc2dec.weight: in bf16 mode on the packed grouped kernel (the arithmetic of every other weight gradient) -- and, when a
            # gradient ride is pending on this stream (the decoder's, posted before the encoder's backward began), as one more product
            # of that ride: the backward recurrence launch below carries it instead of a 13 us launch in front of it
            pw = pmap["enc2dec.weight"]
            if pw.requires_grad:
                gw = pw.grad
                inplace = gw is not None and gw.is_contiguous() and gw.dtype == torch.float32
                if not (sb and inplace and ops.GradRide.try_add(dpre, hcat, gw, sb)):
                    if sb:
                        tgt = gw if inplace else torch.empty_like(pw)
                        wb0 = ops.WgradBatch(sb)
                        wb0.add(dpre, hcat, tgt, inplace)
                        wb0.run()
                        if not inplace:
                            grads["enc2dec.weight"] = tgt
                    else:
                        put("enc2dec.weight", ops.linear_wgrad, dpre, hcat)
            put("enc2dec.bias", ops.colsum, dpre)
            # the context gradient's layout change rides in the launch of d h_t = dpre W_e2d (as the forward's does)
            dy_posted = None
            if dctx is not None and mod.layout_with_bridge:
                dy_posted = ops.empty(L * B, H, **f32)
                dctx = dctx.contiguous()
                _lib.check(lib.vln_layout_post(1, _p(dctx), _p(dy_posted), None, B, L, H, seed, offset.site(1), p_drop, offset.base),
                           "vln_layout_post")
            dhcat = ops.linear_fwd(dpre, sh["w_e2d_t"])
            if dy_posted is not None:
                _lib.check(lib.vln_layout_post_flush(_stream()), "vln_layout_post_flush")
        else:
            dhcat = ops.zeros(B, H, **f32)
            dy_posted = None
        dccat = dct.contiguous() if dct is not None else ops.zeros(B, H, **f32)
        dy = dy_posted
        if dctx is not None and dy is None:
            dy = ops.empty(L * B, H, **f32)
            dctx = dctx.contiguous()
            _lib.check(lib.vln_bm_to_tm(_p(dctx), _p(dy), B, L, H, seed, offset.site(1), p_drop, offset.base, _stream()),
                       "vln_bm_to_tm")
        for k in range(nl - 1, -1, -1):
            x, hprev, cprev, act, tanh_c = ctx.saved[k]
            if k == nl - 1:   # only the last layer's final states are returned (units.py:63-67): their gradients go in as
                # they are ([B, dirs*Hd]); the recurrence reads them in that layout, dh_pass / dc_carry are scratch
                dh_pass = ops.empty(dirs, B, Hd, **f32)
                dc_carry = ops.empty(dirs, B, Hd, **f32)
                init = (dhcat if dhcat.is_contiguous() else dhcat.contiguous(), dccat)
            else:
                dh_pass = ops.zeros(dirs, B, Hd, **f32)
                dc_carry = ops.zeros(dirs, B, Hd, **f32)
                init = (None, None)
            dgates = ops.empty(L * B, dirs * 4 * Hd, **f32)
            # the bias gradients' partial sums (one row per block of 16 episodes) come out of the recurrence itself: its threads
            # hold every dgates value they store; the column sum below then reads [B / 16, 4 Hd] instead of [L * B, 4 Hd]
            nbb = (B + 15) // 16
            bias_part = ops.empty(dirs, nbb, 4 * Hd, **f32)
            sync_p, sync_n = mod._sync_ws(dev, B, Hd, dirs)
            seq_b = offset.seq + 1 if offset.seq >= 0 else -1
            # round 6: the layer's own weight gradients accumulated INSIDE the persistent BPTT launch (four extra waves per workgroup)
            # where the library takes it: no pack launch, no contraction launch over the L * B rows
            E_in = x.shape[1]
            inl = bool(mod.wgrad_inlaunch and sb and x.is_contiguous() and ops.wgrad_precision(True) == 2 and
                       all(pmap[f"lstm.weight_{w_}_l{k}" + sf_].requires_grad for w_ in ("hh", "ih") for sf_ in [""] + (["_reverse"] if dirs == 2 else [])) and
                       lib.vln_lstm_wgrad_inlaunch_ok(B, L, Hd, dirs, E_in, wtype, 2, sync_p, sync_n))
            if inl:
                part = ops.empty(int(lib.vln_lstm_wgrad_part_floats(B, Hd, dirs, E_in)), **f32)
                _lib.check(lib.vln_lstm_seq_bwd_w(_p(dy), _p(sh[f"w_hh_t{k}"]), wtype, _p(lens32), _p(act), _p(tanh_c), _p(cprev), _p(dgates),
                                                  _p(dh_pass), _p(dc_carry), _p(init[0]), _p(init[1]), B, L, Hd, dirs, sync_p, sync_n, seq_b,
                                                  _p(bias_part), _p(x), E_in, _p(hprev), _p(part), part.numel(), _stream()), "vln_lstm_seq_bwd_w")
                o_hh, o_ih = (C.c_void_p * 2)(), (C.c_void_p * 2)()
                a_hh, a_ih = (C.c_int * 2)(), (C.c_int * 2)()
                for d in range(dirs):
                    sfx = f"_l{k}" + ("_reverse" if d == 1 else "")
                    for name, outs_, accs_ in (("lstm.weight_hh" + sfx, o_hh, a_hh), ("lstm.weight_ih" + sfx, o_ih, a_ih)):
                        p = pmap[name]
                        g = p.grad
                        if g is not None and g.is_contiguous() and g.dtype == torch.float32:
                            outs_[d], accs_[d] = g.data_ptr(), 1
                        else:
                            grads[name] = torch.empty_like(p)
                            outs_[d], accs_[d] = grads[name].data_ptr(), 0
                _lib.check(lib.vln_lstm_wgrad_reduce(_p(part), B, Hd, dirs, E_in, o_hh, o_ih, a_hh, a_ih, _stream()), "vln_lstm_wgrad_reduce")
            else:
                _lib.check(lib.vln_lstm_seq_bwd(_p(dy), _p(sh[f"w_hh_t{k}"]), wtype, _p(lens32), _p(act), _p(tanh_c),
                                                _p(cprev), _p(dgates), _p(dh_pass), _p(dc_carry), _p(init[0]), _p(init[1]), B, L, Hd, dirs,
                                                sync_p, sync_n, seq_b, _p(bias_part), _stream()),
                           "vln_lstm_seq_bwd")
            cbt = ops.ColsumBatch()       # ... and its bias gradients
            wb = ops.WgradBatch(sb)       # the layer's weight gradients (all over the same L*B rows): one launch in bf16 mode
            for d in range(dirs):
                sfx = f"_l{k}" + ("_reverse" if d == 1 else "")
                dg = dgates[:, d * 4 * Hd:(d + 1) * 4 * Hd]
                for name, xop in (() if inl else (("lstm.weight_hh" + sfx, hprev[d].view(L * B, Hd)), ("lstm.weight_ih" + sfx, x))):
                    p = pmap[name]
                    if not p.requires_grad:
                        continue
                    g = p.grad
                    if g is not None and g.is_contiguous() and g.dtype == torch.float32:
                        wb.add(dg, xop, g, True)
                    else:
                        grads[name] = torch.empty_like(p)
                        wb.add(dg, xop, grads[name], False)
                outs = []                                                     # both biases feed the same pre-activation
                for name in ("lstm.bias_ih" + sfx, "lstm.bias_hh" + sfx):
                    p = pmap[name]
                    if not p.requires_grad:
                        continue
                    g = p.grad
                    if g is not None and g.is_contiguous() and g.dtype == torch.float32:
                        outs.append((g, True))
                    else:
                        grads[name] = torch.empty_like(p)
                        outs.append((grads[name], False))
                bp = bias_part[d]                                             # [B / 16, 4 Hd]: this direction's partial sums
                if len(outs) == 2 and outs[0][1] == outs[1][1]:
                    cbt.add(bp, outs[0][0], outs[1][0], outs[0][1])           # one pass over the partials, two destinations
                else:
                    for o, acc in outs:
                        cbt.add(bp, o, None, acc)
            need_dx = (k > 0) or mod.embedding.weight.requires_grad
            # d x = dgates W_ih and the pack of the same dgates for the weight gradients do not depend on each other: the product is
            # POSTED and rides in the batch's pack launch (ops.linear_fwd_post; 36 + 21 us as two launches at B = 64, L = 80)
            dx = None
            if need_dx and mod.dx_with_wgrads and wb.jobs:
                w_t = sh[f"w_ih_t{k}"]
                dx = ops.linear_fwd_post(dgates, w_t, ops.empty(L * B, w_t.shape[0], **f32))
            cs_posted = dx is not None and cbt.post()                   # the bias gradients' column sums: the same launch
            wb.run()
            if dx is not None:
                ops.linear_fwd_post_flush(dev, L * B, dx.shape[1])      # (issues it alone if the batch took another form)
            if cs_posted:
                cbt.flush()
            else:
                cbt.run()
            if need_dx:
                if dx is None:
                    dx = ops.linear_fwd(dgates, sh[f"w_ih_t{k}"])
                if k > 0:
                    if p_inter > 0:
                        dy = ops.empty_like(dx)
                        _lib.check(lib.vln_scale_dropout(_p(dx), dx.stride(0), _p(dy), dy.stride(0), L * B, dx.shape[1],
                                                         seed, offset.site(2 + (k - 1)), p_inter, offset.base, _stream()),
                                   "vln_scale_dropout")
                    else:
                        dy = dx
                else:
                    ge = mod.embedding.weight.grad
                    inplace = ge is not None and ge.is_contiguous() and ge.dtype == torch.float32
                    dE = ge if inplace else torch.zeros_like(mod.embedding.weight)
                    pad = mod.embedding.padding_idx
                    if mod.deterministic_embedding_grad and mod.embed_size <= 1024:      # fixed summation order, no float atomics
                        _lib.check(lib.vln_embed_bwd_det(_p(tokens), _p(lens32), _p(dx), _p(dE), B, L, mod.embed_size,
                                                         mod.embedding.num_embeddings, -1 if pad is None else pad, seed,
                                                         offset.site(0), p_emb, offset.base, _stream()), "vln_embed_bwd_det")
                    else:
                        _lib.check(lib.vln_embed_bwd(_p(tokens), _p(lens32), _p(dx), _p(dE), B, L, mod.embed_size,
                                                     -1 if pad is None else pad, seed, offset.site(0), p_emb, offset.base, _stream()),
                                   "vln_embed_bwd")
                    if not inplace:
                        grads["embedding.weight"] = dE
        out = [grads.get(n) for n in mod._param_names]
        return (None, None, None, None, None) + tuple(out)


class EncoderLSTM(nn.Module):
    def __init__(self, vocab_size, embed_size, hidden_size, padding_idx, drop_ratio=0.5, bidirectional=False,
                 num_layers=1, glove=None, compute_dtype=torch.float32):
        super().__init__()
        self.embed_size = embed_size
        self.vocab_size = vocab_size
        self.num_directions = 2 if bidirectional else 1
        self.hidden_size = hidden_size // self.num_directions
        self.num_layers = num_layers
        self.use_glove = glove is not None
        self.drop_ratio = float(drop_ratio)
        self.drop = nn.Dropout(p=drop_ratio)
        if self.use_glove:
            self.embedding = nn.Embedding.from_pretrained(torch.from_numpy(glove), freeze=True)
        else:
            self.embedding = nn.Embedding(vocab_size, embed_size, padding_idx=padding_idx)
        self.lstm = nn.LSTM(self.embed_size, self.hidden_size, dropout=drop_ratio * (num_layers > 1),
                            num_layers=num_layers, batch_first=True, bidirectional=bidirectional)
        self.enc2dec = nn.Linear(self.hidden_size * self.num_directions, self.hidden_size * self.num_directions)
        self.compute_dtype = compute_dtype
        self.dropout_seed = 0xE2C0DE
        # True: the embedding gradient is summed in a fixed order (vln_embed_bwd_det, ~35 us more per iteration at B=64,
        # L=80) and a whole training iteration becomes reproducible bit for bit; False: float atomics like torch's own
        # embedding backward on a GPU.
        self.deterministic_embedding_grad = False
        # True: the backward's d x product is issued inside the weight gradients' pack launch (ops.linear_fwd_post)
        self.dx_with_wgrads = True
        # True: the context's layout changes ride in the launches of the encoder -> decoder bridge's products (vln_layout_post)
        self.layout_with_bridge = True
        # (A/B, off: measured SLOWER, 1.68 vs 1.36 ms) the layer's own weight gradients accumulated INSIDE the persistent BPTT launch
        # by four extra waves per workgroup (vln_lstm_seq_bwd_w; bf16 mode, Hd 256, 256 inputs): same products as the pack +
        # contraction launches (within 5e-7), but the 32 KB of h / x rows every workgroup must pull per step block the recurrence
        # waves' hand-off loads in the compute unit's in-order vector-memory pipeline (csrc/encoder_persist.h, notes section 8)
        self.wgrad_inlaunch = False
        # The input projection formed INSIDE the persistent forward recurrence launch (vln_lstm_seq_fwd_x, round 6) where the library takes
        # it (vln_lstm_inproj_ok: Hd 256, 256 input features): four extra waves per recurrence workgroup form step s + 1's projection
        # while the first four run step s.  The projection launch (32 us) and the 84 MB it writes / the recurrence re-reads disappear:
        # headline 1.403 -> 1.370 ms; bit-identical to the per-step chain that reads the GEMM's output.  False: the GEMM launch (A/B).
        self.inproj = True
        self._calls = 0
        self._shadow = ShadowSet()
        self._param_names = [n for n, _ in self.named_parameters()]

    def _params_cached(self):
        """The module's Parameters in named_parameters() order, resolved once (the tree walk costs ~20 us per call);
        `.to()` / `load_state_dict` keep the Parameter objects, assigning a submodule or parameter drops the cache."""
        c = self.__dict__.get("_pcache")
        if c is None:
            c = [p for _, p in self.named_parameters()]
            object.__setattr__(self, "_pcache", c)
            object.__setattr__(self, "_pmap", dict(self.named_parameters()))
        return c

    def __setattr__(self, name, value):
        if isinstance(value, (nn.Module, nn.Parameter)):
            self.__dict__.pop("_pcache", None)
            self.__dict__.pop("_pmap", None)
        super().__setattr__(name, value)

    def _sync_ws(self, dev, B, Hd, dirs):
        """(pointer, bytes) of the device scratch of the persistent recurrence: group counters + status word, then the
        backward's partial-dh exchange buffer (vln_lstm_sync_ws_bytes)."""
        lib = _lib.load()
        need = int(lib.vln_lstm_sync_ws_bytes(B, Hd, dirs))
        w = getattr(self, "_sync_buf", None)
        if w is None or w.device != dev or w.numel() * 4 < need:
            w = torch.zeros((need + 3) // 4, dtype=torch.int32, device=dev)
            self._sync_buf = w
            self._sync_mode = None
        clock = self.__dict__.get("clock")
        mode = (None if clock is None else id(clock), B, Hd, dirs)
        if self.__dict__.get("_sync_mode") != mode:
            if self.__dict__.get("_sync_mode") is None:
                # a buffer this module has not used before: whatever the library remembers about its ADDRESS belongs to memory the
                # allocator has handed out again (the header gets its fill in front of the next counter-protocol launch)
                _lib.check(lib.vln_lstm_sync_ws_forget(w.data_ptr()), "vln_lstm_sync_ws_forget")
            # The exchange's tags count launches: by the library on the host, or from the clock's device word.  When the
            # counting changes hands (or the layout changes) old tags mean nothing: clear the exchange once.
            go, gb = _lib.i64(), _lib.i64()
            _lib.check(lib.vln_lstm_sync_granule_range(B, Hd, dirs, C.byref(go), C.byref(gb)), "vln_lstm_sync_granule_range")
            if self.__dict__.get("_sync_mode") is not None or clock is not None:
                w[go.value // 4:(go.value + gb.value) // 4].zero_()
            if clock is not None:
                clock.register_sequence(w, int(lib.vln_lstm_sync_seq_offset(B, Hd, dirs)), go.value, gb.value)
            self._sync_mode = mode
        return w.data_ptr(), w.numel() * 4

    def persistent_status(self) -> int:
        """0 = every in-kernel wait of the last persistent launch completed; 1 = a bounded spin timed out."""
        w = getattr(self, "_sync_buf", None)
        return 0 if w is None else int(w[32].item())

    def ran_persistent(self) -> bool:
        """True when persistent recurrence launches have run on this module's scratch buffer: their dependency groups tally how they
        handed off (vln_lstm_fwd_handoff_stats / vln_lstm_handoff_stats; a per-step launch chain leaves the tallies at zero)."""
        w = getattr(self, "_sync_buf", None)
        if w is None:
            return False
        lib = _lib.load()
        n = 0
        for f in (lib.vln_lstm_fwd_handoff_stats, lib.vln_lstm_handoff_stats):
            a, b = C.c_uint32(), C.c_uint32()
            _lib.check(f(w.data_ptr(), C.byref(a), C.byref(b)), "vln_lstm_handoff_stats")
            n += a.value + b.value
        return n > 0

    def _refresh_shadows(self):
        """Per layer: w_ih [dirs*4Hd, I] (+ transpose), bsum = b_ih + b_hh [dirs*4Hd], w_hh [dirs][4Hd,Hd] (+ per-direction
        transposes), and enc2dec -- all written by ONE launch (ops.ShadowBatch), straight into their stacked layouts."""
        dt = self.compute_dtype
        t = self._shadow.t
        dev = self.enc2dec.weight.device
        Hd, dirs = self.hidden_size, self.num_directions
        # the job list only depends on addresses: after an optimizer step (same storage, new values) it is replayed as is
        ck = (dt, tuple(p.data_ptr() for p in self._params_cached()))
        c = self.__dict__.get("_sb_handle")
        if c is not None and c[0] == ck:
            ops.ShadowBatch.replay(c[1])
            return
        sb = ops.ShadowBatch()

        def buf(name, shape, dtype=dt):
            x = t.get(name)
            if x is None or x.dtype != dtype or x.device != dev or x.shape != shape:
                x = t[name] = torch.empty(shape, dtype=dtype, device=dev)
            return x

        for k in range(self.num_layers):
            sfxs = [f"_l{k}"] + ([f"_l{k}_reverse"] if dirs == 2 else [])
            I = getattr(self.lstm, "weight_ih" + sfxs[0]).shape[1]
            w_ih, w_ih_t = buf(f"w_ih{k}", (dirs * 4 * Hd, I)), buf(f"w_ih_t{k}", (I, dirs * 4 * Hd))
            w_hh, w_hh_t = buf(f"w_hh{k}", (dirs, 4 * Hd, Hd)), buf(f"w_hh_t{k}", (dirs, Hd, 4 * Hd))
            bsum = buf(f"bsum{k}", (dirs * 4 * Hd,), torch.float32)
            for d, s in enumerate(sfxs):
                r0, r1 = d * 4 * Hd, (d + 1) * 4 * Hd
                sb.add(getattr(self.lstm, "weight_ih" + s).detach(), w_ih[r0:r1], w_ih_t[:, r0:r1])
                sb.add(getattr(self.lstm, "weight_hh" + s).detach(), w_hh[d], w_hh_t[d])
                sb.add(getattr(self.lstm, "bias_ih" + s).detach().view(1, -1), bsum[r0:r1].view(1, -1), None,
                       src2=getattr(self.lstm, "bias_hh" + s).detach().view(1, -1))
        w = self.enc2dec.weight.detach()
        sb.add(w, buf("w_e2d", tuple(w.shape)), buf("w_e2d_t", (w.shape[1], w.shape[0])))
        object.__setattr__(self, "_sb_handle", (ck, sb.run()))

    def prefresh(self):
        """Refresh the weight shadows NOW if a parameter changed since they were built (see GatedModuleMixin.prefresh)."""
        params = self._params_cached()
        key = ShadowSet.key_of(params, self.compute_dtype)
        if self._shadow.stale(key):
            with torch.no_grad():
                self._refresh_shadows()
            self._shadow.commit(key)

    def forward(self, inputs: torch.Tensor, lengths, already_sorted: bool = True, ride=None):
        """inputs [B, max_len] int64 on the GPU, lengths [B] (CPU or GPU, any int type).  Rows are processed
        independently with packed-sequence semantics, so `already_sorted` needs no special handling.
        ride (optional, staging.DeviceFeatureStore.rollout_ride): a rollout's feature gather that this call carries -- as
        passenger workgroups of the last layer's persistent recurrence launch (half of the CUs are idle there at B = 64)."""
        if not inputs.is_cuda:
            raise _lib.VlnError("EncoderLSTM: inputs must be on the GPU; there is no CPU fallback")
        params = self._params_cached()
        key = ShadowSet.key_of(params, self.compute_dtype)
        if self._shadow.stale(key):
            with torch.no_grad():
                self._refresh_shadows()
            self._shadow.commit(key)
        clock = self.__dict__.get("clock")
        if clock is not None:          # runtime.DeviceClock: offsets / launch sequence relative to device words (graph-capturable)
            r = clock.rel(id(self))
            self._calls = clock.value(r)            # what a host counter would hold: the offset the tests export masks with
            off = _Off(r, clock.ptr, 2 * (r - 1))
        else:
            self._calls += 1
            off = _Off(self._calls)
        tokens = inputs.contiguous()
        if tokens.dtype != torch.int64:
            tokens = tokens.long()
        lens32 = torch.as_tensor(lengths).to(device=inputs.device, dtype=torch.int32)
        p = self.drop_ratio if self.training else 0.0
        object.__setattr__(self, "_ride", ride)
        ctx, dec_init, c_t = _EncoderFn.apply(self, tokens, lens32, p, off, *params)
        lp, self._last_ctx_lp = self.__dict__.get("_last_ctx_lp"), None
        if lp is not None:
            ctx._vln_lp = lp          # picked up by the decoders instead of casting the context again (runtime._ctx_lp)
        ops.stamp(ctx, dec_init, c_t)
        return ctx, dec_init, c_t
