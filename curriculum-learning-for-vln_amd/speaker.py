"""Speaker modules for back-translation: drop-ins for the reference's `SpeakerEncoder` /
`SpeakerDecoder` (src/model/units.py:286-395, themselves after github.com/airsplay/R2R-EnvDrop r2r_src/model.py).

Same constructors, forward contracts and `state_dict` keys.  The math reuses the path's HIP kernels through the C ABI:
the whole-sequence LSTMs run as one input-projection GEMM + the (persistent) recurrence of `vln_lstm_seq_fwd/bwd`
(`_LSTMSeqFn`, unpacked: the reference calls `nn.LSTM` on the padded batch without lengths), the attention over the
36 views / over the encoded path is `SoftDotAttention` on the attention kernels, the vocabulary projection is
`vln_linear_fwd`.  An initial state (`h0`/`c0`, e.g. carried through word-by-word inference) is seeded into the
recurrence's first time slot; only a DIFFERENTIABLE initial state takes the LSTM-cell path.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from . import _lib, ops
from . import functional as Fh
from .decoders import SoftDotAttention, _Seeded, _need_gpu

_p = ops._p


class _LSTMSeqFn(torch.autograd.Function):
    """One (bi)LSTM layer over a full-length batch, zero initial state.  x_tm [L*B, I] time-major;
    params = (w_ih, w_hh, b_ih, b_hh) per direction.  Returns y_tm [L*B, dirs*Hd], h_T, c_T [B, dirs*Hd]."""

    @staticmethod
    def forward(ctx, owner, x_tm, B, L, h0, c0, seq, *params):
        """seq: -1 = the library counts this buffer's launches on the host; >= 0 = the launch's index into the device-resident launch
        sequence of a runtime.DeviceClock (forward `seq`, backward `seq + 1`)."""
        lib = _lib.load()
        dirs = len(params) // 4
        Hd = params[1].shape[1]
        dev = x_tm.device
        f32 = dict(dtype=torch.float32, device=dev)
        dt = owner.compute_dtype                     # bf16: the weights are STREAMED as bf16 shadows, everything else stays fp32
        f32_in = dt != torch.float32 and owner.fp32_input_weights      # this layer's weights streamed in fp32 all the same
        wtype = ops.BF16 if (dt == torch.bfloat16 and not f32_in) else ops.F32     # (f32_in: W_ih in VLN_F32S arithmetic, W_hh in the recurrence's fp32 form)
        sh = owner._shadows(params)                  # stacked copies in the streamed dtype (+ transposes), one launch per optimizer step
        x_tm = x_tm.contiguous()
        xproj = ops.linear_fwd(x_tm, sh["w_ih"], sh["bsum"], split=f32_in)
        hprev = ops.empty(dirs, L, B, Hd, **f32)
        cprev = ops.empty(dirs, L, B, Hd, **f32)
        y = ops.empty(L * B, dirs * Hd, **f32)
        act = ops.empty(L * B, dirs * 4 * Hd, **f32)
        tanh_c = ops.empty(L * B, dirs * Hd, **f32)
        hcat = ops.empty(B, dirs * Hd, **f32)
        ccat = ops.empty(B, dirs * Hd, **f32)
        lens32 = owner._full_lengths(B, L, dev)
        _lib.check(lib.vln_lstm_seq_fwd(_p(xproj), _p(sh["w_hh"]), wtype, _p(lens32), _p(hprev), _p(cprev), _p(y), _p(act),
                                        _p(tanh_c), _p(hcat), _p(ccat), B, L, Hd, dirs, _p(h0), _p(c0),
                                        *owner._sync_ws(dev, B, Hd, dirs), seq, None, _lib.raw_stream()), "vln_lstm_seq_fwd")
        ctx.owner, ctx.dims, ctx.dt, ctx.f32_in, ctx.seq, ctx.wtype, ctx.sh = owner, (B, L, Hd, dirs), dt, f32_in, seq, wtype, sh
        ctx.params = params
        ctx.save_for_backward(x_tm, hprev, cprev, act, tanh_c, lens32)
        ctx.set_materialize_grads(False)
        return y, hcat, ccat

    @staticmethod
    def backward(ctx, dy, dh, dc):
        lib = _lib.load()
        x_tm, hprev, cprev, act, tanh_c, lens32 = ctx.saved_tensors
        B, L, Hd, dirs = ctx.dims
        sh = ctx.sh                                  # the forward's shadows: the weights have not changed in between
        dev = x_tm.device
        f32 = dict(dtype=torch.float32, device=dev)
        z = lambda t: t.reshape(B, dirs, Hd).transpose(0, 1).contiguous() if t is not None else ops.zeros(dirs, B, Hd, **f32)
        dh_pass, dc_carry = z(dh), z(dc)
        dgates = ops.empty(L * B, dirs * 4 * Hd, **f32)
        dyc = dy.contiguous() if dy is not None else None
        _lib.check(lib.vln_lstm_seq_bwd(_p(dyc), _p(sh["w_hh_t"]), ctx.wtype, _p(lens32), _p(act), _p(tanh_c), _p(cprev), _p(dgates),
                                        _p(dh_pass), _p(dc_carry), None, None, B, L, Hd, dirs, *ctx.owner._sync_ws(dev, B, Hd, dirs),
                                        ctx.seq + 1 if ctx.seq >= 0 else -1, None, _lib.raw_stream()), "vln_lstm_seq_bwd")
        # the layer's weight gradients (all over the same L * B rows) as ONE grouped contraction, its bias gradients as one grouped
        # column sum (they were a product + a slab reduction per matrix and two launches per bias pair); split hi + lo planes in bf16
        # mode, as before (never the plain-bf16 form: the 2176-wide input rows, see fp32_input_weights)
        wb = ops.WgradBatch(ctx.dt != torch.float32, never_plain=True)
        cb = ops.ColsumBatch()
        grads = []
        I = x_tm.shape[1]
        for d in range(dirs):
            dg = dgates[:, d * 4 * Hd:(d + 1) * 4 * Hd]
            # (functional.set_grad_in_place: the launches add into the parameters' .grad, autograd gets None)
            (g_ih, a_ih), (g_hh, a_hh), (b1, a1), (b2, a2) = (Fh._gsink(p) for p in ctx.params[4 * d:4 * d + 4])
            wb.add(dg, x_tm, g_ih, a_ih)
            wb.add(dg, hprev[d].view(L * B, Hd), g_hh, a_hh)
            if a1 == a2:
                cb.add(dg, b1, b2, a1)
            else:
                cb.add(dg, b1, None, a1); cb.add(dg, b2, None, a2)
            grads += [Fh._gret(g_ih, a_ih), Fh._gret(g_hh, a_hh), Fh._gret(b1, a1), Fh._gret(b2, a2)]
        wb.run()
        cb.run()
        dx = None
        if ctx.needs_input_grad[1]:
            dx = ops.linear_fwd(dgates, sh["w_ih_t"], split=ctx.f32_in)
        return (None, dx, None, None, None, None, None) + tuple(grads)


class _SeqLSTM(nn.Module):
    """Parameter holder with nn.LSTM's names (`weight_ih_l0[_reverse]`, ...) + the HIP forward."""

    def __init__(self, input_size, hidden_size, bidirectional=False):
        super().__init__()
        # nn.LSTM registers the reference's parameter names and default init; its forward is never called
        self.rnn = nn.LSTM(input_size, hidden_size, 1, batch_first=True, bidirectional=bidirectional)
        self.hidden_size, self.dirs = hidden_size, 2 if bidirectional else 1
        self.compute_dtype = torch.float32
        # bf16 mode: stream this layer's W_ih / W_hh in fp32 all the same.  On for the speaker encoder's FIRST LSTM, whose 2176-wide
        # input rows sum K = 2176 products per gate: its weight gradients sat at 0.9-1.1e-2 of the fp32 reference with bf16 weights
        # (it runs over the <= 7 viewpoints of a path: the fp32 recurrence costs nothing measurable)
        self.fp32_input_weights = False

    def _sync_ws(self, dev, B, Hd, dirs):
        import ctypes as C
        lib = _lib.load()
        need = int(lib.vln_lstm_sync_ws_bytes(B, Hd, dirs))
        w = self.__dict__.get("_sync_buf")
        if w is None or w.device != dev or w.numel() * 4 < need:
            w = torch.zeros((need + 3) // 4, dtype=torch.int32, device=dev)
            object.__setattr__(self, "_sync_buf", w)
            object.__setattr__(self, "_sync_mode", None)
            _lib.check(lib.vln_lstm_sync_ws_forget(w.data_ptr()), "vln_lstm_sync_ws_forget")    # (a reused address: see encoder.py)
        clock = self.__dict__.get("clock")
        mode = (None if clock is None else id(clock), B, Hd, dirs)
        if self.__dict__.get("_sync_mode") != mode:      # the launch counting changes hands / the layout changes: as in EncoderLSTM._sync_ws
            go, gb = _lib.i64(), _lib.i64()
            _lib.check(lib.vln_lstm_sync_granule_range(B, Hd, dirs, C.byref(go), C.byref(gb)), "vln_lstm_sync_granule_range")
            if self.__dict__.get("_sync_mode") is not None or clock is not None:
                w[go.value // 4:(go.value + gb.value) // 4].zero_()
            if clock is not None:
                clock.register_sequence(w, int(lib.vln_lstm_sync_seq_offset(B, Hd, dirs)), go.value, gb.value)
            object.__setattr__(self, "_sync_mode", mode)
        return w.data_ptr(), w.numel() * 4

    def _shadows(self, params):
        """w_ih [dirs*4Hd, I] (+ transpose), w_hh [dirs][4Hd, Hd] (+ per-direction transposes) in the streamed dtype and bsum = b_ih + b_hh,
        rebuilt by ONE launch (ops.ShadowBatch) when a parameter has changed since -- every call cast, concatenated and transposed
        them anew (about twenty launches per layer and iteration)."""
        from .runtime import ShadowSet
        dt = torch.float32 if (self.compute_dtype == torch.float32 or self.fp32_input_weights) else self.compute_dtype
        ss = self.__dict__.get("_shadow")
        if ss is None:
            ss = ShadowSet()
            object.__setattr__(self, "_shadow", ss)
        key = ShadowSet.key_of(params, dt)
        if not ss.stale(key):
            return ss.t
        t, dirs = ss.t, len(params) // 4
        H4, I = params[0].shape
        Hd = params[1].shape[1]
        dev = params[0].device
        ck = (dt, tuple(p.data_ptr() for p in params))
        c = self.__dict__.get("_sb_handle")
        with torch.no_grad():
            if c is not None and c[0] == ck:
                ops.ShadowBatch.replay(c[1])
            else:
                def buf(name, shape, dtype=dt):
                    x = t.get(name)
                    if x is None or x.dtype != dtype or x.device != dev or tuple(x.shape) != tuple(shape):
                        x = t[name] = torch.empty(shape, dtype=dtype, device=dev)
                    return x
                w_ih, w_ih_t = buf("w_ih", (dirs * H4, I)), buf("w_ih_t", (I, dirs * H4))
                w_hh, w_hh_t = buf("w_hh", (dirs, H4, Hd)), buf("w_hh_t", (dirs, Hd, H4))
                bsum = buf("bsum", (dirs * H4,), torch.float32)
                sb = ops.ShadowBatch()
                for d in range(dirs):
                    r0, r1 = d * H4, (d + 1) * H4
                    sb.add(params[4 * d].detach(), w_ih[r0:r1], w_ih_t[:, r0:r1])
                    sb.add(params[4 * d + 1].detach(), w_hh[d], w_hh_t[d])
                    sb.add(params[4 * d + 2].detach().view(1, -1), bsum[r0:r1].view(1, -1), None, src2=params[4 * d + 3].detach().view(1, -1))
                object.__setattr__(self, "_sb_handle", (ck, sb.run()))
        ss.commit(key)
        return t

    def _full_lengths(self, B, L, dev):
        c = self.__dict__.get("_lens")
        if c is None or c[0] != (B, L, dev):
            c = ((B, L, dev), torch.full((B,), L, dtype=torch.int32, device=dev))
            object.__setattr__(self, "_lens", c)
        return c[1]

    def _seq(self):
        """This call's index into the clock's launch sequence (forward; the backward takes the next one), -1 without a clock."""
        clock = self.__dict__.get("clock")
        return -1 if clock is None else 2 * (clock.rel(("seq", id(self)), 31) - 1)

    def _params(self):
        out = []
        for sfx in ["_l0"] + (["_l0_reverse"] if self.dirs == 2 else []):
            out += [getattr(self.rnn, n + sfx) for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
        return out

    def forward(self, x, state=None, time_major=None):
        """x [B, L, I] -> (y [B, L, dirs*H], (h_T, c_T) each [dirs, B, H]) like nn.LSTM(batch_first=True).
        time_major=(B, L): `x` is already the [L*B, I] time-major matrix (row t*B + b) the recurrence reads (vln_embed_fwd's layout)."""
        if time_major is not None:
            B, L = time_major
            x_tm_in = x
            x = None
        else:
            B, L, _ = x.shape
            x_tm_in = None
        H = self.hidden_size
        h0 = c0 = None
        if state is not None:
            h0, c0 = state
            if h0.requires_grad or c0.requires_grad:   # gradient into the initial state: the LSTM-cell path, one direction
                if self.dirs != 1:
                    raise _lib.VlnError("speaker LSTM: a differentiable initial state needs a unidirectional layer")
                w_ih, w_hh, b_ih, b_hh = self._params()
                h, c = h0[0], c0[0]
                ys = []
                if x is None:
                    x = x_tm_in.view(L, B, -1).transpose(0, 1)
                for t in range(L):
                    h, c = Fh.LSTMCellFn.apply(x[:, t].contiguous(), h, c, w_ih, w_hh, b_ih, b_hh, self.compute_dtype)
                    ys.append(h)
                return torch.stack(ys, 1), (h.unsqueeze(0), c.unsqueeze(0))
            h0 = h0.detach().to(torch.float32).contiguous()
            c0 = c0.detach().to(torch.float32).contiguous()
        x_tm = x_tm_in if x_tm_in is not None else x.transpose(0, 1).reshape(L * B, x.shape[-1])
        y, hcat, ccat = _LSTMSeqFn.apply(self, x_tm, B, L, h0, c0, self._seq(), *self._params())
        y = y.view(L, B, self.dirs * H).transpose(0, 1)
        return y, (hcat.view(B, self.dirs, H).transpose(0, 1), ccat.view(B, self.dirs, H).transpose(0, 1))


class _EmbedFn(torch.autograd.Function):
    """nn.Embedding(padding_idx) + Dropout (units.py:364-365) on the library's kernels: vln_embed_fwd gathers the rows straight
    into the TIME-major matrix the sequence LSTM reads (row t*B + b) with the dropout of the site fused (element index in that
    layout), vln_embed_bwd scatters the gradient rows back (float atomics, like torch's embedding backward on a GPU; the pad row
    gets none, units.py:352)."""

    @staticmethod
    def forward(ctx, words, weight, pad, seed, offset, p, base=None, det=False):
        lib = _lib.load()
        ctx.det = det
        words = words.contiguous()
        B, L = words.shape
        E = weight.shape[1]
        out = ops.empty(L * B, E, dtype=torch.float32, device=weight.device)
        _lib.check(lib.vln_embed_fwd(_p(words), _p(weight.detach()), _p(out), B, L, E, seed, offset, p, base, _lib.raw_stream()),
                   "vln_embed_fwd")
        ctx.save_for_backward(words, weight)
        ctx.cfg = (B, L, E, pad, seed, offset, p, base)
        return out

    @staticmethod
    def backward(ctx, dx):
        words, weight = ctx.saved_tensors
        B, L, E, pad, seed, offset, p, base = ctx.cfg
        dE, acc = Fh._gsink(weight)               # (both kernels ADD into dE)
        if not acc:
            dE.zero_()
        lens32 = torch.full((B,), L, dtype=torch.int32, device=weight.device)
        if ctx.det and E <= 1024:           # fixed summation order, no float atomics (SpeakerDecoder.deterministic_embedding_grad)
            _lib.check(_lib.load().vln_embed_bwd_det(_p(words), _p(lens32), _p(dx.contiguous()), _p(dE), B, L, E, weight.shape[0],
                                                     -1 if pad is None else pad, seed, offset, p, base, _lib.raw_stream()), "vln_embed_bwd_det")
        else:
            _lib.check(_lib.load().vln_embed_bwd(_p(words), _p(lens32), _p(dx.contiguous()), _p(dE), B, L, E, -1 if pad is None else pad,
                                                 seed, offset, p, base, _lib.raw_stream()), "vln_embed_bwd")
        return None, Fh._gret(dE, acc), None, None, None, None, None, None


def _rename_lstm_keys(module: nn.Module, names):
    """state_dict compatibility: the holder keeps nn.LSTM's parameters under `<name>.rnn.*`; the reference has them
    directly under `<name>.*` (units.py:303,309,348)."""
    def save_hook(mod, sd, prefix, local):
        for n in names:
            for k in [k for k in sd if k.startswith(prefix + n + ".rnn.")]:
                sd[prefix + n + "." + k[len(prefix + n + ".rnn."):]] = sd.pop(k)

    def load_hook(sd, prefix, *a):
        for n in names:
            for k in [k for k in sd if k.startswith(prefix + n + ".") and not k.startswith(prefix + n + ".rnn.")]:
                sd[prefix + n + ".rnn." + k[len(prefix + n + "."):]] = sd.pop(k)

    module._register_state_dict_hook(save_hook)
    module._register_load_state_dict_pre_hook(load_hook)


class SpeakerEncoder(nn.Module, _Seeded):
    """units.py:286-341: LSTM over the path's action features -> attention over each step's 36 views -> post-LSTM."""

    def __init__(self, feature_size, hidden_size, dropout_ratio, bidirectional, angle_feat_size, feat_dropout,
                 compute_dtype=torch.float32):
        super().__init__()
        self.num_directions = 2 if bidirectional else 1
        self.hidden_size = hidden_size
        self.num_layers = 1
        self.feature_size = feature_size
        self.angle_feat_size = angle_feat_size
        self.drop_ratio, self.feat_drop_ratio = float(dropout_ratio), float(feat_dropout)
        self.lstm = _SeqLSTM(feature_size, hidden_size // self.num_directions, bidirectional)
        self.drop = nn.Dropout(p=dropout_ratio)
        self.drop3 = nn.Dropout(p=feat_dropout)
        self.attention_layer = SoftDotAttention(query_dim=hidden_size, context_dim=feature_size)
        self.post_lstm = _SeqLSTM(hidden_size, hidden_size // self.num_directions, bidirectional)
        self._init_seed(0x59EA)
        _rename_lstm_keys(self, ("lstm", "post_lstm"))
        self.lstm.fp32_input_weights = True
        self.set_compute_dtype(compute_dtype)

    def set_compute_dtype(self, dt):
        """torch.bfloat16: the weight matrices (both LSTMs, the attention's two projections) are streamed as bf16 shadows; features,
        activations, recurrent state, softmax and gradients stay fp32."""
        self.compute_dtype = dt
        self.lstm.compute_dtype = self.post_lstm.compute_dtype = self.attention_layer.compute_dtype = dt

    def forward(self, action_embeds, feature, lengths=None, already_dropfeat=False):
        """action_embeds [B, Lp, F], feature [B, Lp, 36, F] (both mutated in place by the feature dropout like the
        reference, units.py:322,331) -> context [B, Lp, hidden]."""
        _need_gpu(action_embeds, "SpeakerEncoder")
        off, base = self._next(), self._drop_base()
        p, pf = (self.drop_ratio, self.feat_drop_ratio) if self.training else (0.0, 0.0)
        F, ANG = self.feature_size, self.angle_feat_size
        B, Lp, _ = action_embeds.shape
        x = action_embeds
        if not already_dropfeat and pf > 0:
            xc = x if x.is_contiguous() else x.contiguous()
            ops.feat_dropout_inplace(xc, F - ANG, ANG, self.dropout_seed, off + 0, pf, base=base)
            if xc is not x:
                x.copy_(xc)
            fc = feature if feature.is_contiguous() else feature.contiguous()
            ops.feat_dropout_inplace(fc, F - ANG, ANG, self.dropout_seed, off + 1, pf, base=base)
            if fc is not feature:
                feature.copy_(fc)
        ctx, _ = self.lstm(x)
        ctx = Fh.dropout(ctx.contiguous(), p, self.training, self.dropout_seed, off + 2, base)
        x, _ = self.attention_layer(ctx.view(B * Lp, self.hidden_size), feature.reshape(B * Lp, -1, F))
        x = Fh.dropout(x.view(B, Lp, -1), p, self.training, self.dropout_seed, off + 3, base)
        x, _ = self.post_lstm(x)
        return Fh.dropout(x.contiguous(), p, self.training, self.dropout_seed, off + 4, base)


class SpeakerDecoder(nn.Module, _Seeded):
    """units.py:344-395: word embedding -> LSTM -> attention over the encoded path -> vocabulary logits."""

    def __init__(self, vocab_size, embedding_size, padding_idx, hidden_size, dropout_ratio, compute_dtype=torch.float32):
        super().__init__()
        self.hidden_size = hidden_size
        self.drop_ratio = float(dropout_ratio)
        self.embedding = nn.Embedding(vocab_size, embedding_size, padding_idx)
        self.lstm = _SeqLSTM(embedding_size, hidden_size, False)
        self.drop = nn.Dropout(dropout_ratio)
        self.attention_layer = SoftDotAttention(query_dim=hidden_size)
        self.projection = nn.Linear(hidden_size, vocab_size)
        self.baseline_projection = nn.Sequential(nn.Linear(hidden_size, 128), nn.ReLU(), nn.Dropout(dropout_ratio),
                                                 nn.Linear(128, 1))
        self._init_seed(0x5DEC)
        self.deterministic_embedding_grad = False      # True: the embedding gradient in a fixed summation order (no float atomics)
        _rename_lstm_keys(self, ("lstm",))
        self.set_compute_dtype(compute_dtype)

    def set_compute_dtype(self, dt):
        """torch.bfloat16: LSTM, attention and vocabulary-projection weights streamed as bf16 shadows (the embedding rows are
        gathered in fp32)."""
        self.compute_dtype = dt
        self.lstm.compute_dtype = self.attention_layer.compute_dtype = dt

    def forward(self, words, ctx, ctx_mask, h0, c0):
        """words [Bw, Lw] int64, ctx [B, Lp, H], ctx_mask [B, Lp] (True = masked), h0/c0 [1, Bw, H]
        -> (logit [Bw, Lw, vocab], h1, c1).  Bw may be a multiple of B (beam search), units.py:375-376."""
        _need_gpu(ctx, "SpeakerDecoder")
        off, base = self._next(), self._drop_base()
        p = self.drop_ratio if self.training else 0.0
        H = self.hidden_size
        Bw, Lw = words.shape
        # embedding rows + dropout in one launch, written in the time-major layout the recurrence reads (vln_embed_fwd)
        emb_tm = _EmbedFn.apply(words, self.embedding.weight, self.embedding.padding_idx, self.dropout_seed, off + 0, p, base,
                                self.deterministic_embedding_grad)
        x, (h1, c1) = self.lstm(emb_tm, (h0, c0), time_major=(Bw, Lw))
        x = Fh.dropout(x.contiguous(), p, self.training, self.dropout_seed, off + 1, base)
        n = Bw * Lw
        mult = n // ctx.size(0)
        ctx_e = ctx.unsqueeze(1).expand(-1, mult, -1, -1).contiguous().view(n, -1, H)
        mask_e = ctx_mask.unsqueeze(1).expand(-1, mult, -1).contiguous().view(n, -1) if ctx_mask is not None else None
        x, _ = self.attention_layer(x.view(n, H), ctx_e, mask=mask_e)
        x = Fh.dropout(x.view(Bw, Lw, H), p, self.training, self.dropout_seed, off + 2, base)
        logit = Fh.linear(x.view(n, H), self.projection.weight, self.projection.bias, ops.ACT_NONE, self.compute_dtype).view(Bw, Lw, -1)
        return logit, h1, c1


# ---------------------------------------------------------------------------------------------------------------
# The loop around the two modules: teacher-forced training loss, word-by-word inference and the back-translation
# hook of the EnvDrop rollout (agent/speaker.py:235-376, agent/envdrop.py:105-121,155-157).  The reference's Speaker
# object is tied to the simulator (`from_shortest_path`, speaker.py:191-226, walks the environment) and does not run as
# shipped (`np.bool`, `self.decoder.drop_env`); what is reproduced here is its arithmetic on tensors:
# the caller hands over the path features `(can_feats [B,Lp,F], img_feats [B,Lp,36,F], lengths [B])` that
# `from_shortest_path` would have produced.
# ---------------------------------------------------------------------------------------------------------------
PAD, UNK, EOS, BOS = 0, 1, 2, 3            # utils/misc.py:21-25 (base vocabulary of the Tokenizer)


def length2mask(lengths, device, size: Optional[int] = None) -> torch.Tensor:
    """utils/misc.py:481-486: True where the position is past the row's length."""
    lengths = torch.as_tensor(lengths, dtype=torch.int64)
    size = int(lengths.max()) if size is None else size
    return (torch.arange(size, dtype=torch.int64)[None, :] >= lengths[:, None]).to(device)


class Speaker:
    """speaker.py:16-376 without the simulator: `teacher_forcing` (the speaker's training / scoring loss) and
    `infer_batch` (greedy or sampled instruction generation) on the HIP modules."""

    def __init__(self, encoder: "SpeakerEncoder", decoder: "SpeakerDecoder", max_decode: int = 120,
                 pad: int = PAD, unk: int = UNK, eos: int = EOS, bos: int = BOS, seed: int = 0x5BEA):
        self.encoder, self.decoder = encoder, decoder
        self.max_decode = max_decode                      # config.py:118 MAX_DECODE
        self.pad, self.unk, self.eos, self.bos = pad, unk, eos, bos
        self.rnn_dim = decoder.hidden_size
        self.angle_feat_size = encoder.angle_feat_size
        self.seed, self._draws = seed, 0

    def _mode(self, train: bool):
        self.encoder.train(train)
        self.decoder.train(train)

    def _zero_state(self, B, dev):
        return (torch.zeros(1, B, self.rnn_dim, device=dev), torch.zeros(1, B, self.rnn_dim, device=dev))

    def teacher_forcing(self, can_feats, img_feats, lengths, insts, train: bool = True, for_listener: bool = False, ctx_mask=None):
        """speaker.py:235-290.  insts [B, Lw] int64 (<BOS> w1 .. <EOS> <PAD>..).  train -> mean CE over the non-pad
        targets; for_listener -> the un-reduced [B, Lw-1] losses (beam-search scoring); eval -> (loss, word_accu,
        sent_accu).  ctx_mask: `length2mask(lengths)` already on the device (a captured iteration cannot copy it there)."""
        from . import losses
        self._mode(train)
        dev = can_feats.device
        B = can_feats.shape[0]
        ctx = self.encoder(can_feats, img_feats, lengths)
        h_t, c_t = self._zero_state(B, dev)
        if ctx_mask is None:
            ctx_mask = length2mask(lengths, dev, ctx.shape[1])
        logits, _, _ = self.decoder(insts, ctx, ctx_mask, h_t, c_t)                   # [B, Lw, vocab]
        Lw, V = logits.shape[1], logits.shape[2]
        # "-1 for aligning" / "1: ignores <BOS>" (speaker.py:270-271) WITHOUT the copy of logits[:, :-1]: every position is scored, the
        # last one against <PAD> -- ignored by the loss, zero gradient, exactly what the slice's backward would have filled in
        flat = logits.reshape(B * Lw, V)
        tgt = torch.cat((insts[:, 1:], insts.new_full((B, 1), self.pad)), 1).reshape(-1)
        if train and not for_listener:             # mean over the non-pad targets inside the launch (speaker.py:272-273)
            return losses.masked_cross_entropy(flat, tgt, None, "mean", ignore_index=self.pad)
        per_word = losses.masked_cross_entropy(flat, tgt, None, "none", ignore_index=self.pad).view(B, Lw)[:, :-1]
        if for_listener:
            return per_word
        n_words = (insts[:, 1:] != self.pad).sum()
        loss = per_word.sum() / n_words.to(per_word.dtype)
        predict = logits.detach().argmax(dim=2)                                       # [B, Lw]
        gt_mask = insts != self.pad
        correct = (predict[:, :-1] == insts[:, 1:]) & gt_mask[:, 1:]
        n_gt = gt_mask[:, 1:].sum(dim=1)
        word_accu = correct.sum().item() / max(1, int(n_gt.sum().item()))
        sent_accu = (correct.sum(dim=1) == n_gt).sum().item() / B
        return loss.item(), word_accu, sent_accu

    def infer_batch(self, can_feats, img_feats, lengths, sampling: bool = False, train: bool = False, featdropmask=None):
        """speaker.py:292-376.  Greedy (or sampled) decoding from <BOS> until every row has produced <EOS> or
        `max_decode` words; words after a row's <EOS> are <PAD>.  `featdropmask` [F-angle]: the environment-dropout
        mask shared with the follower (applied in place to the image part of both feature tensors, speaker.py:313-315).
        Returns words [B, n] (numpy int64); with sampling and train also (log_probs, hiddens, entropies) as tensors
        that carry gradients."""
        import numpy as np
        from . import losses
        self._mode(train)
        dev = can_feats.device
        B = can_feats.shape[0]
        if featdropmask is not None:
            a = self.angle_feat_size
            img_feats[..., :-a] *= featdropmask
            can_feats[..., :-a] *= featdropmask
        ctx = self.encoder(can_feats, img_feats, lengths, already_dropfeat=featdropmask is not None)
        ctx_mask = length2mask(lengths, dev, ctx.shape[1])
        h_t, c_t = self._zero_state(B, dev)
        ended = np.zeros(B, dtype=bool)
        word = torch.full((B, 1), self.bos, dtype=torch.int64, device=dev)
        unk_mask = torch.zeros(B, self.decoder.projection.out_features, dtype=torch.bool, device=dev)
        unk_mask[:, self.unk] = True                                                  # no <UNK> in inference (speaker.py:336)
        words, log_probs, hiddens, entropies = [], [], [], []
        grad = torch.enable_grad() if (train and sampling) else torch.no_grad()
        with grad:
            for _ in range(self.max_decode):
                logits, h_t, c_t = self.decoder(word, ctx, ctx_mask, h_t, c_t)        # [B, 1, vocab]
                logits = logits.view(B, -1)
                if sampling:
                    self._draws += 1
                    w, lp, ent = losses.sample_action(logits, unk_mask, seed=self.seed, offset=self._draws)
                    log_probs.append(lp if train else lp.detach())
                    hiddens.append(h_t.squeeze(0) if train else h_t.squeeze(0).detach())
                    entropies.append(ent if train else ent.detach())
                else:
                    w = logits.masked_fill(unk_mask, float("-inf")).argmax(dim=1)
                cpu_word = w.cpu().numpy().copy()          # a COPY: the word fed back below keeps the model's choice
                cpu_word[ended] = self.pad
                words.append(cpu_word)
                word = w.view(B, 1)
                ended = np.logical_or(ended, cpu_word == self.eos)
                if ended.all():
                    break
        out = np.stack(words, 1)
        if train and sampling:
            return out, torch.stack(log_probs, 1), torch.stack(hiddens, 1), torch.stack(entropies, 1)
        return out


def env_drop_mask(decoder, n: Optional[int] = None, device=None) -> torch.Tensor:
    """The per-batch environment-dropout mask of back-translation (envdrop.py:106: `self.decoder.drop_env(torch.ones(
    img_feat_size))`, i.e. the follower's feature-dropout module applied to a vector of ones): [n] values in
    {0, 1/(1-p)}, ONE mask shared by every view / candidate / step of the batch and by the speaker.  Drawn from the
    decoder's Philox stream."""
    n = (decoder.feature_size - decoder.angle_feat_size) if n is None else n
    device = next(decoder.parameters()).device if device is None else device
    p = decoder.feat_drop_ratio if decoder.training else 0.0
    if p <= 0.0:
        return torch.ones(n, device=device)
    draws = getattr(decoder, "_env_mask_draws", 0) + 1      # its own counter: the step offsets of the rollouts are untouched
    decoder._env_mask_draws = draws
    return ops.dropout_mask(n, decoder.dropout_seed ^ 0xE17D, draws, p, device)


def back_translate(speaker: Speaker, decoder, can_feats, img_feats, lengths):
    """The self-training branch at the top of `EnvDropAgent.rollout` (envdrop.py:105-121): draw the shared environment
    mask, let the speaker describe the (masked) shortest path greedily, prepend <BOS> and close unfinished sentences
    with <EOS>.  Returns (instr_encoding [B, 1+n] numpy int64, noise [F-angle]); the caller rebuilds the batch with
    these instructions and multiplies the image part of every step's features by `noise`, passing
    `already_dropfeat=True` to the decoder (envdrop.py:155-160)."""
    import numpy as np
    noise = env_drop_mask(decoder, device=can_feats.device)
    insts = speaker.infer_batch(can_feats, img_feats, lengths, featdropmask=noise)
    boss = np.full((insts.shape[0], 1), speaker.bos, np.int64)
    insts = np.concatenate((boss, insts), 1)
    for inst in insts:
        if inst[-1] != speaker.pad:            # the instruction has not ended (envdrop.py:113-114)
            inst[-1] = speaker.eos
    return insts, noise
