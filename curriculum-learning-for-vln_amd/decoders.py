"""Drop-in attention units and the Speaker-Follower / Self-Monitoring decoders on the HIP operators.

Reference surface reproduced (constructor args, forward signatures, return tuples, state_dict keys):
  units.py:77-122  SoftDotAttention          units.py:125-160 VisualSoftDotAttention
  units.py:163-185 ActionScoring             units.py:188-207 PositionalEncoding
  units.py:210-242 MLPwithBN                 policy.py:15-60  AttnDecoderLSTM
  policy.py:67-166 MonitorDecoder
Every Linear / LSTMCell / attention contraction runs in the gfx950 kernels (functional.py); concatenations,
BatchNorm (+ReLU) of the BN-MLP runs on vln_bn_fwd/bwd; a few [B,H]-sized elementwise ops are torch glue.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import _lib, ops
from . import functional as Fh
from .monitor_step import FollowerStepFn, MonitorStepFn, gated_ctx


def _need_gpu(t, who):
    if not t.is_cuda:
        raise _lib.VlnError(f"{who}: tensors must be on the GPU; there is no CPU fallback")


class _Seeded:
    """Per-module Philox stream for the dropout sites (offset advances every forward)."""

    def _init_seed(self, seed):
        self.dropout_seed = seed
        self._calls = 0
        self.compute_dtype = torch.float32

    def _next(self):
        clock = self.__dict__.get("clock")
        if clock is not None:          # runtime.DeviceClock: an offset relative to the clock's device word (see _drop_base)
            r = clock.rel(id(self), 31)
            self._calls = clock.value(r)
            return r * 16
        self._calls += 1
        return self._calls * 16

    def _drop_base(self):
        """Device word the kernels add (x 8) to this module's dropout offsets, or None (absolute host offsets)."""
        clock = self.__dict__.get("clock")
        return None if clock is None else clock.ptr


class SoftDotAttention(nn.Module):
    def __init__(self, query_dim, context_only=False, context_dim=None):
        super().__init__()
        self.context_only = context_only
        ctx_dim = query_dim if context_dim is None else context_dim
        self.linear_in = nn.Linear(query_dim, ctx_dim, bias=False)
        if not context_only:
            self.linear_out = nn.Linear(query_dim + ctx_dim, query_dim, bias=False)
        self.compute_dtype = torch.float32

    def forward(self, h, context, mask=None):
        _need_gpu(h, "SoftDotAttention")
        target = Fh.linear(h, self.linear_in.weight, None, ops.ACT_NONE, self.compute_dtype)
        wc, attn = Fh.soft_dot_core(target, context, context, mask)
        if self.context_only:
            return wc, attn
        h_tilde = Fh.linear(torch.cat((wc, h), 1), self.linear_out.weight, None, ops.ACT_TANH, self.compute_dtype)
        return h_tilde, attn


class VisualSoftDotAttention(nn.Module):
    def __init__(self, h_dim, v_dim=None, dot_dim=256):
        super().__init__()
        self.linear_in_h = nn.Linear(h_dim, dot_dim, bias=True)
        self.use_v_linear = v_dim is not None
        if self.use_v_linear:
            self.linear_in_v = nn.Linear(v_dim, dot_dim, bias=True)
        self.compute_dtype = torch.float32

    def forward(self, h, visual_context, mask=None):
        _need_gpu(h, "VisualSoftDotAttention")
        target = Fh.linear(h, self.linear_in_h.weight, self.linear_in_h.bias, ops.ACT_NONE, self.compute_dtype)
        if self.use_v_linear:
            keys = Fh.linear(visual_context, self.linear_in_v.weight, self.linear_in_v.bias, ops.ACT_NONE, self.compute_dtype)
        else:
            keys = visual_context
        assert keys.shape[-1] == target.shape[-1]
        return Fh.soft_dot_core(target, keys, visual_context, mask)     # weighted sum over the UN-projected context


class ActionScoring(nn.Module):
    def __init__(self, action_size, hidden_size, dot_size=256):
        super().__init__()
        self.linear_act = nn.Linear(action_size, dot_size, bias=True)
        self.linear_hid = nn.Linear(hidden_size, dot_size, bias=True)
        self.linear_out = nn.Linear(dot_size, 1, bias=True)
        self.compute_dtype = torch.float32

    def forward(self, act_cands, h_tilde):
        dt = self.compute_dtype
        target = Fh.linear(h_tilde, self.linear_hid.weight, self.linear_hid.bias, ops.ACT_NONE, dt).unsqueeze(1)
        context = Fh.linear(act_cands, self.linear_act.weight, self.linear_act.bias, ops.ACT_NONE, dt)
        product = context * target
        return Fh.linear(product, self.linear_out.weight, self.linear_out.bias, ops.ACT_NONE, torch.float32).squeeze(2)


class AttnDecoderLSTM(nn.Module, _Seeded):
    """policy.py:15-60."""

    def __init__(self, hidden_size, drop_ratio, action_embed_size=2048 + 128, feature_size=2048 + 128,
                 image_attn_layers=None, compute_dtype=torch.float32):
        super().__init__()
        self.action_embed_size, self.feature_size, self.hidden_size = action_embed_size, feature_size, hidden_size
        self.drop_ratio = float(drop_ratio)
        self.drop = nn.Dropout(p=drop_ratio)
        self.lstm = nn.LSTMCell(action_embed_size + feature_size, hidden_size)
        self.text_attn = SoftDotAttention(hidden_size)
        self.visual_attn = VisualSoftDotAttention(hidden_size, feature_size)
        self.decode_action = ActionScoring(action_embed_size, hidden_size)
        self._init_seed(0xF0110)
        self.fused_step = True            # False: every operator its own autograd node (A/B, and the reference for the fused node)
        self.c_step = True                # the fused node as ONE C call each way (csrc/follower.hip); False: launches driven from Python
        self.split_attention = True       # the C-call step's two attentions on four workgroups per episode (B * 4 <= the device's CUs)
        self._attn_sync = None
        self.set_compute_dtype(compute_dtype)

    def set_compute_dtype(self, dt):
        self.compute_dtype = dt
        for m in (self.text_attn, self.visual_attn, self.decode_action):
            m.compute_dtype = dt

    def _attn_sync_buf(self, dev, B):
        w = self._attn_sync
        if w is None or w[0].device != dev or w[1] < B:
            n = int(_lib.load().vln_attn_sync_bytes(B))
            w = self._attn_sync = (torch.zeros((n + 3) // 4, dtype=torch.int32, device=dev), B)
        return w[0]

    def project_candidates(self, cands_all):
        """ActionScoring's candidate projection `linear_act(a_t_cands)` (units.py:175) for SEVERAL steps in one product: cands_all
        [T, B, C, A] contiguous (the steps' candidate tensors as slices of one buffer) -> [T] matrices [B * C, D] to hand to
        `forward(..., cand_context=...)`.  The projection depends on the batch only, so a teacher-forced rollout (the candidates of
        every step are known when it starts) forms it once, off the steps' dependent chain; its weight gradient is formed by the
        steps' backward as before (it never needed the projection's own backward).  No autograd through the returned matrices."""
        T, B, C_, A = cands_all.shape
        ds = self.decode_action
        with torch.no_grad():
            w = Fh.SHADOWS.get(ds.linear_act.weight, "n", self.compute_dtype)
            y = ops.linear_fwd(cands_all.view(T * B * C_, A), w, ds.linear_act.bias.detach())
        return list(y.view(T, B * C_, -1).unbind(0))

    def forward(self, img_feature, a_t_prev, a_t_cands, h_0, c_0, ctx, ctx_mask=None, cand_context=None):
        _need_gpu(img_feature, "AttnDecoderLSTM")
        site = self._next()
        p, tr, seed = self.drop_ratio, self.training, self.dropout_seed
        if self.fused_step and ctx.dtype == torch.float32 and img_feature.shape[2] % 4 == 0 and self.hidden_size % 4 == 0 \
                and a_t_cands.shape[2] % 4 == 0 and not img_feature.requires_grad and not a_t_cands.requires_grad:
            va, ds = self.visual_attn, self.decode_action            # the whole step as ONE autograd node
            core = FollowerStepFn if self.c_step else Fh.FollowerCoreFn      # one C call per direction / Python-driven launches
            if self._drop_base() is not None and not self.c_step:
                raise _lib.VlnError("AttnDecoderLSTM: a DeviceClock needs the C-call step (c_step=True)")
            if self.c_step:          # the rollout's context gradient accumulates in ONE buffer (monitor_step.gated_ctx)
                ctx = gated_ctx(ctx)[0]
                # every weight shadow the step streams, refreshed (when an optimizer step staled it) by ONE launch
                dt_ = self.compute_dtype
                Fh.SHADOWS.ensure([(va.linear_in_h.weight, "n", dt_), (va.linear_in_h.weight, "t", dt_), (va.linear_in_v.weight, "n", dt_),
                                   (va.linear_in_v.weight, "t", dt_),
                                   (self.text_attn.linear_in.weight, "n", dt_), (self.text_attn.linear_in.weight, "t", dt_),
                                   (self.text_attn.linear_out.weight, "n", dt_), (self.text_attn.linear_out.weight, "t", dt_),
                                   (ds.linear_act.weight, "n", dt_), (ds.linear_hid.weight, "n", dt_), (ds.linear_hid.weight, "t", dt_)],
                                  [(self.lstm.weight_ih, self.lstm.weight_hh, dt_, False), (self.lstm.weight_ih, self.lstm.weight_hh, dt_, True)])
            cfg = (tr, self.compute_dtype, p, seed, site) + ((self._drop_base(),) if self._drop_base() is not None else ())
            if self.c_step and (self.split_attention or cand_context is not None):
                # the exchange buffer of the four-workgroups-per-episode attentions; the candidates' projection formed up front (round 6)
                cfg = cfg[:5] + (self._drop_base(), self._attn_sync_buf(img_feature.device, img_feature.shape[0]) if self.split_attention else None,
                                 cand_context)
            elif cand_context is not None:
                raise _lib.VlnError("AttnDecoderLSTM: cand_context needs the C-call step (c_step=True)")
            logit, h_new, c_new, word_w, view_w = core.apply(
                cfg, ctx_mask, img_feature, a_t_prev, a_t_cands, h_0, c_0, ctx,
                va.linear_in_h.weight, va.linear_in_h.bias, va.linear_in_v.weight, va.linear_in_v.bias,
                self.lstm.weight_ih, self.lstm.weight_hh, self.lstm.bias_ih, self.lstm.bias_hh,
                self.text_attn.linear_in.weight, self.text_attn.linear_out.weight,
                ds.linear_act.weight, ds.linear_act.bias, ds.linear_hid.weight, ds.linear_hid.bias, ds.linear_out.weight, ds.linear_out.bias)
            return logit, (h_new, c_new), (word_w, view_w)
        # operator-by-operator path (shapes the fused node does not take; the reference for it in the tests)
        # (1) look at the panorama with the previous hidden state: [B,36,F] -> [B,F]
        pano, view_w = self.visual_attn(h_0, img_feature)
        # (2) recurrent update on [previous action | attended view]
        cell_in = Fh.dropout(torch.cat((a_t_prev, pano), 1), p, tr, seed, site)
        h_new, c_new = Fh.LSTMCellFn.apply(cell_in, h_0, c_0, self.lstm.weight_ih, self.lstm.weight_hh,
                                           self.lstm.bias_ih, self.lstm.bias_hh, self.compute_dtype)
        # (3) ground in the instruction, (4) score every candidate against the grounded state
        grounded, word_w = self.text_attn(Fh.dropout(h_new, p, tr, seed, site + 1), ctx, ctx_mask)
        return self.decode_action(a_t_cands, grounded), (h_new, c_new), (word_w, view_w)


class PositionalEncoding(nn.Module, _Seeded):
    def __init__(self, d_model, dropout, max_len=80):
        super().__init__()
        self.p = float(dropout)
        self.dropout = nn.Dropout(p=dropout)
        pe = torch.zeros(max_len, d_model)
        position = torch.arange(0, max_len).float().unsqueeze(1)
        div_term = torch.exp(torch.arange(0, d_model, 2).float() * -(math.log(10000.0) / d_model))
        pe[:, 0::2] = torch.sin(position * div_term)
        pe[:, 1::2] = torch.cos(position * div_term)
        self.register_buffer("pe", pe.unsqueeze(0))
        self._init_seed(0x9E)

    def forward(self, x):
        x = x + self.pe[:, :x.size(1)]
        return Fh.dropout(x, self.p, self.training, self.dropout_seed, self._next())


class _HipLinear(nn.Linear):
    compute_dtype = torch.float32

    def forward(self, x):
        return Fh.linear(x, self.weight, self.bias, ops.ACT_NONE, Fh.wdtype(self.compute_dtype, "mlp"))   # (operator path: exact fp32 for an fp32-streamed layer)


class _PhiloxDropout(nn.Dropout, _Seeded):
    def __init__(self, p):
        super().__init__(p)
        self._init_seed(0xD0)

    def forward(self, x):
        return Fh.dropout(x, self.p, self.training, self.dropout_seed, self._next())


class _HipBatchNorm1d(nn.BatchNorm1d):
    """nn.BatchNorm1d (same parameters / buffers / state_dict keys) on the fused HIP kernel; `relu=True` also applies the
    ReLU that follows it in the BN-MLP."""

    def forward(self, x, relu=False):
        _need_gpu(x, "BatchNorm1d")
        if x.dim() != 2 or x.dtype != torch.float32 or (x.shape[1] & 3) or self.momentum is None:
            # no torch fallback in the product path: the agents' BN-MLP only ever sees [rows, 4k] fp32 matrices
            raise _lib.VlnError(f"BatchNorm1d: vln_bn_fwd takes fp32 [rows, features % 4 == 0] with a fixed momentum; got "
                                f"{tuple(x.shape)} {x.dtype}, momentum={self.momentum}")
        return Fh.batch_norm(x, self.weight, self.bias, self.running_mean if self.track_running_stats else None,
                             self.running_var if self.track_running_stats else None,
                             self.num_batches_tracked if self.track_running_stats else None, self.training, self.momentum,
                             self.eps, relu)


class MLPwithBN(nn.Module):
    """units.py:210-242 (same Sequential layout -> same state_dict keys `mlp.{i}.*`)."""

    def __init__(self, input_size, hidden_size, out_size=None, dropout=.0, use_bn=False, use_bias=True, relu=True):
        super().__init__()
        self.in_size = input_size
        layers = []
        if use_bn:
            layers.append(_HipBatchNorm1d(input_size))
        dims = [input_size] + list(hidden_size)
        for i in range(len(dims) - 1):
            layers.append(_HipLinear(dims[i], dims[i + 1], bias=use_bias))
            if use_bn:
                layers.append(_HipBatchNorm1d(dims[i + 1]))
            if dropout > 0:
                layers.append(_PhiloxDropout(dropout))
            if relu:
                layers.append(nn.ReLU(inplace=True))
        self.out_size = hidden_size[-1]
        if out_size:
            layers.append(_HipLinear(dims[-1], out_size, bias=use_bias))
            self.out_size = out_size
        self.mlp = nn.Sequential(*layers)
        self.inputs_in_place = True       # forward_pair: the two batches read from their own arrays (False: concatenated first; A/B)

    def _fused_plan(self):
        """(bn0, [(linear, bn, dropout or None)]) when the Sequential is exactly BatchNorm, (Linear, BatchNorm, [Dropout],
        ReLU)* -- the layout the Self-Monitor agent builds -- else None."""
        plan = self.__dict__.get("_plan", False)
        if plan is not False:
            return plan
        layers = list(self.mlp)
        plan = None
        if layers and isinstance(layers[0], _HipBatchNorm1d):
            i, seq, ok = 1, [], True
            while i < len(layers) and ok:
                if not (isinstance(layers[i], _HipLinear) and i + 1 < len(layers) and isinstance(layers[i + 1], _HipBatchNorm1d)):
                    ok = False
                    break
                lin, bnl = layers[i], layers[i + 1]
                i += 2
                dr = None
                if i < len(layers) and isinstance(layers[i], _PhiloxDropout):
                    dr = layers[i]; i += 1
                if i < len(layers) and isinstance(layers[i], nn.ReLU):
                    i += 1
                else:
                    ok = False
                seq.append((lin, bnl, dr))
            if ok and seq and all(m.track_running_stats and m.momentum is not None and m.affine for m in [layers[0]] + [t[1] for t in seq]):
                plan = (layers[0], seq)
        object.__setattr__(self, "_plan", plan)
        return plan

    def forward(self, x, row_zero=None):
        """`row_zero` [rows] (bool): output rows forced to 0 (the padded candidate slots, policy.py:148-149)."""
        plan = self._fused_plan()
        if plan is not None and x.dim() == 2 and x.dtype == torch.float32 and x.is_cuda and x.shape[1] % 4 == 0 and \
                all(t[0].out_features % 4 == 0 for t in plan[1]):
            bn0, seq = plan
            training = self.training
            drops, bufs, tensors = [], [(bn0.running_mean, bn0.running_var, bn0.num_batches_tracked)], [bn0.weight, bn0.bias]
            for lin, bnl, dr in seq:
                p = dr.p if (dr is not None and training) else 0.0
                drops.append((float(p), dr.dropout_seed if dr is not None else 0, dr._next() if (dr is not None and p > 0) else 0) +
                             ((dr._drop_base(),) if (dr is not None and dr._drop_base() is not None) else ()))
                bufs.append((bnl.running_mean, bnl.running_var, bnl.num_batches_tracked))
                tensors += [lin.weight, lin.bias, bnl.weight, bnl.bias]
            return Fh.bn_mlp(x, row_zero, training, bn0.eps, bn0.momentum, seq[0][0].compute_dtype, drops, bufs, tensors)
        y = self._forward_layers(x)
        if row_zero is not None:
            y = y * (~row_zero).to(y.dtype).unsqueeze(1)
        return y

    def forward_pair(self, x1, x2, row_zero2=None):
        """(forward(x1), forward(x2, row_zero=row_zero2)) -- the reference's two calls with the same MLP on the previous action and
        on the candidates (policy.py:140-149) -- as ONE fused call: every BatchNorm normalises each batch with its own statistics
        and updates the running statistics twice, in that order; the Linear layers, their gradients and their weight gradients
        run once over all rows.  Equal to the two calls up to summation order (the dropout offsets are consumed in the same
        order, masks indexed per batch).  Falls back to the two calls when the fused path does not apply."""
        plan = self._fused_plan()
        ok = plan is not None and x1.dim() == 2 and x2.dim() == 2 and x1.dtype == torch.float32 and x2.dtype == torch.float32 and \
            x1.is_cuda and x1.shape[1] % 4 == 0 and x1.shape[1] == x2.shape[1] and all(t[0].out_features % 4 == 0 for t in plan[1]) and \
            len(plan[1]) <= Fh._lib.BN_MLP_MAX_LAYERS and all(t[0].bias is not None for t in plan[1]) and Fh._BN_MLP_C_CALL[0]
        if not ok:
            return self.forward(x1), self.forward(x2, row_zero=row_zero2)
        bn0, seq = plan
        training = self.training
        bufs, tensors = [(bn0.running_mean, bn0.running_var, bn0.num_batches_tracked)], [bn0.weight, bn0.bias]
        for lin, bnl, dr in seq:
            bufs.append((bnl.running_mean, bnl.running_var, bnl.num_batches_tracked))
            tensors += [lin.weight, lin.bias, bnl.weight, bnl.bias]

        def offsets():                     # one call's worth of dropout offsets, layer by layer (what forward() consumes)
            out = []
            for lin, bnl, dr in seq:
                p = dr.p if (dr is not None and training) else 0.0
                out.append((float(p), dr.dropout_seed if dr is not None else 0, dr._next() if (dr is not None and p > 0) else 0) +
                           ((dr._drop_base(),) if (dr is not None and dr._drop_base() is not None) else ()))
            return out
        drops = offsets()
        offs2 = [d[2] for d in offsets()]
        if not (x1.requires_grad or x2.requires_grad) and x2.stride(1) == 1 and x1.stride(1) == 1 and x2.stride(0) % 4 == 0 and \
                x2.data_ptr() % 16 == 0 and self.inputs_in_place:
            # the two batches read where they are (vln_bn_mlp.x2): no [R, F] concatenated copy per step
            return Fh.bn_mlp(x1, row_zero2, training, bn0.eps, bn0.momentum, seq[0][0].compute_dtype, drops, bufs, tensors,
                             seg=(x1.shape[0], offs2, x2))
        x = torch.cat([x1, x2], 0)
        return Fh.bn_mlp(x, row_zero2, training, bn0.eps, bn0.momentum, seq[0][0].compute_dtype, drops, bufs, tensors,
                         seg=(x1.shape[0], offs2))

    def _forward_layers(self, x):
        layers = list(self.mlp)
        i = 0
        while i < len(layers):                    # BatchNorm followed by ReLU (no dropout between) is one launch
            m = layers[i]
            if isinstance(m, _HipBatchNorm1d) and i + 1 < len(layers) and isinstance(layers[i + 1], nn.ReLU):
                x = m(x, relu=True)
                i += 2
            else:
                x = m(x)
                i += 1
        return x


class MonitorDecoder(nn.Module, _Seeded):
    """policy.py:67-166 (co-grounding + progress monitor).  BASELINE config 2 does not ask for bf16: the default compute dtype is
    fp32.  `compute_dtype=torch.bfloat16` streams bf16 shadows of the weight matrices (activations stay fp32) EXCEPT the ones
    named in `fp32_weights` (of "mlp" = the BN-MLP's Linear layers, "w_tin", "w_vh", "w_cat" = [lstm.weight_ih | weight_hh],
    "w_a" = action_linear, "w_m" = monitor_linear)."""
    # Class-wide default of `fp32_weights`, decided by measurement (scripts/bf16_exceptions_ab.py, profiles/round4_notes.md): with
    # these four every output and gradient of BASELINE config 2 is within 1e-2 of the fp32 reference (the BN-MLP ends in a ReLU
    # whose units flip under ANY rounding of its 2176 -> 1024 weights: 0.16 of the gradient's range with bf16 weights; the two
    # attention queries sit in front of a softmax; the LSTM's K = 3072 gate sums feed the recurrent state).  frozenset() = every
    # matrix bf16 (round 3's mode: faster, outside north_star's tolerance).
    default_fp32_weights = frozenset({"mlp", "w_cat", "w_vh", "w_tin"})

    def __init__(self, rnn_hidden_size, drop_ratio, max_enc_len, mlp_dims=(128, 1024), action_embed_size=2048 + 128,
                 feature_size=2048 + 128, compute_dtype=torch.float32):
        super().__init__()
        self.rnn_hidden_size, self.max_enc_len, self.mlp_dims = rnn_hidden_size, max_enc_len, list(mlp_dims)
        self.feature_size, self.action_embed_size = feature_size, action_embed_size
        self.img_hidden_size = self.mlp_dims[-1]
        self.drop_ratio = float(drop_ratio)
        self.proj_navigable_mlp = MLPwithBN(input_size=action_embed_size, hidden_size=self.mlp_dims, use_bn=True,
                                            dropout=0.5, use_bias=True, relu=True)
        self.position = PositionalEncoding(rnn_hidden_size, dropout=0.1, max_len=max_enc_len)
        self.text_attn = SoftDotAttention(rnn_hidden_size, context_only=True)
        self.visual_attn = VisualSoftDotAttention(rnn_hidden_size, None, self.img_hidden_size)
        self.drop = nn.Dropout(p=drop_ratio)
        self.lstm = nn.LSTMCell(self.img_hidden_size * 2 + rnn_hidden_size, rnn_hidden_size)
        self.action_linear = nn.Linear(rnn_hidden_size * 2, self.img_hidden_size)
        self.monitor_linear = nn.Linear(rnn_hidden_size + self.img_hidden_size, rnn_hidden_size, bias=True)
        self.critic = nn.Sequential(nn.Linear(max_enc_len + rnn_hidden_size, 1), nn.Tanh())
        self._init_seed(0x5E1F)
        self.fused_step = True            # False: every operator its own autograd node (A/B, and the reference for the fused node)
        self.c_step = True                # the fused node as ONE C call each way (csrc/monitor.hip); False: launches driven from Python
        self.merge_projections = False    # True: the BN-MLP's two calls per step as one two-batch call (MLPwithBN.forward_pair)
        self.fp32_weights = frozenset(type(self).default_fp32_weights)
        self.set_compute_dtype(compute_dtype)

    def set_compute_dtype(self, dt, fp32_weights=None):
        self.compute_dtype = dt
        if fp32_weights is not None:
            self.fp32_weights = frozenset(fp32_weights)
        unknown = self.fp32_weights - {"mlp", "w_tin", "w_vh", "w_cat", "w_a", "w_m"}
        if unknown:
            raise ValueError(f"MonitorDecoder.fp32_weights: unknown matrix name(s) {sorted(unknown)}")
        for m in self.modules():
            if m is not self and hasattr(m, "compute_dtype"):
                m.compute_dtype = dt
        self.text_attn.compute_dtype = self._wd("w_tin")
        self.visual_attn.compute_dtype = self._wd("w_vh")
        for m in self.proj_navigable_mlp.modules():
            if isinstance(m, _HipLinear):      # fp32-streamed in bf16 mode: (bf16, {"mlp"}) -> fp32 arrays, split-bf16 arithmetic (VLN_F32S)
                m.compute_dtype = (dt, frozenset({"mlp"})) if (dt != torch.float32 and "mlp" in self.fp32_weights) else dt

    def _wd(self, name):
        return torch.float32 if name in self.fp32_weights else self.compute_dtype

    def _node_dtype(self):
        """what the fused step nodes get as `dtype`: the compute dtype, or (compute dtype, fp32 names) when an override is active"""
        if self.compute_dtype == torch.float32 or not (self.fp32_weights - {"mlp"}):
            return self.compute_dtype
        return (self.compute_dtype, self.fp32_weights)

    def _prefresh_shadows(self):
        """Every weight shadow a step streams (the BN-MLP's Linear layers, the step's five matrices + the [W_ih | W_hh] pair, each in
        its own streaming dtype), refreshed -- when an optimizer step staled it -- by ONE launch at the top of the rollout's first
        forward (Fh.SHADOWS.ensure) instead of a cast / transpose launch per matrix."""
        nd = self._node_dtype()
        wants = []
        for m in self.proj_navigable_mlp.mlp:
            if isinstance(m, _HipLinear):
                wd = Fh.wdtype(m.compute_dtype, "mlp")
                wants += [(m.weight, "n", wd), (m.weight, "t", wd)]
        for name, W in (("w_tin", self.text_attn.linear_in.weight), ("w_vh", self.visual_attn.linear_in_h.weight),
                        ("w_a", self.action_linear.weight), ("w_m", self.monitor_linear.weight)):
            wd = Fh.wdtype(nd, name)
            wants += [(W, "n", wd), (W, "t", wd)]
        wc = Fh.wdtype(nd, "w_cat")
        Fh.SHADOWS.ensure(wants, [(self.lstm.weight_ih, self.lstm.weight_hh, wc, False), (self.lstm.weight_ih, self.lstm.weight_hh, wc, True)])

    def policy_net(self, weighted_ctx, hidden, cands_rep):
        """logit[b,c] = cands_rep[b,c,:] . W_a [weighted_ctx ; hidden]      (policy.py:108-117)"""
        query = Fh.linear(torch.cat((weighted_ctx, hidden), 1), self.action_linear.weight, self.action_linear.bias,
                          ops.ACT_NONE, self._wd("w_a"))
        return Fh.AttnDotFn.apply(cands_rep, query)

    def progress_monitor(self, h_0, c_1, weighted_cands, ctx_attn, site=None):
        """tanh(W_c [ctx_attn ; drop(sigmoid(W_m [h_0 ; weighted_cands]) * tanh(c_1))])      (policy.py:119-130)

        This is the OPERATOR-BY-OPERATOR form of the head (`c_step = False`, the A/B / test reference of the one-call step): its gate
        product and the final Linear are library launches, `sigmoid(gate) * tanh(c_1)` and the two concatenations are torch
        elementwise ops on purpose -- autograd differentiates them, which is what `tests/test_hip_agents.py::test_monitor_fused_step_equals_operator_path` holds the
        fused `monitor_head_fwd/bwd` kernels of the default path (csrc/monitor.hip) against."""
        site = self._next() if site is None else site
        gate = Fh.linear(torch.cat((h_0, weighted_cands), 1), self.monitor_linear.weight, self.monitor_linear.bias,
                         ops.ACT_NONE, self._wd("w_m"))
        mem = Fh.dropout(torch.sigmoid(gate) * torch.tanh(c_1), self.drop_ratio, self.training, self.dropout_seed, site)
        head = self.critic[0]
        return Fh.linear(torch.cat((ctx_attn, mem), 1), head.weight, head.bias, ops.ACT_TANH, torch.float32).squeeze()

    def forward(self, img_feature, a_t_prev, a_t_cands, h_0, c_0, ctx, ctx_mask=None, candidate_mask=None):
        _need_gpu(a_t_prev, "MonitorDecoder")
        site = self._next()
        B, C, _ = a_t_cands.shape
        if self.fused_step and self.c_step:
            self._prefresh_shadows()
        # BN-MLP twice (previous action rows, then all B*C candidate rows incl. padded ones): two sets of batch
        # statistics and two running-stat updates per step, as in the reference
        if self.merge_projections:
            # both projections in ONE BN-MLP call (two batches, per-batch statistics): half the launches of this part of the step
            prev_rep, cand_rep = self.proj_navigable_mlp.forward_pair(a_t_prev, a_t_cands.reshape(B * C, self.action_embed_size),
                                                                      row_zero2=candidate_mask.reshape(B * C))
            cand_rep = cand_rep.view(B, C, -1)
        else:
            prev_rep = self.proj_navigable_mlp(a_t_prev)
            # the BN-MLP zeroes the padded candidate slots itself (row_zero): one autograd node per call
            cand_rep = self.proj_navigable_mlp(a_t_cands.reshape(B * C, self.action_embed_size),
                                               row_zero=candidate_mask.reshape(B * C)).view(B, C, -1)
        if self.fused_step and ctx.dtype == torch.float32 and cand_rep.shape[2] % 4 == 0 and self.rnn_hidden_size % 4 == 0 \
                and ctx_mask is not None and candidate_mask is not None:
            # everything after the BN-MLP as ONE autograd node (functional.MonitorCoreFn)
            pos = self.position
            cfg = (self.training, self._node_dtype(), pos.p, (pos.dropout_seed, pos._next()), self.drop_ratio, self.dropout_seed,
                   site, site + 1)
            if self._drop_base() is not None:
                if not self.c_step:
                    raise _lib.VlnError("MonitorDecoder: a DeviceClock needs the C-call step (c_step=True)")
                cfg = cfg + (self._drop_base(),)
            head = self.critic[0]
            core = MonitorStepFn if self.c_step else Fh.MonitorCoreFn       # one C call per direction / Python-driven launches
            if self.c_step:          # the rollout's context gradient accumulates in ONE buffer (monitor_step.gated_ctx)
                ctx = gated_ctx(ctx)[0]
            logit, progress, h_new, c_new, word_w, move_w = core.apply(
                cfg, pos.pe[0, :ctx.shape[1]], ctx_mask, candidate_mask, prev_rep, cand_rep, h_0, c_0, ctx,
                self.text_attn.linear_in.weight, self.visual_attn.linear_in_h.weight, self.visual_attn.linear_in_h.bias,
                self.lstm.weight_ih, self.lstm.weight_hh, self.lstm.bias_ih, self.lstm.bias_hh,
                self.action_linear.weight, self.action_linear.bias, self.monitor_linear.weight, self.monitor_linear.bias,
                head.weight, head.bias)
            return (logit, progress), (h_new, c_new), (word_w, move_w)
        # operator-by-operator path (shapes the fused node does not take)
        # co-grounding: words (position-encoded context) and candidates, both queried by h_0
        words, word_w = self.text_attn(h_0, self.position(ctx), ctx_mask)
        moves, move_w = self.visual_attn(h_0, cand_rep, candidate_mask)
        h_new, c_new = Fh.LSTMCellFn.apply(torch.cat((prev_rep, moves, words), 1), h_0, c_0, self.lstm.weight_ih,
                                           self.lstm.weight_hh, self.lstm.bias_ih, self.lstm.bias_hh, self._wd("w_cat"))
        logit = self.policy_net(words, Fh.dropout(h_new, self.drop_ratio, self.training, self.dropout_seed, site), cand_rep)
        progress = self.progress_monitor(h_0, c_new, moves, word_w, site + 1)
        return (logit, progress), (h_new, c_new), (word_w, move_w)
