"""Fused clip-grad-norm + RMSprop over flat buffers.

Reference semantics (engine/trainer.py:380-381,423-427): `torch.optim.RMSprop(params, lr)` with torch defaults and
`clip_grad_norm(module.parameters(), 40.)` per module before the step.  Here parameters, gradients and the
square-average state of all groups live in three flat fp32 buffers (each `p.data` / `p.grad` is a view), so
`zero_grad` is one memset, the data-parallel exchange is one RCCL all-reduce of the gradient buffer, and
clip + update are two launches (`vln_rmsprop_clip_step`) instead of ~10 multi-tensor launches.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence

import torch

from . import _lib
from .dp import BucketReducer


class FusedRMSprop:
    def __init__(self, groups: Sequence[Sequence[torch.nn.Parameter]], lr: float = 1e-2, alpha: float = 0.99,
                 eps: float = 1e-8, clip_norm: float = 0.0):
        """groups: one parameter list per clip group (e.g. [encoder.parameters(), decoder.parameters()])."""
        self.groups: List[List[torch.nn.Parameter]] = [[p for p in g if p.requires_grad] for g in groups]
        self.lr, self.alpha, self.eps, self.clip_norm = lr, alpha, eps, clip_norm
        first = self.groups[0][0]
        if not first.is_cuda:
            raise _lib.VlnError("FusedRMSprop: parameters must be on the GPU")
        begins, off = [], 0
        for g in self.groups:
            off = (off + 3) // 4 * 4
            begins.append(off)
            off += sum(p.numel() for p in g)
        begins.append((off + 3) // 4 * 4)
        total = begins[-1]
        dev = first.device
        self.flat_p = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
        self.sq = torch.zeros(total, dtype=torch.float32, device=dev)
        self._begins = (C.c_int64 * len(begins))(*begins)
        # group g covers [begins[g], begins[g+1]); the pad before the next group belongs to group g (zeros)
        self.params, self.views = [], []
        with torch.no_grad():
            for g, b in zip(self.groups, begins):
                o = b
                for p in g:
                    n = p.numel()
                    self.flat_p[o:o + n].copy_(p.detach().reshape(-1))
                    p.data = self.flat_p[o:o + n].view_as(p)
                    v = self.flat_g[o:o + n].view_as(p)
                    p.grad = v
                    self.params.append(p); self.views.append(v)
                    o += n
        lib = _lib.load()
        nb = lib.vln_rmsprop_partial_floats(self._begins, len(self.groups))
        self._partial = torch.empty(max(int(nb), 1), dtype=torch.float32, device=dev)
        self.norms = torch.zeros(len(self.groups), dtype=torch.float32, device=dev)
        self._reducer = BucketReducer(self.flat_g)

    def zero_grad(self, set_to_none: bool = False):
        self.flat_g.zero_()
        for p, v in zip(self.params, self.views):
            if p.grad is not v:
                p.grad = v

    def start_allreduce(self, group_index: int, group=None):
        """Begin the RCCL all-reduce of clip group `group_index`'s gradients now, asynchronously (call when they are
        final -- e.g. from `EnvDropDecoder.grads_ready_hook`, which fires before the encoder's BPTT starts);
        `allreduce()` later reduces the remaining groups and waits."""
        self._reducer.start(self._begins[group_index], self._begins[group_index + 1], group)

    def allreduce(self, group=None):
        self._reducer.finish(group)

    @torch.no_grad()
    def step(self, grad_scale: float = 1.0):
        lib = _lib.load()
        _lib.check(lib.vln_rmsprop_clip_step(self.flat_p.data_ptr(), self.flat_g.data_ptr(), self.sq.data_ptr(), self._begins,
                                             len(self.groups), self._partial.data_ptr(), self.norms.data_ptr(), self.lr,
                                             self.alpha, self.eps, self.clip_norm, grad_scale,
                                             _lib.raw_stream()), "vln_rmsprop_clip_step")
        for p in self.params:                       # in-place update outside autograd: tell version-keyed caches
            torch.autograd.graph.increment_version(p)
