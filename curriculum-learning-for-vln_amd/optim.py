"""Fused clip-grad-norm + optimizer step over flat buffers.

Reference semantics (engine/trainer.py): `optim_switcher` = {adam, rms, sgd} with torch defaults (:17-21); EnvDrop
trains with one RMSprop over encoder+decoder+critic after `clip_grad_norm(module.parameters(), 40.)` per module
(:380-381,423-427); Follower with two Adam instances (:65-67), Self-Monitor with one (:219-222).  Here parameters,
gradients and the optimizer state of all groups live in flat fp32 buffers (each `p.data` / `p.grad` is a view), so
`zero_grad` is one memset, the data-parallel exchange is one RCCL all-reduce of the gradient buffer (slices of it
may start early, see `start_allreduce`), and clip + update are two launches (`vln_*_clip_step`) instead of ~10
multi-tensor launches.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence

import torch

from . import _lib
from .dp import BucketReducer


class _FusedFlat:
    """Shared plumbing: flat parameter / gradient buffers, clip groups, overlapped all-reduce."""
    n_state = 0

    def __init__(self, groups: Sequence[Sequence[torch.nn.Parameter]], lr: float, clip_norm=0.0):
        """groups: one parameter list per clip group (e.g. [encoder.parameters(), decoder.parameters()]).
        clip_norm: one max-norm for every group, or one per group with 0 = that group is not clipped -- the reference clips
        encoder and decoder at 40 and leaves the critic alone although all three share one RMSprop (trainer.py:380-381,
        425-427): `FusedRMSprop([enc, dec, critic], clip_norm=[40, 40, 0])`.
        The flat-buffer plumbing (views, zero_grad, the overlapped all-reduce) is device-agnostic -- the world-size-2 gloo
        test drives it on CPU -- but `step()` is the HIP kernel: it raises on CPU parameters, there is no CPU update."""
        self.groups: List[List[torch.nn.Parameter]] = [[p for p in g if p.requires_grad] for g in groups]
        if not 1 <= len(self.groups) <= 8:
            raise ValueError("1..8 clip groups")
        self.lr = lr
        self.clip_norm = clip_norm
        first = self.groups[0][0]
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
        self.state = [torch.zeros(total, dtype=torch.float32, device=dev) for _ in range(self.n_state)]
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
        nb = sum((begins[i + 1] - begins[i] + 4095) // 4096 for i in range(len(self.groups)))   # = vln_rmsprop_partial_floats
        self._partial = torch.empty(max(int(nb), 1), dtype=torch.float32, device=dev)
        self.norms = torch.zeros(len(self.groups), dtype=torch.float32, device=dev)
        self._reducer = BucketReducer(self.flat_g)
        self.steps = 0
        self._cleared_by_step = False
        self._started, self._exchanged = [], False      # this iteration's early all-reduce slices / its exchange is complete

    def zero_grad(self, set_to_none: bool = False):
        if self._cleared_by_step:          # step(zero_grads=True) cleared the buffer while it read it
            self._cleared_by_step = False
        else:
            self.flat_g.zero_()
        for p, v in zip(self.params, self.views):
            if p.grad is not v:
                p.grad = v

    def start_allreduce(self, group_index: int, group=None):
        """Begin the RCCL all-reduce of clip group `group_index`'s gradients now, asynchronously (call when they are
        final -- e.g. from `EnvDropDecoder.grads_ready_hook`, which fires before the encoder's BPTT starts);
        `allreduce()` later reduces the remaining groups and waits."""
        self._reducer.start(self._begins[group_index], self._begins[group_index + 1], group)
        self._started.append(group_index)

    def allreduce(self, group=None):
        self._reducer.finish(group)
        self._exchanged = True

    def abandon_iteration(self, early_groups=(), group=None):
        """Error path of a data-parallel loop: this rank's iteration raised (e.g. VlnError from vln_persistent_check) somewhere
        between its backward and its update while the OTHER ranks run the iteration to the end.  Issue exactly the collectives a
        normal iteration issues -- the early slices `early_groups` that were not started yet, then the rest -- so that the
        ranks' collective sequences stay matched; the reduced numbers are garbage and the caller discards the iteration on
        every rank (it all-reduces a flag at its next synchronisation point).  A no-op when the exchange had completed."""
        if not self._exchanged:
            for gi in early_groups:
                if gi not in self._started:
                    self.start_allreduce(gi, group)
            self._reducer.finish(group)
        self._started, self._exchanged = [], False

    def _launch(self, lib, grad_scale: float) -> int:
        raise NotImplementedError

    @property
    def clip_norm(self):
        return self._clip

    @clip_norm.setter
    def clip_norm(self, v):
        vals = [float(x) for x in v] if isinstance(v, (list, tuple)) else [float(v)] * len(self.groups)
        if len(vals) != len(self.groups):
            raise ValueError(f"clip_norm: {len(vals)} values for {len(self.groups)} groups")
        self._clip = vals
        self._clip_c = (C.c_float * len(vals))(*vals)

    @torch.no_grad()
    def step(self, grad_scale: float = 1.0, zero_grads: bool = False):
        """zero_grads=True: the update kernel -- the last reader of the gradient buffer -- also clears it, and the NEXT
        `zero_grad()` is free (a 41 MB memset per iteration for the EnvDrop agent).  Only for loops that do not touch the
        gradients between `step()` and `zero_grad()`."""
        if not self.flat_p.is_cuda:
            raise _lib.VlnError(f"{type(self).__name__}.step: parameters must be on the GPU; there is no CPU update")
        self.steps += 1
        lib = _lib.load()
        st = lib.vln_persistent_check()         # a timed-out persistent recurrence in this iteration: do not train on its numbers
        if st:
            _lib.check(st, "vln_persistent_check")
        if grad_scale <= 0:
            raise ValueError("grad_scale must be positive")
        st = self._launch(lib, -grad_scale if zero_grads else grad_scale)
        self._cleared_by_step = bool(zero_grads)
        self._started, self._exchanged = [], False
        if st:
            _lib.check(st, type(self).__name__ + ".step")
        for p in self.params:                       # in-place update outside autograd: tell version-keyed caches
            torch.autograd.graph.increment_version(p)


class FusedRMSprop(_FusedFlat):
    """torch.optim.RMSprop(params, lr) with torch defaults (alpha 0.99, eps 1e-8, no momentum, not centered)."""
    n_state = 1

    def __init__(self, groups, lr: float = 1e-2, alpha: float = 0.99, eps: float = 1e-8, clip_norm=0.0):
        super().__init__(groups, lr, clip_norm)
        self.alpha, self.eps = alpha, eps
        self.sq = self.state[0]

    def _launch(self, lib, grad_scale):
        return lib.vln_rmsprop_clip_step(self.flat_p.data_ptr(), self.flat_g.data_ptr(), self.sq.data_ptr(), self._begins,
                                         len(self.groups), self._partial.data_ptr(), self.norms.data_ptr(), self.lr, self.alpha,
                                         self.eps, self._clip_c, grad_scale, _lib.raw_stream())


class FusedAdam(_FusedFlat):
    """torch.optim.Adam(params, lr) with torch defaults (betas (0.9, 0.999), eps 1e-8, no weight decay, no amsgrad)."""
    n_state = 2

    def __init__(self, groups, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, clip_norm=0.0):
        super().__init__(groups, lr, clip_norm)
        self.betas, self.eps = betas, eps
        self.exp_avg, self.exp_avg_sq = self.state

    def use_clock(self, clock):
        """The step count (Adam's bias corrections) in a DEVICE word that `clock.tick()` bumps once per iteration, so that
        `step()` can be captured into a whole-iteration graph (graphs.IterationGraph).  Call before the next iteration's tick."""
        self._step_word = torch.full((1,), self.steps, dtype=torch.int64, device=self.flat_p.device)
        clock.register_counter(self._step_word, 1)
        return self

    def _launch(self, lib, grad_scale):
        word = self.__dict__.get("_step_word")
        return lib.vln_adam_clip_step(self.flat_p.data_ptr(), self.flat_g.data_ptr(), self.exp_avg.data_ptr(),
                                      self.exp_avg_sq.data_ptr(), self._begins, len(self.groups), self._partial.data_ptr(),
                                      self.norms.data_ptr(), self.lr, self.betas[0], self.betas[1], self.eps,
                                      0 if word is not None else self.steps, None if word is None else word.data_ptr(),
                                      self._clip_c, grad_scale, _lib.raw_stream())


class FusedSGD(_FusedFlat):
    """torch.optim.SGD(params, lr) with torch defaults (no momentum, no weight decay)."""
    n_state = 0

    def __init__(self, groups, lr: float = 1e-3, clip_norm=0.0):
        super().__init__(groups, lr, clip_norm)

    def _launch(self, lib, grad_scale):
        return lib.vln_sgd_clip_step(self.flat_p.data_ptr(), self.flat_g.data_ptr(), self._begins, len(self.groups),
                                     self._partial.data_ptr(), self.norms.data_ptr(), self.lr, self._clip_c, grad_scale,
                                     _lib.raw_stream())


# the reference's optim_switcher (engine/trainer.py:17-21)
optim_switcher = {"adam": FusedAdam, "rms": FusedRMSprop, "sgd": FusedSGD}
