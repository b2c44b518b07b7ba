"""Tensor-level wrappers over the C ABI: PyTorch supplies device memory and the
current HIP stream, nothing else.  All tensors must live on the GPU; fp32
activations, fp32 or bf16 streamed operands."""
from __future__ import annotations

from typing import Optional

import ctypes as C

import torch

from . import _lib

F32, BF16 = 0, 1
F32S = 2      # weight operands: fp32 in memory, split-bf16 arithmetic (include/vln_hip.h VLN_F32S)
F32X = 3      # ... three bf16 pieces per operand, six products: fp32-grade (VLN_F32X)
ACT_NONE, ACT_TANH, ACT_RELU = 0, 1, 2
ACT_ACCUM = 8          # flag: out += act(x @ w.T + bias)  (VLN_ACT_ACCUM)


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _stream():
    return _lib.raw_stream()


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise TypeError(f"unsupported dtype {t.dtype}")


def _req(t: torch.Tensor, name: str, dtype=torch.float32):
    if not t.is_cuda:
        raise _lib.VlnError(f"{name}: expected a GPU tensor (the HIP path has no CPU fallback)")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if t.shape[-1] > 1 and t.stride(-1) != 1:
        raise ValueError(f"{name}: innermost dim must be contiguous")
    return t


# ---- allocation ------------------------------------------------------------------------------------------------------
# Every buffer the modules create on the per-iteration path goes through `empty` / `zeros` / `empty_like`.  Normally that
# is torch.empty.  With a `RolloutArena` active the n-th request of an iteration returns the SAME memory as the n-th
# request two iterations earlier, so the device addresses in a decoder step's argument block repeat and the step can be
# replayed as a hipGraph (EnvDropDecoder.step_graphs) -- PyTorch's caching allocator gives no such guarantee.
_arena = None


class RolloutArena:
    """Opt-in allocator for training loops with a fixed per-iteration allocation sequence.

        arena = vln.ops.RolloutArena(); vln.ops.set_arena(arena)
        for batch in loader:
            arena.begin()          # top of every iteration
            ... rollout, backward, optimizer step ...

    Lifetime: a tensor produced by the modules during iteration i shares its memory with iteration i + `generations`
    (default 2); copy (`.clone()` / `.item()`) anything that must live longer.  This is CHECKED, not only documented: every
    tensor a module returns carries a generation stamp (`stamp`), and every module entry point that receives a tensor
    (`check_live`: decoder states / ctx, critic input, loss logits) raises VlnError when the stamp says the memory has been
    handed out again since -- e.g. a trainer that keeps `hidden_states` (envdrop.py:163) across iterations.  Tensors derived
    by torch ops lose the stamp; `RolloutArena(poison=True)` (debugging) additionally fills every buffer with NaN right
    before it is handed out again, so ANY late read shows.  A request whose shape or dtype differs from the recorded
    sequence simply gets fresh memory (correct, but that step's graph will not replay)."""

    def __init__(self, generations: int = 2, poison: bool = False):
        self.gens = [[] for _ in range(max(1, generations))]
        self.g = 0
        self.i = 0
        self.misses = 0
        self.epoch = 0             # iterations begun: the generation stamp of everything handed out since
        self.poison = bool(poison)

    def begin(self):
        self.g = (self.g + 1) % len(self.gens)
        self.i = 0
        self.epoch += 1
        if self.poison:
            for t in self.gens[self.g]:
                if t.is_floating_point():
                    t.fill_(float("nan"))

    def get(self, shape, dtype, device, alias=True):
        lst = self.gens[self.g]
        i = self.i
        self.i = i + 1
        if i < len(lst):
            t = lst[i]
            if t.shape != shape or t.dtype != dtype or t.device != device:
                self.misses += 1
                t = lst[i] = torch.empty(shape, dtype=dtype, device=device)
        else:
            t = torch.empty(shape, dtype=dtype, device=device)
            lst.append(t)
        # a fresh alias per request: autograd metadata (grad_fn) of an earlier iteration must not stick to the memory
        return t.detach() if alias else t

    def reserve(self, i0: int, n: int, first_ptr: int) -> bool:
        """True if the next `n` requests of this generation are the recorded buffers starting at index i0 (the caller
        then reuses its own aliases of them and advances `i` by n itself)."""
        lst = self.gens[self.g]
        return self.i == i0 and i0 + n <= len(lst) and n > 0 and lst[i0].data_ptr() == first_ptr


def stamp(*tensors):
    """Mark module outputs that live in arena memory with the arena's current generation (no-op without an arena)."""
    a = _arena
    if a is not None:
        mark = (a, a.epoch)
        for t in tensors:
            if t is not None:
                t._vln_born = mark
    return tensors[0] if len(tensors) == 1 else tensors


def check_live(t, who: str):
    """Raise if `t` was produced under a RolloutArena whose memory has been handed out again since (see RolloutArena)."""
    mark = getattr(t, "_vln_born", None)
    if mark is not None:
        a, born = mark
        if a.epoch - born >= len(a.gens):
            raise _lib.VlnError(f"{who}: this tensor was produced {a.epoch - born} iterations ago under ops.RolloutArena(generations="
                                f"{len(a.gens)}); its memory has been handed out again -- clone() what must outlive an iteration")


def set_arena(arena: Optional["RolloutArena"]):
    global _arena
    _arena = arena


def current_arena() -> Optional["RolloutArena"]:
    return _arena


def empty(*size, dtype=torch.float32, device=None):
    if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)):
        size = tuple(size[0])
    a = _arena
    if a is not None and device is not None:
        return a.get(torch.Size(size), dtype, device if isinstance(device, torch.device) else torch.device(device))
    return torch.empty(size, dtype=dtype, device=device)


def zeros(*size, dtype=torch.float32, device=None):
    if _arena is None:
        if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)):
            size = tuple(size[0])
        return torch.zeros(size, dtype=dtype, device=device)
    return empty(*size, dtype=dtype, device=device).zero_()


def zeros_like(t: torch.Tensor):
    if _arena is None:
        return torch.zeros_like(t)
    return empty(t.shape, dtype=t.dtype, device=t.device).zero_()


def empty_like(t: torch.Tensor):
    if _arena is None:
        return torch.empty_like(t)
    return empty(t.shape, dtype=t.dtype, device=t.device)


_ws_cache = {}


def workspace(device, floats: int) -> torch.Tensor:
    """Per-(device, stream) split-K scratch, grown on demand.  Reuse is ordered by the stream it belongs to, so work
    issued on a side stream never shares scratch with the main stream."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), _lib.raw_stream())
    w = _ws_cache.get(key)
    if w is None or w.numel() < floats:
        w = torch.empty(max(floats, 1 << 22), dtype=torch.float32, device=device)
        _ws_cache[key] = w
    return w


def linear_fwd(x, w, bias=None, act=ACT_NONE, out=None, split=False):
    """y = act(x @ w.T + bias); x [M,K] fp32 (row stride free), w [N,K] fp32|bf16.  split (fp32 `w` only): the product on the bf16
    matrix pipe with both operands split hi + lo (VLN_F32S) instead of the exact fp32 MFMA -- the bf16 mode's fp32-streamed
    matrices (functional.wdtype).  split="x6": three bf16 pieces per operand, six products (VLN_F32X): fp32-grade, for products in
    front of a ReLU."""
    lib = _lib.load()
    _req(x, "x"); _req(w, "w", None)
    M, K = x.shape
    N = w.shape[0]
    assert w.shape[1] == K
    if out is None:
        out = empty(M, N, dtype=torch.float32, device=x.device)
    ws = workspace(x.device, min(16 * M * N, 1 << 24))
    wt = (F32X if split == "x6" else F32S) if (split and w.dtype == torch.float32) else _dt(w)
    _lib.check(lib.vln_linear_fwd(_p(x), x.stride(0), _p(w), wt, w.stride(0), _p(out), out.stride(0), M, N, K,
                                  _p(bias), act, _p(ws), ws.numel(), _stream()), "vln_linear_fwd")
    return out


def linear_fwd_post(x, w, out):
    """POST out = x @ w.T (no bias, no activation; `w` fp32 or bf16 as stored) for the NEXT `WgradBatch.run()` of this thread to issue as
    extra workgroups of its pack launch (`vln_linear_fwd_post`): the encoder backward's d x beside the pack of the same dgates.  Call
    `linear_fwd_post_flush()` after that run: it issues a post nobody took.  Same bits as `linear_fwd`."""
    _req(x, "x"); _req(w, "w", None); _req(out, "out")
    M, K = x.shape
    N = w.shape[0]
    assert w.shape[1] == K and tuple(out.shape) == (M, N)
    _lib.check(_lib.load().vln_linear_fwd_post(_p(x), x.stride(0), _p(w), _dt(w), w.stride(0), _p(out), out.stride(0), M, N, K),
               "vln_linear_fwd_post")
    return out


def linear_fwd_post_flush(device, M=0, N=0):
    ws = workspace(device, max(1 << 22, min(16 * M * N, 1 << 24)))
    _lib.check(_lib.load().vln_linear_fwd_post_flush(_p(ws), ws.numel(), _stream()), "vln_linear_fwd_post_flush")


def linear_fwd_slabs(x, w, split=False, ws_floats: Optional[int] = None):
    """x @ w.T left as its split-K partial slabs: a [n, M, N] view of the shared workspace (valid until the next op that uses the
    workspace) whose sum over n, in order, is the product -- hand it to a consumer that adds the partials while loading
    (`lstm_pointwise_fwd`): the form the one-call decoder steps use between the LSTM's gate product and its pointwise stage."""
    lib = _lib.load()
    _req(x, "x"); _req(w, "w", None)
    M, K = x.shape
    N = w.shape[0]
    assert w.shape[1] == K
    ws = workspace(x.device, ws_floats if ws_floats is not None else min(16 * M * N, 1 << 24))
    wt = (F32X if split == "x6" else F32S) if (split and w.dtype == torch.float32) else _dt(w)
    n = C.c_int32(0)
    _lib.check(lib.vln_linear_fwd_slabs(_p(x), x.stride(0), _p(w), wt, w.stride(0), M, N, K, _p(ws), ws.numel(), C.byref(n), _stream()),
               "vln_linear_fwd_slabs")
    return ws[:n.value * M * N].view(n.value, M, N)


def linear_wgrad(dy, x, out=None, accumulate=False, split_bf16=False):
    """dW[N,K] (+)= dy[Mt,N].T @ x[Mt,K].  split_bf16: both operands as bf16 hi + lo planes on the bf16 MFMA (three
    products, fp32 accumulation) instead of the exact fp32 MFMA -- the bf16 compute mode's weight gradients, and (round 6) the
    fp32 compute mode's too unless `set_wgrad_precision_fp32("exact")`: the flag is then taken as set."""
    split_bf16 = bool(split_bf16) or _WGRAD_F32[0] == 1
    lib = _lib.load()
    _req(dy, "dy"); _req(x, "x")
    Mt, N = dy.shape
    K = x.shape[1]
    assert x.shape[0] == Mt
    if out is None:
        out = torch.empty(N, K, dtype=torch.float32, device=x.device)   # a weight gradient: may become p.grad
        accumulate = False
    ws = workspace(x.device, 1 << 22)
    _lib.check(lib.vln_linear_wgrad_p(_p(dy), dy.stride(0), _p(x), x.stride(0), _p(out), out.stride(0), Mt, N, K,
                                      int(accumulate), 1 if split_bf16 else 0, _p(ws), ws.numel(), _stream()), "vln_linear_wgrad_p")
    return out


def colsum(a, out=None, accumulate=False):
    lib = _lib.load()
    _req(a, "a")
    rows, cols = a.shape
    if out is None:
        out = torch.empty(cols, dtype=torch.float32, device=a.device)   # a bias gradient: may become p.grad
        accumulate = False
    ws = workspace(a.device, 1 << 22)
    _lib.check(lib.vln_colsum(_p(a), a.stride(0), _p(out), rows, cols, int(accumulate), _p(ws), ws.numel(), _stream()),
               "vln_colsum")
    return out


def bn0_grads_from_wgrad(dW, db, W, gamma, beta, gW, gb, g_gamma, g_beta, accumulate=True):
    """The input BatchNorm's d gamma / d beta of a BN-MLP from its first Linear layer's weight / bias gradient of ONE rollout
    (include/vln_hip.h, vln_bn0_grads_from_wgrad): d gamma_k = sum_n (dW[n,k] - beta_k db[n]) / gamma_k W[n,k], d beta_k =
    sum_n db[n] W[n,k]; dW / db are handed on into gW / gb.  accumulate: the four targets are added to (else stored)."""
    lib = _lib.load()
    N, K = W.shape
    for t, what in ((dW, "dW"), (W, "W"), (gW, "gW")):
        assert t.shape == (N, K) and t.dtype == torch.float32 and t.stride(1) == 1, what
    assert dW.is_contiguous() and gW.is_contiguous()
    ws = workspace(W.device, 64 * K)
    acc = 1 if accumulate else 0
    _lib.check(lib.vln_bn0_grads_from_wgrad(dW.data_ptr(), db.data_ptr(), W.data_ptr(), W.stride(0), gamma.data_ptr(), beta.data_ptr(),
                                            gW.data_ptr(), _p(gb), _p(g_gamma), _p(g_beta), N, K, acc, acc, acc, ws.data_ptr(), ws.numel(),
                                            _stream()), "vln_bn0_grads_from_wgrad")


def transpose_cast(w, dtype=torch.float32, out=None):
    lib = _lib.load()
    _req(w, "w")
    N, K = w.shape
    if out is None:
        out = torch.empty(K, N, dtype=dtype, device=w.device)           # a weight shadow: lives across iterations
    _lib.check(lib.vln_transpose_cast(_p(w), w.stride(0), _p(out), _dt(out), out.stride(0), N, K, _stream()),
               "vln_transpose_cast")
    return out


def cast_copy(w, dtype=torch.bfloat16, out=None):
    lib = _lib.load()
    _req(w, "w")
    w2 = w.reshape(-1, w.shape[-1])
    if out is None:
        out = torch.empty(w.shape, dtype=dtype, device=w.device)
    o2 = out.reshape(-1, out.shape[-1])
    _lib.check(lib.vln_cast_copy(_p(w2), w2.stride(0), _p(o2), _dt(o2), o2.stride(0), w2.shape[0], w2.shape[1],
                                 _stream()), "vln_cast_copy")
    return out


class ShadowBatch:
    """Collects (matrix -> compute-dtype copy / transposed copy) jobs and issues them as ONE launch
    (`vln_shadow_refresh`): a module's weight shadows are refreshed once per optimizer step."""

    def __init__(self):
        self.jobs = []
        self.keep = []

    def add(self, src, dst=None, dst_t=None, src2=None):
        """src [N,K] fp32 (row stride free); dst [N,K] / dst_t [K,N] preallocated (fp32 or bf16, same dtype)."""
        N, K = src.shape
        ref = dst if dst is not None else dst_t
        j = _lib.ShadowJob(src.data_ptr(), _p(src2), _p(dst), _p(dst_t), src.stride(0), 0 if dst is None else dst.stride(0),
                           0 if dst_t is None else dst_t.stride(0), N, K, _dt(ref), 0)
        self.jobs.append(j)
        self.keep += [src, src2, dst, dst_t]

    # While a collector list is installed (`collect()`), run() / replay() hand their job arrays to it instead of launching: the
    # caller issues them together with other work in ONE launch (runtime.DeviceClock.prologue -> vln_prologue).
    _collector = None

    @classmethod
    def collect(cls):
        import contextlib

        @contextlib.contextmanager
        def scope():
            prev, cls._collector = cls._collector, []
            try:
                yield cls._collector
            finally:
                cls._collector = prev
        return scope()

    def run(self):
        """Issues the launch; returns a replayable handle (the job array is valid as long as the sources and destinations
        keep their addresses: `ShadowBatch.replay(handle)` refreshes again without rebuilding the jobs)."""
        if not self.jobs:
            return None
        arr = (_lib.ShadowJob * len(self.jobs))(*self.jobs)
        handle = (arr, len(self.jobs), self.keep)
        self.jobs, self.keep = [], []
        ShadowBatch.replay(handle)
        return handle

    @staticmethod
    def replay(handle):
        arr, n, _keep = handle
        if ShadowBatch._collector is not None:
            ShadowBatch._collector.append(handle)
            return
        _lib.check(_lib.load().vln_shadow_refresh(arr, n, _stream()), "vln_shadow_refresh")


class GradRide:
    """Collects the WgradBatch / ColsumBatch launches issued inside `with GradRide.collect():` and, instead of launching them, posts
    them as ONE gradient ride (`vln_wgrad_ride_post`): the next `vln_lstm_seq_bwd` on the stream carries them as passenger workgroups
    of its persistent launch (csrc/wgrad_ride.h) or issues them itself; `GradRide.flush()` issues a ride nobody carried.  Batches the
    ride cannot take (other row counts, more than 8 products / 4 column sums, exact-fp32 precision) launch as usual."""
    MAX_W, MAX_C = 8, 4
    _active = None
    _posted = None

    def __init__(self):
        self.w, self.c, self.keep, self.rows, self.prec, self.dev = [], [], [], None, None, None

    @classmethod
    def collect(cls):
        import contextlib

        @contextlib.contextmanager
        def scope():
            prev, cls._active = cls._active, cls()
            try:
                yield cls._active
            finally:
                ride, cls._active = cls._active, prev
            ride.post()
        return scope()

    def take_w(self, jobs, keep, rows, prec):
        if prec not in (1, 2) or (self.rows not in (None, rows)) or (self.prec not in (None, prec)) or len(self.w) + len(jobs) > self.MAX_W:
            return False
        self.w += jobs; self.keep += keep; self.rows, self.prec, self.dev = rows, prec, keep[0].device
        return True

    def take_c(self, jobs, keep, rows):
        if not self.w or self.rows != rows or len(self.c) + len(jobs) > self.MAX_C:       # column sums only ride along with products
            return False
        self.c += jobs; self.keep += keep
        return True

    def post(self):
        if not self.w:
            return
        MS = (self.rows + 31) // 32
        area = sum(2 * ((j.N + 15) // 16 + (j.K + 15) // 16) * MS * 1024 for j in self.w) // 4
        ws = workspace(self.dev, max(1 << 22, area))
        wa = (_lib.WgradJob * len(self.w))(*self.w)
        ca = (_lib.ColsumJob * len(self.c))(*self.c) if self.c else None
        _lib.check(_lib.load().vln_wgrad_ride_post(wa, len(self.w), ca, len(self.c), self.rows, self.prec, _p(ws), ws.numel(), _stream()),
                   "vln_wgrad_ride_post")
        GradRide._posted = (ws, self.keep)           # alive until the next ride is posted (the carrying launch is long past by then)
        self.w, self.c, self.keep = [], [], []

    @staticmethod
    def try_add(dy, x, out, split_bf16=True) -> bool:
        """One more product out += dy.T @ x, over FEWER rows than the pending ride's, joins the ride posted on this stream
        (vln_wgrad_ride_add): the encoder head's weight gradient (64 rows) beside the decoder's (steps x batch rows) -- carried by
        the same backward recurrence launch instead of standing in front of it.  False: nothing pending / no room / another
        precision -- the caller launches the product itself."""
        if GradRide._posted is None:
            return False
        _req(dy, "dy"); _req(x, "x"); _req(out, "out")
        Mt, N = dy.shape
        K = x.shape[1]
        job = _lib.WgradJob(dy.data_ptr(), x.data_ptr(), out.data_ptr(), dy.stride(0), x.stride(0), out.stride(0), N, K, 1, 0)
        ok = bool(_lib.load().vln_wgrad_ride_add(C.byref(job), Mt, wgrad_precision(split_bf16), _stream()))
        if ok:
            GradRide._posted = (GradRide._posted[0], GradRide._posted[1] + [dy, x, out])
        return ok

    @staticmethod
    def flush():
        _lib.check(_lib.load().vln_wgrad_ride_flush(_stream()), "vln_wgrad_ride_flush")

    @staticmethod
    def stats():
        out = (C.c_int64 * 2)()
        _lib.check(_lib.load().vln_wgrad_ride_stats(out), "vln_wgrad_ride_stats")
        return {"carried": int(out[0]), "issued_alone": int(out[1])}


class ColsumBatch:
    """Bias gradients over the SAME rows as one launch (`vln_colsum_grouped`): add(a, out1, out2=None, accumulate)."""

    def __init__(self):
        self.jobs, self.keep, self.rows = [], [], None

    def add(self, a, out1, out2=None, accumulate=False):
        _req(a, "a")
        rows, cols = a.shape
        assert self.rows is None or self.rows == rows
        self.rows = rows
        if cols % 4 or a.stride(0) % 4 or a.data_ptr() % 16:          # odd shapes: the single-matrix kernel
            colsum(a, out1, accumulate)
            if out2 is not None:
                colsum(a, out2, accumulate)
            return
        self.jobs.append(_lib.ColsumJob(a.data_ptr(), _p(out1), _p(out2), a.stride(0), cols, int(accumulate)))
        self.keep += [a, out1, out2]

    def post(self) -> bool:
        """POST the jobs (at most 12) for the next `WgradBatch.run()` that carries a posted product (`linear_fwd_post`) to issue in the
        same launch; call `flush()` after that run.  False: not posted (a gradient ride is collecting, too many jobs): call run()."""
        if not self.jobs or len(self.jobs) > 12 or GradRide._active is not None:
            return False
        arr = (_lib.ColsumJob * len(self.jobs))(*self.jobs)
        _lib.check(_lib.load().vln_colsum_post(arr, len(self.jobs), self.rows), "vln_colsum_post")
        return True

    def flush(self):
        """After the run that could have taken the post: issues the sums as their own launch if it did not."""
        ws = workspace(self.keep[0].device, 1 << 22)
        _lib.check(_lib.load().vln_colsum_post_flush(_p(ws), ws.numel(), _stream()), "vln_colsum_post_flush")
        self.jobs, self.keep, self.rows = [], [], None

    def run(self):
        lib = _lib.load()
        if self.jobs and GradRide._active is not None and GradRide._active.take_c(self.jobs, self.keep, self.rows):
            self.jobs, self.keep, self.rows = [], [], None
            return
        for i in range(0, len(self.jobs), 12):
            chunk = self.jobs[i:i + 12]
            arr = (_lib.ColsumJob * len(chunk))(*chunk)
            ws = workspace(self.keep[0].device, 1 << 22)
            _lib.check(lib.vln_colsum_grouped(arr, len(chunk), self.rows, _p(ws), ws.numel(), _stream()), "vln_colsum_grouped")
        self.jobs, self.keep, self.rows = [], [], None


# Weight gradients in bf16 compute mode:
#   "bf16"  (default) plain bf16 operands, fp32 accumulation -- ONE MFMA per product, what mixed-precision training computes;
#           gradients within 2-5e-3 of the fp64 result on the same weights (measured, tests/parity.py SAME_BF16_GRAD), inside
#           north_star's 1e-2 bound for bf16;
#   "split" both operands as hi + lo bf16 planes, three MFMAs per product: fp32-grade gradients (5e-5 of the fp64 result) for
#           +50 us per EnvDrop iteration (1.80 -> 1.85 ms, profiles/round2_notes.md).
_WGRAD_BF16 = [1 if __import__('os').environ.get('VLN_WGRAD') == 'split' else 2]      # VLN_WGRAD=split: process-wide default


def set_wgrad_precision(mode: str):
    """"bf16" (default) or "split": the form of the weight-gradient contractions in bf16 compute mode (see above)."""
    if mode not in ("split", "bf16"):
        raise ValueError("wgrad precision: 'split' or 'bf16'")
    _WGRAD_BF16[0] = 1 if mode == "split" else 2


def get_wgrad_precision() -> str:
    return "split" if _WGRAD_BF16[0] == 1 else "bf16"


# ... and in fp32 compute mode (round 6):
#   "split" (default) the same hi + lo planes, three bf16 MFMAs per product, as ONE packed contraction per rollout: every fp32
#           parity test holds 1e-4 with it (worst weight gradient 1.6e-5 of the oracle's, profiles/round6_notes.md section 10);
#           Self-Monitor B 128 5.25 -> 4.40 ms, EnvDrop fp32 2.12 -> 1.93 ms;
#   "exact" the fp32 MFMA (v_mfma_f32_16x16x4_f32: 1/16 of the bf16 MFMA rate), one launch per product and step -- rounds 1-5.
_WGRAD_F32 = [0 if __import__('os').environ.get('VLN_WGRAD_FP32') == 'exact' else 1]    # VLN_WGRAD_FP32=exact: process-wide default


def set_wgrad_precision_fp32(mode: str):
    """"split" (default) or "exact": the form of the weight-gradient contractions in fp32 compute mode (see above)."""
    if mode not in ("split", "exact"):
        raise ValueError("fp32 wgrad precision: 'split' or 'exact'")
    _WGRAD_F32[0] = 1 if mode == "split" else 0


def get_wgrad_precision_fp32() -> str:
    return "split" if _WGRAD_F32[0] == 1 else "exact"


def wgrad_precision(lp: bool) -> int:
    return _WGRAD_BF16[0] if lp else _WGRAD_F32[0]


class WgradBatch:
    """Collects dW (+)= dy.T @ x products over the SAME rows and issues them as one launch (`vln_wgrad_grouped`)."""

    def __init__(self, split_bf16=False, never_plain=False):
        """never_plain: in bf16 mode use split operands even when the process default is the plain-bf16 form (gradients behind a
        BatchNorm: csrc/bn_mlp.hip does the same in the C call)."""
        self.jobs, self.keep, self.Mt, self.split, self.never_plain = [], [], None, split_bf16, never_plain

    def add(self, dy, x, out, accumulate=False):
        _req(dy, "dy"); _req(x, "x"); _req(out, "out")
        Mt, N = dy.shape
        K = x.shape[1]
        assert x.shape[0] == Mt and out.shape == (N, K) and (self.Mt is None or self.Mt == Mt)
        self.Mt = Mt
        self.jobs.append(_lib.WgradJob(dy.data_ptr(), x.data_ptr(), out.data_ptr(), dy.stride(0), x.stride(0), out.stride(0),
                                       N, K, int(accumulate), 0))
        self.keep += [dy, x, out]

    def run(self):
        if not self.jobs:
            return
        arr = (_lib.WgradJob * len(self.jobs))(*self.jobs)
        # workspace of the packed form: both bf16 planes of every operand in fragment order + (few output tiles, many
        # rows: the encoder's L*B) the slabs of a row split
        MS = (self.Mt + 31) // 32
        tiles = sum(((j.N + 127) // 128) * ((j.K + 127) // 128) for j in self.jobs)
        msplit = max(1, min(256 // max(tiles, 1), MS // 4)) if tiles < 256 else 1
        area = sum(2 * ((j.N + 15) // 16 + (j.K + 15) // 16) * MS * 1024 for j in self.jobs) // 4
        ws = workspace(self.keep[0].device, max(1 << 22, area + (msplit * sum(j.N * j.K for j in self.jobs) if msplit > 1 else 0)))
        prec = wgrad_precision(self.split)
        if prec == 2 and self.never_plain:
            prec = 1
        if GradRide._active is not None and GradRide._active.take_w(self.jobs, self.keep, self.Mt, prec):
            self.jobs, self.keep, self.Mt = [], [], None
            return
        _lib.check(_lib.load().vln_wgrad_grouped(arr, len(self.jobs), self.Mt, prec, _p(ws), ws.numel(),
                                                 _stream()), "vln_wgrad_grouped")
        self.jobs, self.keep, self.Mt = [], [], None


def select_rows_multi(srcs, indices):
    """[src_t[arange(B), index_t] for t] in ONE launch (`vln_select_rows_multi`): the previous-action rows of a teacher-forced rollout
    (`a_t_prev = a_t_cand[arange, a_t]`, monitor.py:191 / follower.py:164) -- every step's row is known when the rollout starts.
    srcs: [B, C_t, F] fp32 contiguous; indices: [B] int64.  Returns views of one [T, B, F] buffer (no gradient)."""
    lib = _lib.load()
    T = len(srcs)
    B, _, F = srcs[0].shape
    out = empty(T, B, F, dtype=torch.float32, device=srcs[0].device)
    res = []
    for i in range(0, T, _lib.CE_MAX_STEPS):
        steps = []
        for t in range(i, min(T, i + _lib.CE_MAX_STEPS)):
            x, ix = srcs[t].detach(), indices[t]
            _req(x, "src")
            assert x.is_contiguous() and x.shape[0] == B and x.shape[2] == F and ix.dtype == torch.int64 and ix.is_contiguous()
            steps.append(_lib.SelectStep(x.data_ptr(), ix.data_ptr(), out[t].data_ptr(), x.shape[1]))
        arr = (_lib.SelectStep * len(steps))(*steps)
        _lib.check(lib.vln_select_rows_multi(arr, len(steps), B, F, _stream()), "vln_select_rows_multi")
    for t in range(T):
        res.append(out[t])
    return res


def attn_dot(ctx, vec):
    """dots[b,s] = ctx[b,s,:] . vec[b,:]"""
    lib = _lib.load()
    _req(ctx, "ctx", None); _req(vec, "vec")
    B, S, D = ctx.shape
    assert ctx.is_contiguous()
    dots = empty(B, S, dtype=torch.float32, device=ctx.device)
    _lib.check(lib.vln_attn_dot(_p(ctx), _dt(ctx), _p(vec), vec.stride(0), _p(dots), B, S, D, _stream()),
               "vln_attn_dot")
    return dots


def attn_softmax_wsum(ctx, logits, mask=None, out=None):
    lib = _lib.load()
    _req(ctx, "ctx", None); _req(logits, "logits")
    B, S, D = ctx.shape
    assert ctx.is_contiguous() and logits.is_contiguous()
    attn = empty(B, S, dtype=torch.float32, device=ctx.device)
    if out is None:
        out = empty(B, D, dtype=torch.float32, device=ctx.device)
    m8 = None
    if mask is not None:
        m8 = mask.to(torch.uint8).contiguous()
    _lib.check(lib.vln_attn_softmax_wsum(_p(ctx), _dt(ctx), _p(logits), _p(m8), _p(attn), _p(out), out.stride(0),
                                         B, S, D, _stream()), "vln_attn_softmax_wsum")
    return out, attn


def rows_wsum(ctx, w, out=None):
    lib = _lib.load()
    _req(ctx, "ctx", None); _req(w, "w")
    B, S, D = ctx.shape
    assert ctx.is_contiguous() and w.is_contiguous()
    if out is None:
        out = empty(B, D, dtype=torch.float32, device=ctx.device)
    _lib.check(lib.vln_rows_wsum(_p(ctx), _dt(ctx), _p(w), _p(out), out.stride(0), B, S, D, _stream()),
               "vln_rows_wsum")
    return out


def attn_sync_buffer(B, device):
    """Zero-initialised exchange scratch for the four-workgroups-per-row attention (csrc/attention_split.h): keep it for the
    attention calls alone, for as long as they are issued."""
    return torch.zeros((int(_lib.load().vln_attn_sync_bytes(B)) + 3) // 4, dtype=torch.int32, device=device)


def attn_fwd_rows(ctx, vec, mask=None, out=None, want_attn=True, sync=None):
    """softmax(mask(ctx . vec)) and the weighted sum of the context rows in ONE launch (units.py:106-118).
    sync = attn_sync_buffer(B, device): four workgroups per row instead of one."""
    lib = _lib.load()
    _req(ctx, "ctx", None); _req(vec, "vec")
    B, S, D = ctx.shape
    assert ctx.is_contiguous()
    attn = empty(B, S, dtype=torch.float32, device=ctx.device) if want_attn else None
    if out is None:
        out = empty(B, D, dtype=torch.float32, device=ctx.device)
    m8 = None
    if mask is not None:
        m8 = mask.view(torch.uint8) if (mask.dtype == torch.bool and mask.is_contiguous()) else mask.to(torch.uint8).contiguous()
    scratch = empty(B, S, dtype=torch.float32, device=ctx.device)
    _lib.check(lib.vln_attn_fwd_rows(_p(ctx), _dt(ctx), _p(vec), vec.stride(0), _p(m8), _p(attn), _p(out), out.stride(0),
                                     _p(scratch), B, S, D, _p(sync), 0 if sync is None else sync.numel() * sync.element_size(), _stream()),
               "vln_attn_fwd_rows")
    return out, attn


def attn_bwd_rows(ctx, attn, dwc, dattn_ext=None, want_dl=False, out=None, sync=None):
    """Backward of attn_fwd_rows w.r.t. the query: returns (dvec [B,D], dl [B,S] or None)."""
    lib = _lib.load()
    _req(ctx, "ctx", None); _req(attn, "attn"); _req(dwc, "dwc")
    B, S, D = ctx.shape
    assert ctx.is_contiguous() and attn.is_contiguous()
    dvec = out if out is not None else empty(B, D, dtype=torch.float32, device=ctx.device)
    dl = empty(B, S, dtype=torch.float32, device=ctx.device) if want_dl else None
    if dattn_ext is not None:
        dattn_ext = dattn_ext.contiguous()
    scratch = empty(B, S, dtype=torch.float32, device=ctx.device)
    _lib.check(lib.vln_attn_bwd_rows(_p(ctx), _dt(ctx), _p(attn), _p(dwc), dwc.stride(0), _p(dattn_ext), _p(dvec),
                                     dvec.stride(0), _p(dl), _p(scratch), B, S, D, _p(sync),
                                     0 if sync is None else sync.numel() * sync.element_size(), _stream()), "vln_attn_bwd_rows")
    return dvec, dl


def attn_dctx_deferred(alpha_ptrs, dl_ptrs, g_ptrs, ldg, q_ptrs, ldq, out, accumulate=False, drop=None, dk=None):
    """out[b,s,:] (+)= sum_t (alpha_t[b,s] g_t[b,:] + dl_t[b,s] q_t[b,:]) [* mask_t]; the four lists hold T device
    addresses ((dl_t, q_t) may both be None: a pure outer product); drop = [(seed, offset, p)] per step multiplies step
    t's term by that dropout mask over the flat [B,S,D] index (the attended tensor's own per-step dropout).
    dk ([B,S,D], no drop): the (dl, q) half lands there instead of in `out` (the projected-context form, runtime.CtxGate)."""
    lib = _lib.load()
    T = len(alpha_ptrs)
    B, S, D = out.shape
    assert out.is_contiguous() and len(dl_ptrs) == T and len(g_ptrs) == T and len(q_ptrs) == T
    arr = (C.c_void_p * (4 * T))(*alpha_ptrs, *dl_ptrs, *g_ptrs, *q_ptrs)
    base = C.addressof(arr)
    if drop is None and (dk is not None or None not in dl_ptrs):
        assert dk is None or (dk.is_contiguous() and dk.shape == out.shape)
        _lib.check(lib.vln_attn_dctx_deferred(base, base + 8 * T, base + 16 * T, ldg, base + 24 * T, ldq, T, _p(out), B, S, D,
                                              1 if accumulate else 0, _p(dk), _stream()), "vln_attn_dctx_deferred")
        return out
    drop = drop or [(0, 0, 0.0)] * T
    seeds = (C.c_uint64 * T)(*[d[0] for d in drop]); offs = (C.c_uint64 * T)(*[d[1] for d in drop])
    ps = (C.c_float * T)(*[d[2] for d in drop])
    bases = {d[3] for d in drop if len(d) > 3}            # (seed, offset, p[, device offset base]): one clock for all steps
    if len(bases) > 1:
        raise ValueError("attn_dctx_deferred: the steps' dropout offsets must share one device base")
    _lib.check(lib.vln_attn_dctx_deferred_drop(base, base + 8 * T, base + 16 * T, ldg, base + 24 * T, ldq, T, _p(out), B, S, D,
                                               1 if accumulate else 0, C.addressof(seeds), C.addressof(offs), C.addressof(ps),
                                               bases.pop() if bases else None, _stream()),
               "vln_attn_dctx_deferred_drop")
    return out


def attn_bwd(ctx, attn, dalpha, dattn_ext=None, dwc=None, vec=None, dctx=None, want_dl=False):
    """Returns (dvec, dl).  If dctx is given it is accumulated in place."""
    lib = _lib.load()
    B, S, D = ctx.shape
    dvec = empty(B, D, dtype=torch.float32, device=ctx.device)
    dl = empty(B, S, dtype=torch.float32, device=ctx.device) if want_dl else None
    _lib.check(lib.vln_attn_bwd(_p(ctx), _dt(ctx), _p(attn), _p(dalpha), _p(dattn_ext), _p(dwc),
                                dwc.stride(0) if dwc is not None else 0, _p(vec),
                                vec.stride(0) if vec is not None else 0, _p(dvec), dvec.stride(0), _p(dctx), _p(dl),
                                B, S, D, _stream()), "vln_attn_bwd")
    return dvec, dl


EW_MUL, EW_ADD_SCALAR, EW_TANH_GRAD, EW_MUL_ROWSUM = 0, 1, 2, 3


def ew(op: int, a, b, out=None, nb: int = 0, bcast: bool = False):
    """Row-wise elementwise forms (vln_ew): a [R,C] (row stride free; None for the plain row sums of op 3), b [R,C] /
    [R,nb] or, with `bcast`, one row vector / scalar."""
    lib = _lib.load()
    if a is not None:
        R, Cn = a.shape
    else:
        R, Cn = b.shape[0], 1
    if out is None:
        out = empty(R, Cn, dtype=torch.float32, device=b.device)
    ldb = 0 if bcast else (b.stride(0) if b.dim() == 2 else 0)
    _lib.check(lib.vln_ew(op, _p(a), a.stride(0) if a is not None else 0, _p(b), ldb, nb, _p(out), out.stride(0), R, Cn, _stream()),
               "vln_ew")
    return out


def dropout_mask(n: int, seed: int, offset: int, p: float, device) -> torch.Tensor:
    """The exact pre-scaled keep mask (0 or 1/(1-p)) the kernels use for (seed, offset)."""
    lib = _lib.load()
    out = empty(n, dtype=torch.float32, device=device)
    _lib.check(lib.vln_dropout_mask(_p(out), n, seed, offset, p, _stream()), "vln_dropout_mask")
    return out


def scale_dropout(x, seed: int, offset: int, p: float, base: Optional[int] = None, out=None):
    """y = x * (pre-scaled keep mask of (seed, offset)), x [rows, cols] fp32 -- the same mask `dropout_mask(rows * cols, ...)` returns.
    base: address of a DEVICE word; the Philox offset is then word * 8 + offset (runtime.DeviceClock)."""
    lib = _lib.load()
    _req(x, "x")
    rows, cols = x.shape
    if out is None:
        out = empty(rows, cols, dtype=torch.float32, device=x.device)
    _lib.check(lib.vln_scale_dropout(_p(x), x.stride(0), _p(out), out.stride(0), rows, cols, seed, offset, p, base, _stream()),
               "vln_scale_dropout")
    return out


def feat_dropout_inplace(x, img: int, angle: int, seed: int, offset: int, p: float, copy_bf16=None, base=None):
    """base: device word the kernel adds (x 8) to `offset` (runtime.DeviceClock), or None (an absolute host offset)."""
    lib = _lib.load()
    _req(x, "x", None)
    assert x.is_contiguous() and x.shape[-1] == img + angle
    rows = x.numel() // (img + angle)
    _lib.check(lib.vln_feat_dropout_inplace(_p(x), _dt(x), rows, img, angle, seed, offset, p, _p(copy_bf16), base,
                                            _stream()), "vln_feat_dropout_inplace")
    return x


def lstm_pointwise_fwd(gates, b_ih, b_hh, c0, seed=0, offset=0, p=0.0, want_drop=False):
    """gates: [nsplit,B,4H] pre-activation slabs."""
    lib = _lib.load()
    ns, B, H4 = gates.shape
    H = H4 // 4
    dev = gates.device
    h1 = empty(B, H, device=dev); c1 = empty(B, H, device=dev)
    act = empty(B, H4, device=dev); tc = empty(B, H, device=dev)
    hd = empty(B, H, device=dev) if want_drop else None
    _lib.check(lib.vln_lstm_pointwise_fwd(_p(gates), ns, B * H4, _p(b_ih), _p(b_hh), _p(c0), _p(h1), _p(c1), _p(act),
                                          _p(tc), _p(hd), seed, offset, p, B, H, _stream()), "vln_lstm_pointwise_fwd")
    return h1, c1, act, tc, hd


def lstm_pointwise_bwd(dh1, dh1_drop, dc1, act, tanh_c1, c0, seed=0, offset=0, p=0.0):
    lib = _lib.load()
    B, H = c0.shape
    dg = empty(B, 4 * H, device=c0.device); dc0 = empty(B, H, device=c0.device)
    _lib.check(lib.vln_lstm_pointwise_bwd(_p(dh1), _p(dh1_drop), _p(dc1), seed, offset, p, _p(act), _p(tanh_c1),
                                          _p(c0), _p(dg), _p(dc0), B, H, _stream()), "vln_lstm_pointwise_bwd")
    return dg, dc0


def attn_textk_fwd(ctx, kctx, mask, gates, b_ih, b_hh, c0, sync, seed=0, offset=0, p=0.0):
    """The projected-context text attention with the LSTM cell's pointwise stage in the same launch (vln_attn_textk_fwd):
    gates [n,B,4H] slabs, ctx [B,S,H] fp32|bf16, kctx [B,S,H] fp32 -> (h1, c1, act, tanh_c1, tcat [B,2H], alpha [B,S])."""
    lib = _lib.load()
    B, S, H = ctx.shape
    n = gates.shape[0]
    dev = ctx.device
    h1, c1 = empty(B, H, device=dev), empty(B, H, device=dev)
    act, tc = empty(B, 4 * H, device=dev), empty(B, H, device=dev)
    tcat, alpha = empty(B, 2 * H, device=dev), empty(B, S, device=dev)
    _lib.check(lib.vln_attn_textk_fwd(_p(ctx), _dt(ctx), _p(kctx), _p(mask), _p(gates), n, gates.stride(0), _p(b_ih), _p(b_hh), _p(c0),
                                      _p(h1), _p(c1), _p(act), _p(tc), _p(tcat), _p(alpha), seed, offset, p, B, S, H, _p(sync),
                                      sync.numel() * sync.element_size(), _stream()), "vln_attn_textk_fwd")
    return h1, c1, act, tc, tcat, alpha


def attn_textk_bwd(ctx, kctx, alpha, dtcat, dh1, dc1, act, tanh_c1, c0, sync, seed=0, offset=0, p=0.0):
    """Backward of attn_textk_fwd (vln_attn_textk_bwd): dtcat [n,B,2H] slabs -> (dq [B,H], dl [B,S], dwc [B,2H] (cols [0,H)),
    dgates [B,4H], dc0 [B,H])."""
    lib = _lib.load()
    B, S, H = ctx.shape
    n = dtcat.shape[0]
    dev = ctx.device
    dq, dl, dwc = empty(B, H, device=dev), empty(B, S, device=dev), zeros(B, 2 * H, device=dev)
    dg, dc0 = empty(B, 4 * H, device=dev), empty(B, H, device=dev)
    _lib.check(lib.vln_attn_textk_bwd(_p(ctx), _dt(ctx), _p(kctx), _p(alpha), _p(dtcat), n, dtcat.stride(0), _p(dwc), _p(dq), _p(dl),
                                      _p(dh1), _p(dc1), _p(act), _p(tanh_c1), _p(c0), _p(dg), _p(dc0), seed, offset, p, B, S, H,
                                      _p(sync), sync.numel() * sync.element_size(), _stream()), "vln_attn_textk_bwd")
    return dq, dl, dwc, dg, dc0
