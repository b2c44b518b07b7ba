"""Host-side runtime shared by the drop-in modules.

* `ShadowSet`  -- compute-dtype / transposed copies of the fp32 master weights,
  refreshed only when a parameter's version or storage changes (i.e. once per
  optimizer step), so every GEMM (forward, dX) streams a K-contiguous operand.
* `Stash`      -- the rollout arena.  Every Linear's X operand is written by the
  forward kernels straight into row-block t of a per-site [rows, dim] buffer
  and its dY operand by the backward kernels, so each weight gradient is ONE
  contraction over (steps x batch) per optimizer step instead of a
  read-modify-write of the whole dW per decoder step.
* `WeightGate` / `CtxGate` -- identity autograd nodes.  Autograd runs a gate's
  backward only after every decoder step that consumed its outputs has run
  its own backward; that is where the deferred dW GEMMs / the in-place
  accumulated dctx are handed to autograd.
"""
from __future__ import annotations

import weakref
from typing import Dict, List, Optional, Sequence

import torch

from . import ops


class DeviceClock:
    """Device-resident counters for everything a training iteration used to take from the HOST per launch: the Philox offset
    base of every dropout site and the launch sequence of the persistent recurrence's data-tagged hand-offs.  With a clock
    attached (`clock.attach(encoder, decoder, ...)`) the modules hand the kernels a DEVICE word plus a small per-call index
    instead of an absolute value, so the launch arguments of an iteration repeat and the WHOLE iteration -- encoder, decoder
    steps, loss, backward, optimizer -- can be captured as one hipGraph (`graphs.IterationGraph`) whose replays still draw
    fresh dropout masks.  `tick()` = one 1-thread launch at the top of every iteration (inside the graph when captured).

    Offsets: word[0] after k ticks = k * STRIDE; call number r (1, 2, ...) of a module since the last tick uses Philox offset
    (word + r) * 8 + site -- exactly what a host counter that stood at `host` would have produced (`value(r)`), which is how
    the tests export the masks.  Launch sequences: every registered recurrence buffer's 32-bit word is bumped by the same
    launch; before its low 24 bits wrap the exchange is cleared (between replays, on the stream)."""
    STRIDE = 64

    def __init__(self, device):
        from . import _lib
        self.device = torch.device(device)
        self.word = torch.zeros(1, dtype=torch.int64, device=self.device)
        self.ptr = self.word.data_ptr()
        self.host = 0                # value of the device word once every tick issued so far has run
        self.epoch = 0
        self._rel: Dict[object, int] = {}
        self._seqs = []              # [sync buffer (int32 tensor), seq word index, granule slice (int32 view), host mirror]
        self._counters = []          # [(int64 word tensor, increment per tick)]  e.g. FusedAdam's step count
        self._items = None

    def attach(self, *modules):
        """Every module (and every submodule that owns dropout sites) reads its offsets from this clock from now on."""
        for m in modules:
            if isinstance(m, torch.nn.Module):
                for sub in m.modules():
                    object.__setattr__(sub, "clock", self)
            else:
                m.clock = self
        return self

    def register_counter(self, word: torch.Tensor, inc: int = 1):
        """An int64 device word bumped by `inc` on every tick (an optimizer's step count)."""
        self._counters.append((word, int(inc)))
        self._items = None

    def rel(self, key, limit: Optional[int] = None) -> int:
        r = self._rel.get(key, 0) + 1
        if r > (self.STRIDE - 1 if limit is None else limit):
            raise RuntimeError(f"DeviceClock: more than {self.STRIDE - 1 if limit is None else limit} calls of one module between two ticks")
        self._rel[key] = r
        return r

    def value(self, rel: int) -> int:
        """The absolute Philox offset (in the host-counter convention) of call `rel` since the last tick."""
        return self.host + rel

    def register_sequence(self, buf: torch.Tensor, seq_byte_offset: int, gran_byte_offset: int, gran_bytes: int):
        """A recurrence scratch buffer (EncoderLSTM._sync_ws) joins: its exchange is cleared once (tags of the host-counted
        form may be anything), its sequence word starts at 0 and is bumped by every tick from now on."""
        for s in self._seqs:
            if s[0] is buf:
                return
        gran = buf[gran_byte_offset // 4:(gran_byte_offset + gran_bytes) // 4]
        gran.zero_()
        buf[seq_byte_offset // 4].zero_()
        self._seqs.append([buf, seq_byte_offset // 4, gran, 0])
        self._items = None

    def _tick_items(self):
        from . import _lib
        if self._items is None:
            rows = [(self.ptr, self.STRIDE, 8)] + [(s[0].data_ptr() + 4 * s[1], self.STRIDE, 4) for s in self._seqs] + \
                [(w.data_ptr(), inc, 8) for w, inc in self._counters]
            if len(rows) > 8:
                raise RuntimeError("DeviceClock: at most 7 recurrence buffers / counters per clock")
            self._items = ((_lib.TickItem * len(rows))(*[_lib.TickItem(p, inc, w, 0) for p, inc, w in rows]), len(rows))
        return self._items

    def _maintain(self):
        """Host mirror of the bump + the wrap guard of the 24-bit launch sequences (runs OUTSIDE a captured graph)."""
        self.host += self.STRIDE
        self.epoch += 1
        self._rel.clear()
        for s in self._seqs:
            s[3] += self.STRIDE
            if s[3] + 2 * self.STRIDE >= (1 << 24):      # clear the exchange and restart the sequence, stream-ordered
                s[2].zero_()
                s[0][s[1]].zero_()
                s[3] = 0

    def restart_sequences(self):
        """Clear every registered exchange and restart its launch sequence (called right before a graph capture, so that the
        wrap guard cannot fire INSIDE the captured iteration)."""
        for s in self._seqs:
            s[2].zero_()
            s[0][s[1]].zero_()
            s[3] = 0

    def tick(self):
        """Top of an iteration: bump the device words (one launch on the current stream) and the host mirror."""
        from . import _lib
        self._maintain()        # the clear (if due) is ordered BEFORE the bump: after it the word holds STRIDE, like the mirror
        items, n = self._tick_items()
        _lib.check(_lib.load().vln_tick(items, n, _lib.raw_stream()), "vln_tick")

    def prologue(self, feed=None, modules=()):
        """tick() + (optional) the pull of the selected batch (staging.HostBatchFeed) + the weight-shadow refresh of `modules`
        (whatever the last optimizer step staled) as ONE launch (`vln_prologue`): the three do not depend on each other and sat as
        four launches at the top of every iteration's dependent chain.  The modules' own forward then finds its shadows current."""
        from . import _lib, ops
        self._maintain()
        with ops.ShadowBatch.collect() as handles:
            for m in modules:
                m.prefresh()
        items, n = self._tick_items()
        njobs = sum(h[1] for h in handles)
        lib = _lib.load()
        if njobs > _lib.SHADOW_MAX_JOBS:          # more jobs than one argument block holds: the refreshes keep their own launches
            for h in handles:
                _lib.check(lib.vln_shadow_refresh(h[0], h[1], _lib.raw_stream()), "vln_shadow_refresh")
            handles, njobs = [], 0
        jobs = None
        if njobs:
            jobs = (_lib.ShadowJob * njobs)(*[h[0][i] for h in handles for i in range(h[1])])
        f = feed.fetch_args() if feed is not None else (None, 0, None, None, None, 0)
        _lib.check(lib.vln_prologue(*f, items, n, jobs, njobs, _lib.raw_stream()), "vln_prologue")

    def uncount(self):
        """Undo the host side of one tick(): a tick that was CAPTURED did not run, the device words are unchanged."""
        self.host -= self.STRIDE
        self.epoch -= 1
        for s in self._seqs:
            s[3] = max(0, s[3] - self.STRIDE)

    def replayed(self):
        """A captured graph that contains this clock's tick launch is about to be replayed: advance the host side only."""
        self._maintain()


class ShadowSet:
    def __init__(self):
        self._key = None
        self.t: Dict[str, torch.Tensor] = {}

    @staticmethod
    def key_of(params: Sequence[torch.Tensor], dtype) -> tuple:
        return (dtype,) + tuple((p._version, p.data_ptr()) for p in params)

    def stale(self, key) -> bool:
        return key != self._key

    def commit(self, key):
        self._key = key


class StepSlot:
    """Row block [r0, r0+rows) of one chunk, owned by one decoder step."""
    __slots__ = ("chunk", "r0", "rows", "done")

    def __init__(self, chunk, r0, rows):
        self.chunk, self.r0, self.rows, self.done = chunk, r0, rows, False

    def view(self, site: str) -> torch.Tensor:
        return self.chunk.bufs[site][self.r0:self.r0 + self.rows]

    def ptr(self, site: str) -> int:
        """Device address of the row block (no tensor view is built: the decoder asks ~15 times per step)."""
        base, row_bytes = self.chunk.addr[site]
        return base + self.r0 * row_bytes


class Chunk:
    __slots__ = ("bufs", "cap", "used", "slots", "owners", "addr")

    def __init__(self, bufs, cap):
        self.bufs, self.cap = bufs, cap
        self.addr = {k: (t.data_ptr(), t.stride(0) * t.element_size()) for k, t in bufs.items()}
        self.used, self.slots, self.owners = 0, [], []

    def recyclable(self) -> bool:
        # every rollout that wrote here is gone (its context tensor died => its autograd graph died)
        return all(o() is None for o in self.owners) and not any(s.done for s in self.slots)

    def clear(self):
        self.used, self.slots, self.owners = 0, [], []


class Stash:
    """Row arena for the deferred weight-gradient operands of one module (`sites`: name -> width)."""
    CHUNK_ROWS = 1024

    def __init__(self, sites: Dict[str, int], device):
        self.sites = dict(sites)
        self.device = device
        self.chunks: List[Chunk] = []
        self._pool: List[Chunk] = []

    def reset(self):
        for c in self.chunks:
            c.clear()
        self._pool.extend(self.chunks)
        self.chunks = []

    def _new_chunk(self, rows: int) -> Chunk:
        # reclaim chunks of dead rollouts first (bounds memory when graphs are built but never backpropagated,
        # e.g. the reference's validation rollouts)
        live = []
        for c in self.chunks[:-1] if self.chunks else []:
            if c.recyclable():
                c.clear()
                self._pool.append(c)
            else:
                live.append(c)
        if self.chunks:
            live.append(self.chunks[-1])
        self.chunks = live
        for i, c in enumerate(self._pool):
            if c.cap >= rows:
                ch = self._pool.pop(i)
                break
        else:
            cap = max(rows, self.CHUNK_ROWS)
            ch = Chunk({k: torch.empty(cap, w, dtype=torch.float32, device=self.device)
                        for k, w in self.sites.items()}, cap)
        self.chunks.append(ch)
        return ch

    def take(self, rows: int, owner_ref) -> StepSlot:
        ch = self.chunks[-1] if self.chunks else None
        if ch is None or ch.used + rows > ch.cap:
            ch = self._new_chunk(rows)
        slot = StepSlot(ch, ch.used, rows)
        ch.used += rows
        ch.slots.append(slot)
        if not ch.owners or ch.owners[-1] is not owner_ref:
            ch.owners.append(owner_ref)
        return slot

    def untake(self, slot: StepSlot):
        """Give back the most recent take() (a caller that found its cached plan stale re-takes on the full path)."""
        ch = slot.chunk
        if ch.slots and ch.slots[-1] is slot:
            ch.slots.pop()
            ch.used -= slot.rows

    def done_runs(self):
        """Yield (bufs, r0, r1) for maximal runs of steps whose backward ran; consumes the flags."""
        for c in self.chunks:
            run = None
            for st in c.slots:
                if st.done:
                    st.done = False
                    if run is not None and run[1] == st.r0:
                        run[1] = st.r0 + st.rows
                        continue
                    if run is not None:
                        yield c.bufs, run[0], run[1]
                    run = [st.r0, st.r0 + st.rows]
                elif run is not None:
                    yield c.bufs, run[0], run[1]
                    run = None
            if run is not None:
                yield c.bufs, run[0], run[1]


class WeightGate(torch.autograd.Function):
    """out = ONE token tensor that every decoder step takes as an input (the dependency edge is all autograd needs: a
    step Function fed the ten parameter aliases paid ~40 us of argument handling per call); backward =
    owner._deferred_wgrads(), run once every step that consumed the token has run its own backward."""

    @staticmethod
    def forward(ctx, owner_ref, *params):
        ctx.owner_ref = owner_ref
        ctx.n = len(params)
        ctx.set_materialize_grads(False)
        return params[0].detach().view(-1)[:1]

    @staticmethod
    def backward(ctx, *unused):
        owner = ctx.owner_ref()
        if owner is None:
            return (None,) + (None,) * ctx.n
        grads = owner._deferred_wgrads()
        return (None,) + tuple(grads)


class CtxEntry:
    """Per-context bookkeeping (one per rollout): the gated alias, the in-place dctx accumulator and the
    low-precision copy.  Autograd nodes only hold it weakly so no tensor<->node cycle can form."""
    __slots__ = ("ref", "dctx", "lp", "gated", "mask_src", "mask8", "terms", "shape", "kctx", "k_w", "k_split", "k_hd", "tdesc", "__weakref__")

    def __init__(self, t):
        self.ref = weakref.ref(t)
        self.dctx = None
        # deferred context gradient: per decoder step (alpha_t ptr, dl ptr, [dwc|.] ptr, query ptr, keep-alive tensors...)
        self.terms = []
        self.shape = None
        self.lp = None
        self.gated = None
        self.mask_src = self.mask8 = None     # the caller's ctx_mask and its uint8 form (converted once per rollout)
        # projected context K = ctx W_in of the rollout (EnvDropDecoder.project_context): None = not decided yet, False = not used,
        # else the [B,L,H] fp32 tensor; k_w = text_attn.linear_in's streamed shadow [H,H], k_split = multiply it in the split form
        self.kctx = None
        self.k_w = None
        self.k_split = False
        self.k_hd = None        # callable (address, rows) -> [rows, H] strided view of the module's drop(h_1) stash rows
        # terms of another layout than the EnvDrop step's (monitor_step): [ldg, ldq, [per-term (seed, offset, p[, base]) or None]]
        self.tdesc = None


class CtxGate(torch.autograd.Function):
    """Identity on the encoder context; backward returns the dctx buffer the decoder steps accumulated
    into in place (one [B,L,H] buffer per rollout instead of one per step)."""

    @staticmethod
    def forward(ctx, entry, x):
        ctx.entry_ref = weakref.ref(entry)
        ctx.set_materialize_grads(False)
        return x.detach()

    @staticmethod
    def backward(ctx, g):
        e = ctx.entry_ref()
        d = None
        if e is not None:
            d, e.dctx, e.gated = e.dctx, None, None
            if e.terms:
                terms, e.terms = e.terms, []
                B, L, H = e.shape
                acc = d is not None
                kmode = e.kctx is not None and e.kctx is not False
                if d is None and not kmode:
                    d = ops.empty(B, L, H, dtype=torch.float32, device=terms[0][4].device)
                if kmode:
                    # projected context (vln_envdrop_step.kctx): the steps' queries never existed.  dctx = sum_t alpha_t g_t +
                    # (sum_t dl_t hd_t) W_in^T with hd_t = drop(h_1) of step t (t[3]: the tcat stash rows, columns [H, 2H))
                    dev = terms[0][4].device
                    if d is None:
                        d = ops.empty(B, L, H, dtype=torch.float32, device=dev)
                    T = len(terms)
                    row = B * 2 * H * 4
                    x = None
                    if e.k_hd is not None and all(terms[i][3] == terms[0][3] - i * row for i in range(T)):
                        x = e.k_hd(terms[-1][3], T * B)                              # [T * B, H] view (row stride 2H) of the stash
                    if x is not None:
                        # the steps' hd rows are consecutive row blocks of the stash (the backward visits them last step first):
                        # (sum_t dl_t hd_t) W_in^T = sum_t dl_t (W_in hd_t) -- the steps' QUERIES, formed now for all steps by ONE
                        # product over (steps x batch) rows, then the one-launch form of the context gradient (no dK tensor)
                        q = ops.linear_fwd(x, e.k_w, split=e.k_split)                # W_in hd_t, step T-1-i at rows [i * B, (i + 1) * B)
                        qp = [q.data_ptr() + (T - 1 - i) * B * H * 4 for i in range(T)]
                        ops.attn_dctx_deferred([t[0] for t in terms], [t[1] for t in terms], [t[2] for t in terms], 2 * H, qp, H, d,
                                               accumulate=acc)
                    else:
                        # -- ONE pass over the steps' vectors writes both sums, ONE product adds the second through W_in^T
                        dk = ops.empty(B, L, H, dtype=torch.float32, device=dev)
                        ops.attn_dctx_deferred([t[0] for t in terms], [t[1] for t in terms], [t[2] for t in terms], 2 * H,
                                               [t[3] for t in terms], 2 * H, d, accumulate=acc, dk=dk)
                        ops.linear_fwd(dk.view(B * L, H), e.k_w, act=ops.ACT_ACCUM, out=d.view(B * L, H), split=e.k_split)
                elif e.tdesc is not None:
                    ldg, ldq, drops = e.tdesc
                    e.tdesc = None
                    ops.attn_dctx_deferred([t[0] for t in terms], [t[1] for t in terms], [t[2] for t in terms], ldg,
                                           [t[3] for t in terms], ldq, d, accumulate=acc,
                                           drop=[x if x is not None else (0, 0, 0.0) for x in drops] if any(x is not None for x in drops) else None)
                else:
                    ops.attn_dctx_deferred([t[0] for t in terms], [t[1] for t in terms], [t[2] for t in terms], 2 * H,
                                           [t[3] for t in terms], H, d, accumulate=acc)
        if g is not None:
            d = g if d is None else d + g
        return None, d


class GatedModuleMixin:
    """Bookkeeping for modules whose per-step autograd nodes defer weight grads through a WeightGate."""

    def _init_gating(self):
        self._shadow = ShadowSet()
        self._stash: Optional[Stash] = None
        self._gate_outs = None
        self._ctx_entries: Dict[int, CtxEntry] = {}
        self._step_counter = 0
        self.compute_dtype = torch.float32      # dtype of the streamed operands (weights / features / ctx)
        self.dropout_seed = 0x5EED
        self._shadow_ready = None               # event recorded by prepare() (side-stream shadow refresh)
        self._open_versions = None              # parameter versions when the current gate opened

    # -- provided by the module -----------------------------------------------------------------------
    def _gated_params(self) -> List[torch.Tensor]:
        raise NotImplementedError

    def _stash_sites(self) -> Dict[str, int]:
        raise NotImplementedError

    def _refresh_shadows(self):
        raise NotImplementedError

    def _deferred_wgrads(self) -> List[Optional[torch.Tensor]]:
        raise NotImplementedError

    # -------------------------------------------------------------------------------------------------
    def prepare(self):
        """Optional hint, call right after optimizer.step(): refresh the weight shadows NOW on a side stream so the
        casts/transposes overlap whatever runs next on the main stream (e.g. the instruction encoder).  Forward
        joins on an event; without this call the refresh happens lazily in forward -- same results either way."""
        params = self._gated_params()
        key = ShadowSet.key_of(params, self.compute_dtype)
        if not params[0].is_cuda or not self._shadow.stale(key):
            return
        main = torch.cuda.current_stream()
        side = getattr(self, "_prep_stream", None)
        if side is None or side.device != main.device:
            side = self._prep_stream = torch.cuda.Stream(main.device)
        side.wait_stream(main)                      # the optimizer updated the masters on the main stream
        with torch.cuda.stream(side), torch.no_grad():
            self._refresh_shadows()
        self._shadow.commit(key)
        self._gate_outs = None
        self._shadow_ready = torch.cuda.Event()
        self._shadow_ready.record(side)

    def _ensure_current(self, need_grad: bool, params=None):
        ev = self._shadow_ready
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)
            self._shadow_ready = None
        if params is None:
            params = self._gated_params()
        # A gate is open (a rollout is being recorded): only an in-place update could have staled the shadows since it
        # opened, and that bumps `_version` -- ten attribute reads instead of the full (version, address) key per step.
        if need_grad and self._gate_outs is not None and self._open_versions == [p._version for p in params]:
            return self._gate_outs
        key = ShadowSet.key_of(params, self.compute_dtype)
        if self._shadow.stale(key):
            with torch.no_grad():
                self._refresh_shadows()
            self._shadow.commit(key)
            self._gate_outs = None
        if need_grad and self._gate_outs is None:
            dev = params[0].device
            if self._stash is None or self._stash.device != dev:
                self._stash = Stash(self._stash_sites(), dev)
            else:
                self._stash.reset()
            self._ctx_entries = {}
            self._gate_opened()
            self._gate_outs = WeightGate.apply(weakref.ref(self), *params)
            self._open_versions = [p._version for p in params]
            self._rebase_offsets(params[0].device)
        return self._gate_outs

    def prefresh(self):
        """Refresh the weight shadows NOW if a parameter changed since they were built (what the first forward after an optimizer
        step would do): lets the caller issue the refresh where it likes -- e.g. inside the iteration's prologue launch."""
        params = self._gated_params()
        key = ShadowSet.key_of(params, self.compute_dtype)
        if self._shadow.stale(key):
            with torch.no_grad():
                self._refresh_shadows()
            self._shadow.commit(key)
            self._gate_outs = None

    def _gate_opened(self):
        """A new rollout recording begins (hook): whatever an earlier, abandoned one left pending in the library is stale."""

    def _gate_consumed(self):
        # called from _deferred_wgrads once the dW GEMMs are issued: the next forward opens a new gate
        self._gate_outs = None

    def _gated_ctx(self, ctx_t: torch.Tensor, need_grad: bool):
        """-> (tensor to hand to the step Function, CtxEntry)."""
        k = id(ctx_t)
        e = self._ctx_entries.get(k)
        if e is None or e.ref() is not ctx_t:
            if len(self._ctx_entries) > 16:
                self._ctx_entries = {i: x for i, x in self._ctx_entries.items() if x.ref() is not None}
            e = CtxEntry(ctx_t)
            self._ctx_entries[k] = e
        if need_grad and ctx_t.requires_grad:
            if e.gated is None:
                e.gated = CtxGate.apply(e, ctx_t)
            return e.gated, e
        return ctx_t, e

    def _next_offset(self) -> int:
        clock = self.__dict__.get("clock")
        if clock is not None:        # DeviceClock: call r since the tick -> the offset a host counter at clock.host would give
            self._step_counter = clock.value(clock.rel(id(self)))
            return self._step_counter
        self._step_counter += 1
        return self._step_counter

    def _rebase_offsets(self, device):
        """Dropout offsets of the rollouts recorded under the gate that just opened are counted from a base value held
        in a device word (one of two, alternating per gate: the previous gate's backward may still read the other one).
        Written once here; the steps then carry only their small relative offset (see vln_envdrop_step.offset_base_dev)."""
        clock = self.__dict__.get("clock")
        if clock is not None:        # the clock's word IS the base: nothing to write, the tick launch bumps it
            self._base_value, self._base_ptr = clock.host, clock.ptr
            return
        b = self.__dict__.get("_offset_bases")
        if b is None or b.device != device:
            b = self._offset_bases = torch.zeros(2, dtype=torch.int64, device=device)
            self._base_epoch = 0
        self._base_epoch += 1
        slot = self._base_epoch & 1
        self._base_value = self._step_counter
        b[slot].fill_(self._base_value)
        self._base_ptr = b.data_ptr() + 8 * slot

    @staticmethod
    def _ctx_lp(entry: CtxEntry, ctx_t: torch.Tensor, dtype):
        if dtype == torch.float32:
            return None
        if entry.lp is None:
            lp = getattr(ctx_t, "_vln_lp", None)      # EncoderLSTM (bf16 mode) already wrote the copy
            if lp is not None and lp.dtype == dtype and lp.shape == ctx_t.shape and lp.device == ctx_t.device:
                entry.lp = lp
                return lp
            src = ctx_t.detach().contiguous()
            entry.lp = ops.cast_copy(src, dtype, ops.empty(src.shape, dtype=dtype, device=src.device))
        return entry.lp
