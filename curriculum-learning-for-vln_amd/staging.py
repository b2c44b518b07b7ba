"""Getting per-step inputs into HBM (reference: agent/base.py:114-178 marshalling, `.to(device)` from pageable
host memory every decoder step).

* `DeviceFeatureStore` -- the MI355X-first layout: the precomputed ResNet table stays resident in HBM (fp32 2.9 GB
  or bf16 1.5 GB for R2R's 10,567 viewpoints; 288 GB per GPU) and each step sends indices only; one HIP pass
  gathers rows, appends angle features, applies the feature dropout and emits the bf16 stream copy
  (`vln_gather_pano` / `vln_gather_cands`).
* `PinnedStager` -- for callers that keep features on the host: a ring of pinned buffers and a dedicated copy
  stream so `hipMemcpyAsync` of the next step's features overlaps compute; tensors become valid on the compute
  stream through an event, the host never blocks on the copy.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence

import numpy as np
import ctypes as C_

import torch

from . import _lib, ops

_p = ops._p


def loc_embedding_table(angle_size: int = 128, views: int = 36) -> torch.Tensor:
    """[viewIndex, absView, angle] static panorama angle features (utils/misc.py:296-317): 12 headings x 3
    elevations 30 degrees apart, heading relative to the current view."""
    inc = math.pi / 6.0
    n = angle_size // 4
    t = np.zeros((views, views, angle_size), np.float32)
    for vi in range(views):
        for av in range(views):
            rel = (av - vi) % 12 + (av // 12) * 12
            h, e = (rel % 12) * inc, (rel // 12 - 1) * inc
            t[vi, av] = np.repeat(np.array([math.sin(h), math.cos(h), math.sin(e), math.cos(e)], np.float32), n)
    return torch.from_numpy(t)


def read_feature_tsv(path: str, views: int = 36, image_w: int = 640, image_h: int = 480, vfov: int = 60):
    """The precomputed-feature file of the reference (`ImageFeatures.read_in`, utils/misc.py:253-279): one TSV row per
    viewpoint, fields scanId, viewpointId, image_w, image_h, vfov, features = base64 of float32 [views, 2048].
    Returns (table float32 [N, views, IMG] in file order, ids) with ids[i] = scanId + "_" + viewpointId -- the layout
    `DeviceFeatureStore` keeps in HBM instead of the reference's dict of per-viewpoint arrays."""
    import base64
    import csv
    import sys
    csv.field_size_limit(sys.maxsize)
    rows, ids = [], []
    with open(path, "r") as f:
        for item in csv.DictReader(f, delimiter="\t", fieldnames=["scanId", "viewpointId", "image_w", "image_h", "vfov", "features"]):
            if int(item["image_h"]) != image_h or int(item["image_w"]) != image_w or int(item["vfov"]) != vfov:
                raise ValueError(f"{path}: unexpected camera parameters in row {len(ids)}")
            a = np.frombuffer(base64.b64decode(item["features"].encode("ascii")), dtype=np.float32)
            if a.size % views:
                raise ValueError(f"{path}: row {len(ids)} does not hold {views} views")
            rows.append(a.reshape(views, -1))
            ids.append(item["scanId"] + "_" + item["viewpointId"])
    if not rows:
        raise ValueError(f"{path}: no feature rows")
    return torch.from_numpy(np.stack(rows)), ids


class DeviceFeatureStore:
    def __init__(self, table: torch.Tensor, ids: Optional[Sequence[str]] = None, device="cuda", dtype=torch.float32,
                 angle_size: int = 128, seed: int = 0xFEA7):
        """table [N, V, IMG] (host or device); ids[i] = the reference's long_id (scan_viewpoint) of row i."""
        assert table.dim() == 3
        self.device = torch.device(device)
        self.table = table.to(self.device, dtype).contiguous()
        self.N, self.V, self.IMG = self.table.shape
        self.ANG = angle_size
        self.row_of: Dict[str, int] = {k: i for i, k in enumerate(ids)} if ids is not None else {}
        self.angle_table = loc_embedding_table(angle_size, self.V).to(self.device)
        self.seed, self._calls = seed, 0
        self._side_event = None
        # every gather of this table is range-checked on the device from here on (out-of-range index -> zero row + a sticky
        # error that the next vln_persistent_check raises); the angle table holds one row block per view index
        if self.table.is_cuda:
            _lib.check(_lib.load().vln_feature_table_extent(self.table.data_ptr(), self.N, self.angle_table.shape[0]),
                       "vln_feature_table_extent")

    def __del__(self):
        try:
            if self.table.is_cuda:
                _lib.load().vln_feature_table_extent(self.table.data_ptr(), 0, 0)       # the address may be reused by another tensor
        except Exception:
            pass

    def validate_indices(self, rows=None, view_index=None, cand_rows=None, cand_views=None):
        """Host-side check of a batch's (or a whole tape's) index tensors when it is REGISTERED -- one synchronising reduction
        per tensor, not something to call per step (the device kernels check every index they use anyway and zero the row):
        rows in [0, N), view_index in [0, angle views), cand_rows < N (negative = empty slot), cand_views in [0, V) where the
        slot is not empty.  Raises ValueError naming the first offending tensor."""
        def bad(name, t, lo, hi, where=None):
            if t is None:
                return
            t = torch.as_tensor(t)
            ok = (t >= lo) & (t < hi)
            if where is not None:
                ok = ok | ~where
            if not bool(ok.all()):
                i = int((~ok).reshape(-1).nonzero()[0])
                raise ValueError(f"DeviceFeatureStore: {name}[{i}] = {int(t.reshape(-1)[i])} is outside [{lo}, {hi})")
        bad("rows", rows, 0, self.N)
        bad("view_index", view_index, 0, int(self.angle_table.shape[0]))
        if cand_rows is not None:
            cr = torch.as_tensor(cand_rows)
            bad("cand_rows", cr, -(1 << 62), self.N)
            bad("cand_views", cand_views, 0, self.V, where=(cr >= 0) if cand_views is not None else None)

    @classmethod
    def from_tsv(cls, path: str, device="cuda", dtype=torch.float32, views: int = 36, **kw):
        """Load the reference's feature TSV once and keep it resident (utils/misc.py:253-279)."""
        table, ids = read_feature_tsv(path, views)
        return cls(table, ids, device=device, dtype=dtype, **kw)

    def rows_for(self, long_ids: Sequence[str]) -> torch.Tensor:
        """int64 row indices (on the store's device) of `scanId_viewpointId` keys -- what a step ships instead of features."""
        return torch.tensor([self.row_of[k] for k in long_ids], dtype=torch.int64, device=self.device)

    def _drop(self, p):
        self._calls += 1
        return self.seed, self._calls, float(p)

    def gather_pano(self, rows: torch.Tensor, view_index: torch.Tensor, p_feat: float = 0.0, want_bf16: bool = False,
                    want_f32: bool = True):
        """rows int64 [B], view_index int32 [B] (device) -> img_feature [B, V, IMG+ANG] (+ bf16 copy).
        want_f32=False (with want_bf16): only the bf16 rows are written; a bf16 EnvDropDecoder accepts them as its
        `img_feature` (it streams nothing else), which halves this pass's HBM writes."""
        lib = _lib.load()
        B = rows.shape[0]
        F = self.IMG + self.ANG
        out = ops.empty(B, self.V, F, dtype=torch.float32, device=self.device) if (want_f32 or not want_bf16) else None
        lp = ops.empty(B, self.V, F, dtype=torch.bfloat16, device=self.device) if want_bf16 else None
        seed, off, p = self._drop(p_feat)
        _lib.check(lib.vln_gather_pano(_p(self.table), ops._dt(self.table), _p(rows), _p(view_index), _p(self.angle_table),
                                       _p(out), _p(lp), B, self.V, self.IMG, self.ANG, seed, off, p,
                                       _lib.raw_stream()), "vln_gather_pano")
        return (out, lp, (seed, off)) if want_bf16 else (out, (seed, off))

    def gather_cands(self, rows: torch.Tensor, views: torch.Tensor, heading: torch.Tensor, elevation: torch.Tensor,
                     p_feat: float = 0.0, want_bf16: bool = False, want_f32: bool = True):
        """rows int64 [B,C] (-1 = STOP slot / padding), views int32 [B,C], heading/elevation fp32 [B,C]."""
        lib = _lib.load()
        B, C = rows.shape
        F = self.IMG + self.ANG
        out = ops.empty(B, C, F, dtype=torch.float32, device=self.device) if (want_f32 or not want_bf16) else None
        lp = ops.empty(B, C, F, dtype=torch.bfloat16, device=self.device) if want_bf16 else None
        seed, off, p = self._drop(p_feat)
        _lib.check(lib.vln_gather_cands(_p(self.table), ops._dt(self.table), _p(rows.contiguous()), _p(views.contiguous()),
                                        _p(heading.contiguous()), _p(elevation.contiguous()), _p(out), _p(lp), B * C,
                                        self.V, self.IMG, self.ANG, seed, off, p,
                                        _lib.raw_stream()), "vln_gather_cands")
        return (out, lp, (seed, off)) if want_bf16 else (out, (seed, off))


    def gather_step(self, rows, view_index, crows, cviews, heading, elevation, p_feat: float = 0.0,
                    want_bf16: bool = False, want_f32: bool = True, stream: Optional["torch.cuda.Stream"] = None):
        """gather_pano + gather_cands of one decoder step as ONE launch -> ((img, img_bf16), (cand, cand_bf16),
        ((seed, off_pano), (seed, off_cand))); entries not asked for are None.
        `stream`: issue the gather on this side stream.  It reads only the resident table and index vectors -- nothing a
        decoder step produced (with teacher forcing not even the path depends on the logits) -- so it runs beside the
        encoder / the previous step's kernels instead of in line with them; the current stream is made to wait for it.
        With a RolloutArena the caller fences the side stream once per iteration (`stream.wait_stream(current)` before the
        first gather: the buffers it writes were last read two iterations earlier); without one the fence is taken here,
        per call, because the caching allocator may hand out memory whose last use is still in flight."""
        lib = _lib.load()
        raw = _lib.raw_stream()
        if stream is not None:
            if ops.current_arena() is None:
                stream.wait_stream(torch.cuda.current_stream(self.device))
            raw = stream.cuda_stream
        B, C = crows.shape
        F = self.IMG + self.ANG
        f32 = want_f32 or not want_bf16
        dev = self.device
        img = ops.empty(B, self.V, F, dtype=torch.float32, device=dev) if f32 else None
        cand = ops.empty(B, C, F, dtype=torch.float32, device=dev) if f32 else None
        img_lp = ops.empty(B, self.V, F, dtype=torch.bfloat16, device=dev) if want_bf16 else None
        cand_lp = ops.empty(B, C, F, dtype=torch.bfloat16, device=dev) if want_bf16 else None
        seed, off1, p = self._drop(p_feat)
        _, off2, _ = self._drop(p_feat)
        _lib.check(lib.vln_gather_step(_p(self.table), ops._dt(self.table), _p(self.angle_table), _p(rows), _p(view_index),
                                       _p(crows.contiguous()), _p(cviews.contiguous()), _p(heading.contiguous()),
                                       _p(elevation.contiguous()), _p(img), _p(img_lp), _p(cand), _p(cand_lp), B, self.V, C,
                                       self.IMG, self.ANG, seed, off1, off2, p, raw), "vln_gather_step")
        if stream is not None:
            ev = self._side_event
            if ev is None:
                ev = self._side_event = torch.cuda.Event()
            ev.record(stream)
            torch.cuda.current_stream(self.device).wait_event(ev)
        return (img, img_lp), (cand, cand_lp), ((seed, off1), (seed, off2))


    def rollout_ride(self, steps, p_feat: float = 0.0, want_bf16: bool = False, want_f32: bool = True, out=None):
        """`gather_rollout` as a DESCRIPTION instead of a launch: hand it to `EncoderLSTM.forward(..., ride=...)` and the gather
        runs as passenger workgroups of the encoder's persistent recurrence launch (on the compute units that launch leaves
        idle); `ride.outputs` is what gather_rollout would have returned, valid once the encoder's forward has been issued."""
        return self.gather_rollout(steps, p_feat, want_bf16, want_f32, out, _ride=True)

    def gather_rollout(self, steps, p_feat: float = 0.0, want_bf16: bool = False, want_f32: bool = True, out=None, _ride=False):
        """`gather_step` for EVERY step of a teacher-forced rollout in ONE launch (the path is known when the rollout starts:
        base.py:141-157 driven by the ground-truth actions).  steps: sequence of (rows, view_index, crows, cviews, heading,
        elevation); returns a list of ((img, img_bf16), (cand, cand_bf16)) like gather_step.  Same Philox stream as calling
        gather_step for the steps in order (two offsets per step), so the results are bit-identical.
        `out`: a previous result of this call with the same shapes -- the rows are written into those buffers again
        (address-stable destinations for captured graphs)."""
        lib = _lib.load()
        dev = self.device
        F = self.IMG + self.ANG
        f32 = want_f32 or not want_bf16
        arr = (_lib.GatherRolloutStep * len(steps))()
        res, keep = [], []
        seed, p = 0, 0.0
        clock = self.__dict__.get("clock")        # runtime.DeviceClock (whole-iteration graphs)
        for t, (rows, view_index, crows, cviews, heading, elevation) in enumerate(steps):
            B, C = crows.shape
            if (rows.dtype != torch.int64 or crows.dtype != torch.int64 or view_index.dtype != torch.int32 or cviews.dtype != torch.int32
                    or heading.dtype != torch.float32 or elevation.dtype != torch.float32):
                raise TypeError("gather_rollout: rows / cand rows int64, view_index / cand views int32, heading / elevation float32")
            if rows.shape != (B,) or view_index.shape != (B,) or cviews.shape != (B, C) or heading.shape != (B, C) or elevation.shape != (B, C):
                raise ValueError("gather_rollout: step %d: index tensors do not share one [B] / [B, C] layout" % t)
            if out is not None:
                (img, img_lp), (cand, cand_lp) = out[t]
                ok = lambda x, n, dt: (x is None) == (dt is None) and (x is None or (tuple(x.shape) == (B, n, F) and x.dtype == dt))
                if not (ok(img, self.V, torch.float32 if f32 else None) and ok(cand, C, torch.float32 if f32 else None) and
                        ok(img_lp, self.V, torch.bfloat16 if want_bf16 else None) and ok(cand_lp, C, torch.bfloat16 if want_bf16 else None)):
                    raise ValueError("gather_rollout: `out` buffers do not match this rollout's shapes / dtypes")
            else:
                img = ops.empty(B, self.V, F, dtype=torch.float32, device=dev) if f32 else None
                cand = ops.empty(B, C, F, dtype=torch.float32, device=dev) if f32 else None
                img_lp = ops.empty(B, self.V, F, dtype=torch.bfloat16, device=dev) if want_bf16 else None
                cand_lp = ops.empty(B, C, F, dtype=torch.bfloat16, device=dev) if want_bf16 else None
            if clock is not None:      # offsets relative to the clock's device word: (word * 8 + r), r < 8 * STRIDE
                seed, p = self.seed, float(p_feat)
                off1 = clock.rel(id(self), 8 * clock.STRIDE - 1)
                off2 = clock.rel(id(self), 8 * clock.STRIDE - 1)
            else:
                seed, off1, p = self._drop(p_feat)
                _, off2, _ = self._drop(p_feat)
            rw_, vi_ = rows.contiguous(), view_index.contiguous()
            cr, cv, hd, el = crows.contiguous(), cviews.contiguous(), heading.contiguous(), elevation.contiguous()
            keep += [rw_, vi_, cr, cv, hd, el]
            q = arr[t]
            q.rows, q.view_index, q.crows, q.cviews, q.heading, q.elevation = _p(rw_), _p(vi_), _p(cr), _p(cv), _p(hd), _p(el)
            q.out, q.out_bf16, q.cout, q.cout_bf16 = _p(img), _p(img_lp), _p(cand), _p(cand_lp)
            q.offset_pano, q.offset_cand = off1, off2
            res.append(((img, img_lp), (cand, cand_lp)))
            if t and (B, C) != tuple(steps[0][2].shape):
                raise ValueError("gather_rollout: every step must have the same [B, C] candidate layout")
        B, C = steps[0][2].shape
        if _ride:
            r = _lib.GatherRide()
            r.table, r.angle_table, r.steps = _p(self.table), _p(self.angle_table), C_.addressof(arr)
            r.ttype, r.T, r.B, r.V, r.C, r.IMG, r.ANG = ops._dt(self.table), len(steps), B, self.V, C, self.IMG, self.ANG
            r.seed, r.p_feat, r.offset_base_dev = seed, p, None if clock is None else clock.ptr
            return RolloutRide(r, res, (arr, keep, [s_[0] for s_ in steps], [s_[1] for s_ in steps]))
        _lib.check(lib.vln_gather_rollout(_p(self.table), ops._dt(self.table), _p(self.angle_table), arr, len(steps), B, self.V, C,
                                          self.IMG, self.ANG, seed, p, None if clock is None else clock.ptr, _lib.raw_stream()),
                   "vln_gather_rollout")
        return res


class RolloutRide:
    """A rollout's feature gather waiting for a carrier launch (DeviceFeatureStore.rollout_ride)."""
    __slots__ = ("struct", "outputs", "_keep")

    def __init__(self, struct, outputs, keep):
        self.struct, self.outputs, self._keep = struct, outputs, keep

    def carry_batch_tail(self, feed: "HostBatchFeed"):
        """The carrier launch also pulls the TAIL of the current batch blob (`feed.split_at(offset)`: what only the decoder reads) out
        of pinned host memory with one passenger workgroup -- PCIe traffic under the recurrence instead of in front of it."""
        t = feed.tail_args()
        if t is not None:
            r = self.struct
            r.fetch_slots, r.fetch_ring, r.fetch_seq, r.fetch_dst, r.fetch_offset, r.fetch_bytes = t
            self._keep = (self._keep, feed)
        return self

    def carry_shadows(self, modules):
        """The carrier launch also refreshes the weight shadows of `modules` (whatever the last optimizer step staled, as
        `DeviceClock.prologue(modules=...)` would): modules the carrier does NOT read itself -- the decoder, first used after the
        instruction encoder.  The passengers do it once their rows are gathered, so those bytes leave the iteration's dependent
        chain (the prologue launch was bound by them).  The modules' own forward then finds its shadows current."""
        with ops.ShadowBatch.collect() as handles:
            for m in modules:
                m.prefresh()
        return self.carry_shadow_jobs([h[0][i] for h in handles for i in range(h[1])], handles)

    def carry_shadow_jobs(self, jobs, keep=None):
        """`carry_shadows` for explicit jobs (a sequence of _lib.ShadowJob, e.g. ops.ShadowBatch().jobs): same results as
        vln_shadow_refresh on them."""
        n = len(jobs)
        if n:
            arr = (_lib.ShadowJob * n)(*jobs)
            self.struct.shadow_jobs, self.struct.n_shadow_jobs = C_.addressof(arr), n
            self._keep = (self._keep, arr, keep)
        return self


class PinnedStager:
    """Ring of pinned host buffers + a copy stream.  `put(name->array)` returns device tensors that are ordered
    after the async copy on the CURRENT stream; the slot is recycled `depth` calls later."""

    def __init__(self, device="cuda", depth: int = 3):
        self.device = torch.device(device)
        self.depth = depth
        self.copy_stream = torch.cuda.Stream(self.device)
        self.slots = [dict(host={}, dev={}, free=None) for _ in range(depth)]
        self.i = 0

    @staticmethod
    def _fits(t, shape, dtype):
        return t is not None and t.dtype == dtype and t.numel() >= int(np.prod(shape))

    def put(self, arrays: Dict[str, "np.ndarray | torch.Tensor"]) -> Dict[str, torch.Tensor]:
        slot = self.slots[self.i]
        self.i = (self.i + 1) % self.depth
        cur = torch.cuda.current_stream(self.device)
        if slot["free"] is not None:
            if slot.get("released"):
                slot["free"].synchronize()      # the consumer of this slot (depth calls ago) has finished
            else:
                cur.synchronize()               # consumer never called release(): be conservative
        slot["released"] = False
        self._last = slot
        out = {}
        with torch.cuda.stream(self.copy_stream):
            for k, a in arrays.items():
                t = torch.from_numpy(a) if isinstance(a, np.ndarray) else a
                n = t.numel()
                h = slot["host"].get(k)
                if not self._fits(h, t.shape, t.dtype):
                    h = torch.empty(max(n, 1), dtype=t.dtype).pin_memory()
                    slot["host"][k] = h
                    slot["dev"][k] = torch.empty(max(n, 1), dtype=t.dtype, device=self.device)
                hv = h[:n].view(t.shape)
                hv.copy_(t)                      # host memcpy into pinned memory
                dv = slot["dev"][k][:n].view(t.shape)
                dv.copy_(hv, non_blocking=True)  # hipMemcpyAsync on the copy stream
                out[k] = dv
            ready = torch.cuda.Event()
            ready.record(self.copy_stream)
        cur.wait_event(ready)
        slot["free"] = torch.cuda.Event()
        return out

    def release(self):
        """Mark the tensors of the last `put` as consumed at this point of the current stream."""
        slot = getattr(self, "_last", None)
        if slot is not None and slot["free"] is not None:
            slot["free"].record(torch.cuda.current_stream(self.device))
            slot["released"] = True


class HostBatchFeed:
    """The per-batch hand-over with the GPU pulling (round 4): packed batches wait in pinned host memory, the live device buffer is
    filled by a KERNEL that reads the batch's address from a ring of pinned slot words (`vln_host_fetch`).  `select(blob)` is one
    host store into the next slot; `fetch()` launches the pull -- as the first launch of a captured iteration its arguments repeat,
    so a whole-iteration hipGraph replays on a new batch without any copy call between replays (reference: agent/base.py:114-178
    marshals on the host, `.to(device)` copies).  Exactly one `select` per executed `fetch`, in order.  The host may run ahead of
    the device by `ring` - 1 batches: `launched()` (call it after the launch / replay that contains the fetch) records an event,
    and `select` waits for the event of the iteration that last used the slot it is about to overwrite.  A registered blob must
    stay untouched until the fetch that reads it has run.
    The protocol is CHECKED on the host (round 5): the device picks its slot by a count of the fetches that have run, the host
    writes the slot of its count of selects, and nothing else ties the two -- so a fetch issued without a select before it (an
    eager warm-up iteration of a capture, a retry after an exception), or a second select before the first one's fetch was issued,
    raises here instead of silently pulling a stale or empty slot from then on.  `resync()` realigns the two counts.
    prefetch=True (round 5): `select(blob)` also SENDS the blob ahead -- one asynchronous H2D copy on a copy stream into a ring of
    device-resident staging slots, ordered before whatever the current stream launches next -- and the slot word holds the staging
    slot's address: the fetch kernel then moves the batch HBM -> HBM.  The host runs ahead of the device, so the copy crosses PCIe
    under the PREVIOUS iteration's compute instead of inside the iteration's first launch (a GPU pulling 118 KB over PCIe is bound
    by its outstanding read requests: 27 us of the dependent chain at B = 64, profiles/round5_notes.md)."""

    def __init__(self, live: torch.Tensor, ring: int = 16, prefetch: bool = False):
        if not live.is_cuda or live.dtype != torch.uint8 or live.numel() % 16 or live.data_ptr() % 16:
            raise ValueError("HostBatchFeed: the live buffer is a 16-byte aligned uint8 device tensor whose size is a multiple of 16")
        self.live, self.ring = live, int(ring)
        self.slots = torch.zeros(self.ring, dtype=torch.int64).pin_memory()     # device addresses of the next `ring` blobs
        self._slots_dev = self._devptr(self.slots)
        self._state = torch.zeros(4, dtype=torch.int64, device=live.device)     # [0] = seq (fetches run), [1] = done scratch
        self._addr = {}
        self._selected = 0                                                      # host mirror: selects made
        self._issued = 0                                                        # fetch executions issued (eager launches + replays)
        self._events = [None] * self.ring
        self.head = live.numel()        # bytes the fetch itself pulls; the rest (split_at) rides in a later launch of the iteration
        self.prefetch = bool(prefetch)
        if self.prefetch:
            self._stage = torch.empty(self.ring, live.numel(), dtype=torch.uint8, device=live.device)
            self._copy_stream = torch.cuda.Stream(live.device)
            self._sent = [torch.cuda.Event() for _ in range(self.ring)]
            self._ahead = None                  # (select index, blob address) send_ahead() has already sent

    def split_at(self, offset: int):
        """The fetch pulls bytes [0, offset) only; [offset, end) -- the part nothing reads before the decoder -- is pulled by whoever
        takes `tail_args()` later in the SAME iteration (RolloutRide.carry_batch_tail: a passenger workgroup of the encoder's recurrence
        launch).  offset: a multiple of 16; 0 or the blob size = no split."""
        offset = int(offset)
        if offset % 16 or not (0 <= offset <= self.live.numel()):
            raise ValueError("HostBatchFeed.split_at: a multiple of 16 bytes inside the blob")
        self.head = offset if 0 < offset < self.live.numel() else self.live.numel()
        return self

    def tail_args(self):
        """(slots, ring, seq word, destination, offset, bytes) of the part the fetch leaves behind, or None."""
        n = self.live.numel()
        if self.head >= n:
            return None
        return (self._slots_dev, self.ring, self._state.data_ptr(), self.live.data_ptr() + self.head, self.head, n - self.head)

    @staticmethod
    def _devptr(t: torch.Tensor) -> int:
        d = C_.c_void_p()
        _lib.check(_lib.load().vln_host_device_pointer(t.data_ptr(), C_.byref(d)), "vln_host_device_pointer")
        return int(d.value)

    def register(self, blob: torch.Tensor) -> torch.Tensor:
        """-> the blob as a pinned host tensor the device can read (its device address is looked up once)."""
        if blob.numel() * blob.element_size() != self.live.numel():
            raise ValueError("HostBatchFeed: blob and live buffer differ in size")
        b = blob.detach().cpu().contiguous()
        if not b.is_pinned():
            b = b.pin_memory()
        self._addr[b.data_ptr()] = self._devptr(b)
        return b

    def select(self, blob: torch.Tensor):
        """The next `fetch()` that runs -- eager or replayed -- pulls this (registered) blob."""
        if self._selected != self._issued:
            raise _lib.VlnError("HostBatchFeed.select: the fetch of the previous select has not been issued (one select per executed "
                                "fetch; after a failed iteration call resync())")
        i = self._selected % self.ring
        ev = self._events[i]
        if ev is not None:                      # the iteration that read this slot `ring` selects ago must have run
            ev.synchronize()
            self._events[i] = None
        if self.prefetch:
            # the blob travels on the copy stream -- now, unless send_ahead() already sent it to this slot -- and what the current
            # stream launches next (the iteration with the fetch) waits for it
            if blob.data_ptr() not in self._addr:
                raise KeyError("HostBatchFeed.select: the blob was not registered")
            if self._ahead != (self._selected, blob.data_ptr()):
                self._send(i, blob)
            self._ahead = None
            torch.cuda.current_stream(self.live.device).wait_event(self._sent[i])
            self.slots[i] = self._stage[i].data_ptr()
        else:
            self.slots[i] = self._addr[blob.data_ptr()]
        self._selected += 1

    def _send(self, i, blob):
        with torch.cuda.stream(self._copy_stream):
            self._stage[i].copy_(blob.view(torch.uint8).view(-1), non_blocking=True)
            self._sent[i].record(self._copy_stream)

    def send_ahead(self, blob: torch.Tensor):
        """prefetch only: start the copy of the blob the NEXT select() will pick (call it right after a select: the copy then has
        a whole iteration to cross PCIe in).  A select() of a different blob simply sends that one."""
        if not self.prefetch:
            return
        i = self._selected % self.ring
        ev = self._events[i]
        if ev is not None:                      # the iteration that read this staging slot `ring` selects ago must have run
            ev.synchronize()
            self._events[i] = None
        self._send(i, blob)
        self._ahead = (self._selected, blob.data_ptr())

    def launched(self):
        """Call after issuing the launch / graph replay that contains the fetch of the last `select`."""
        if self._issued == self._selected - 1:          # a replayed graph ran the fetch (an eager fetch counted itself)
            self._issued = self._selected
        elif self._issued != self._selected or self._selected == 0:
            raise _lib.VlnError("HostBatchFeed.launched: a fetch ran without a select before it -- it pulled a stale slot (resync())")
        i = (self._selected - 1) % self.ring
        ev = torch.cuda.Event()
        ev.record()
        self._events[i] = ev

    def _count(self):
        """An eager fetch is being issued (a captured one runs at replay time: `launched()` counts it then)."""
        if torch.cuda.is_current_stream_capturing():
            return
        if self._selected != self._issued + 1:
            raise _lib.VlnError("HostBatchFeed: a fetch is being issued without a select before it (it would pull a stale slot and "
                                "every later fetch the wrong batch): select() the batch first")
        self._issued += 1

    def resync(self):
        """After an iteration that raised: forget selects whose fetch never ran and point the device's count at the host's."""
        torch.cuda.current_stream(self.live.device).synchronize()
        self._selected = self._issued
        self._state[0] = self._issued
        self._events = [None] * self.ring
        if self.prefetch:
            self._ahead = None

    def fetch_args(self):
        """The pull as arguments of `vln_prologue` (runtime.DeviceClock.prologue issues it together with the tick and the refreshes)."""
        self._count()
        st = self._state
        return (self._slots_dev, self.ring, st.data_ptr(), st.data_ptr() + 8, self.live.data_ptr(), self.head)

    def fetch(self):
        self._count()
        st = self._state
        _lib.check(_lib.load().vln_host_fetch(self._slots_dev, self.ring, st.data_ptr(), st.data_ptr() + 8, self.live.data_ptr(),
                                              self.head, _lib.raw_stream()), "vln_host_fetch")
