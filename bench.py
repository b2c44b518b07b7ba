#!/usr/bin/env python3
"""bench.py -- EnvDrop agent training steps/sec on MI355X (BASELINE.json metric, config 1).

One "step" = one agent training iteration of the reference's EnvDrop IL path
(trainer.py:411-427 with feedback="teacher"): instruction encoder forward, T teacher-forced decoder steps
with the in-place candidate mask + cross-entropy (envdrop.py:151-179), `ml_loss * ML_WEIGHT / B`, full
backward, gradient all-reduce (N > 1), clip-norm 40 on encoder and decoder, RMSprop step.  Batch 64 episodes
per GPU, 36 x (2048+128) view features, <= 80 instruction tokens, synthetic data, dropout ON.
Inputs are resident in HBM before the timed region: the FULL-size ResNet table (10,567 viewpoints x 36 x 2048, 1.56 GB in
bf16) and 8 different episode batches (tokens, viewpoint / candidate indices, targets) that the timed loop rotates through,
so every iteration gathers rows it has not touched for 8 iterations from a table six times the Infinity Cache.

    python bench.py                       # N=1, prints ONE JSON line
    python bench.py --gpus N              # starts N ranks itself (torch.distributed.run, RCCL, 127.0.0.1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

`roofline`: the K timed steps replayed with per-kernel HIP-event timers (vln_prof_*); the kernel with the largest total time;
`traffic` / `mfma_util` from the committed rocprofv3 --pmc passes IF they were taken on these kernel sources (hash-checked).
`cpu_baseline`: the CPU oracle (oracle/torch_port.py, kind "port") on tape 0 of the same workload, on rank 0 at N=1 only:
all usable cores (2 warm-ups, median of 10) and 1 thread.  `secondary` (N=1): ms per iteration of the fp32 path, of the
PCIe-inclusive path (pinned fp32 host features), of EnvDrop IL + A2C at the reference's episode cap 35 and of the Self-Monitor
agent at B=128 -- driver-timed side numbers, never `value`.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import torch
import torch.nn.functional as Fn

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
ML_WEIGHT = 0.2            # configs/envdrop/envdrop_config.yaml:45
CLIP = 40.0                # trainer.py:425-426
LR = 1e-4                  # envdrop_config.yaml:19


N_VIEWPOINTS = 10567      # panoramas in the R2R ResNet-152 feature TSV (ImageFeatures.read_in, utils/misc.py:253-279)
N_TAPES = 8               # distinct episode batches rotated through the timed loop


def make_tape(B, L, T, C_max, seed, vocab=992, V=36, IMG=2048, ANG=128, n_rows=None):
    """Synthetic episode batch (BASELINE.md §3 / SURVEY.md §8d), CPU tensors.  Features are defined the way the
    reference's environment builds them (common_env.py:272,287-291,307-308): a per-viewpoint ResNet table
    [viewpoints, 36 views, 2048] (post-ReLU, non-negative), the agent's viewIndex selecting the static angle
    table, and each candidate = (view of the current panorama, its relative heading/elevation).
    n_rows=None: the tape brings its own compact table (T*B viewpoints) and the explicit img/cand tensors of every step
    (tensor path, CPU baseline, tests).  n_rows=N: INDEX-ONLY tape over a resident table of N viewpoints (the bench's
    DeviceFeatureStore): every step visits B random viewpoints, one row of every step has all C_max candidate slots in use
    so the padded candidate width is the same for every tape."""
    g = torch.Generator().manual_seed(seed)
    lens = torch.sort(torch.randint(8, L + 1, (B,), generator=g), descending=True).values
    lens[0] = L
    tokens = torch.zeros(B, L, dtype=torch.long)
    for i, n in enumerate(lens.tolist()):
        tokens[i, 0] = 3                                        # <BOS>
        tokens[i, 1:n - 1] = torch.randint(4, vocab, (n - 2,), generator=g)
        tokens[i, n - 1] = 2                                    # <EOS>
    seq_mask = tokens == 0
    table = None if n_rows is not None else torch.randn(T * B, V, IMG, generator=g).abs() * 0.5
    T_i = torch.randint(min(4, T), T + 1, (B,), generator=g)
    T_i[0] = T
    steps = []
    for t in range(T):
        rows = torch.arange(B) + t * B if n_rows is None else torch.randint(0, n_rows, (B,), generator=g)
        vidx = torch.randint(0, V, (B,), generator=g).int()
        ncand = torch.randint(3, C_max + 1, (B,), generator=g)      # candidates incl. the STOP slot
        if n_rows is not None:
            ncand[int(torch.randint(0, B, (1,), generator=g))] = C_max
        Ct = int(ncand.max())
        cmask = torch.arange(Ct)[None, :] >= ncand[:, None]
        real = torch.arange(Ct)[None, :] < (ncand - 1)[:, None]     # STOP slot + padding are all-zero rows
        crow = torch.where(real, rows[:, None].expand(B, Ct), torch.full((B, Ct), -1))
        cview = torch.randint(0, V, (B, Ct), generator=g).int()
        chead = (torch.rand(B, Ct, generator=g) - 0.5) * 6.0
        celev = (torch.rand(B, Ct, generator=g) - 0.5) * 1.04
        ended = t >= T_i
        tgt = torch.where(t == T_i - 1, ncand - 1, (torch.rand(B, generator=g) * (ncand - 1).float()).long())
        tgt = torch.where(ended, torch.full_like(tgt, -1), tgt)
        ah = torch.rand(B, generator=g) * 6.283 - 3.1415
        st = dict(cand_mask=cmask, angle=angle_feat(ah, torch.zeros(B), ANG), target=tgt, rows=rows,
                  vidx=vidx, crow=crow, cview=cview, chead=chead, celev=celev)
        if table is not None:
            st.update(materialize_step(st, table, ANG))
        steps.append(st)
    return dict(tokens=tokens, lengths=lens, seq_mask=seq_mask, steps=steps, table=table, B=B, L=L, T=T, IMG=IMG, ANG=ANG)


def angle_feat(h, e, ANG=128):                                  # utils/misc.py:285-293
    return torch.stack([h.sin(), h.cos(), e.sin(), e.cos()], -1).repeat_interleave(ANG // 4, dim=-1)


def materialize_step(st, table, ANG=128):
    """The explicit img [B,36,F] / cand [B,C,F] tensors of a step from the ResNet table (any device), built the way the
    reference's marshalling does (agent/base.py:141-157): what the tensor / host feature modes and the CPU baseline consume."""
    import vln_amd
    dev = table.device
    V = table.shape[1]
    loc_table = vln_amd.staging.loc_embedding_table(ANG, V).to(dev)            # [V, V, ANG], misc.py:296-317
    rows, crow = st["rows"].to(dev), st["crow"].to(dev)
    real = (crow >= 0)
    img = torch.cat((table[rows].float(), loc_table[st["vidx"].to(dev).long()]), -1)
    cand = torch.cat((table[crow.clamp_min(0), st["cview"].to(dev).long()].float(),
                      angle_feat(st["chead"].to(dev), st["celev"].to(dev), ANG)), -1) * real[..., None]
    return dict(img=img, cand=cand)


def tape_to(tape, dev, store_dtype=None, host_dtype=None, store=None):
    """Device copy.  With `store_dtype` the tape's own ResNet table becomes a resident DeviceFeatureStore (or `store` = an
    existing one, for index-only tapes) and the per-step img/cand tensors are NOT uploaded (a step only needs its index
    vectors).  With `host_dtype` the per-step img/cand tensors stay on the HOST, pinned, in that dtype (fp32 = what the
    reference's ImageFeatures holds, utils/misc.py:253-279; bf16 = converted once at load time): every step then pays its
    H2D copy (PCIe-inclusive mode, never the headline value)."""
    skip = ("steps", "table")
    out = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in tape.items() if k not in skip}
    out["lengths32"] = tape["lengths"].to(dev, torch.int32)
    drop = ("img", "cand") if (store_dtype is not None or host_dtype is not None or store is not None) else ()
    out["steps"] = [{k: v.to(dev) for k, v in s.items() if k not in drop} for s in tape["steps"]]
    if host_dtype is not None:
        for so, si in zip(out["steps"], tape["steps"]):
            so["img_host"] = si["img"].to("cpu", host_dtype).contiguous().pin_memory()
            so["cand_host"] = si["cand"].to("cpu", host_dtype).contiguous().pin_memory()
    if store is not None:
        out["store"] = store
    elif store_dtype is not None:
        import vln_amd
        out["store"] = vln_amd.DeviceFeatureStore(tape["table"], device=dev, dtype=store_dtype, angle_size=tape["ANG"])
    return out


def build_store(vln, dev, dtype, n_rows=N_VIEWPOINTS, V=36, IMG=2048, ANG=128, seed=2020):
    """The full-size resident feature table: n_rows x 36 x 2048 (1.56 GB in bf16, 3.1 GB in fp32 -- the reference keeps 2.9 GB
    of fp32 on the host), generated on the device chunk by chunk.  Post-ReLU statistics: |N(0,1)| * 0.5."""
    g = torch.Generator(device=dev).manual_seed(seed)
    table = torch.empty(n_rows, V, IMG, dtype=dtype, device=dev)
    for r0 in range(0, n_rows, 512):
        r1 = min(n_rows, r0 + 512)
        table[r0:r1] = (torch.randn(r1 - r0, V, IMG, generator=g, device=dev).abs_() * 0.5).to(dtype)
    return vln.DeviceFeatureStore(table, device=dev, dtype=dtype, angle_size=ANG)


class LiveBatch:
    """The small per-batch tensors of the CURRENT episode batch (tokens, lengths, masks, per-step index vectors, targets,
    angle inputs: ~0.4 MB) at FIXED device addresses.  A trainer marshals every new batch into the same buffers (once per
    iteration), so the modules' step plans and hipGraphs -- keyed by device addresses -- keep replaying while the DATA changes
    every iteration.  Where the packed batches wait (`source`):
      "push"    (round 5, A/B) in PINNED HOST memory; `load(k)` sends batch k (and k + 1) ahead: one asynchronous H2D copy on
                a copy stream into a device-resident ring slot (staging.HostBatchFeed(prefetch=True)), under the previous iteration's
                compute; the iteration's first launch moves it from the slot into the live buffers (same kernel as "pull", reading HBM);
      "pull"    (round 4) in PINNED HOST memory -- what a trainer's data loader hands over; base.py:114-178
                marshals every batch on the host.  `load(k)` stores batch k's address in a pinned slot (one host store) and the
                iteration's FIRST launch pulls the blob through PCIe into the live buffers (staging.HostBatchFeed, vln_host_fetch):
                the agent calls `fetch()` at the top of the iteration, so a captured iteration contains it;
      "copy"    in pinned host memory, `load(k)` = one hipMemcpyAsync H2D in front of the iteration (A/B: +130 us per iteration in
                front of a graph replay, profiles/round4_notes.md);
      "device"  on the device, `load(k)` = one device-to-device copy (round 3's form, A/B)."""
    TOP = ("tokens", "lengths32", "seq_mask")
    STEP = ("rows", "vidx", "crow", "cview", "chead", "celev", "cand_mask", "angle", "target")
    # Blob order: what the ENCODER and the feature gather need first (tokens, lengths, every step's index vectors: 118 KB at B 64 / T 7),
    # then what only the decoder reads (sequence mask, every step's candidate mask, angle features, targets: 242 KB).  `split` is where
    # the second part starts: with the gather riding in the encoder's recurrence launch that part is pulled by a passenger workgroup of
    # the same launch (HostBatchFeed.split_at / RolloutRide.carry_batch_tail) and the iteration's first launch pulls the head only.
    HEAD_TOP, HEAD_STEP = ("tokens", "lengths32"), ("rows", "vidx", "crow", "cview", "chead", "celev")

    def __init__(self, tapes, source="device"):
        assert source in ("push", "pull", "copy", "device")
        self.source = source
        self.send_ahead = True
        t0 = tapes[0]
        self.layout, off, self.split = [], 0, 0
        for name, t in self._items(t0):
            n = t.numel() * t.element_size()
            if name == "seq_mask":
                self.split = off                   # first byte of the decoder-only part
            self.layout.append((name, off, n, t.dtype, tuple(t.shape)))
            off = (off + n + 15) & ~15
        self.nbytes = off
        dev = t0["tokens"].device
        self.live_blob = torch.zeros(self.nbytes, dtype=torch.uint8, device=dev)
        self.feed = None
        if source in ("pull", "push"):
            import vln_amd
            self.feed = vln_amd.HostBatchFeed(self.live_blob, prefetch=source == "push")
        self.blobs = []
        for tp in tapes:
            blob = torch.zeros(self.nbytes, dtype=torch.uint8, device=dev)
            for (name, o, n, dt, shape), (name2, t) in zip(self.layout, self._items(tp)):
                if name != name2 or tuple(t.shape) != shape or t.dtype != dt:
                    raise ValueError(f"LiveBatch: tape layouts differ at {name}: {tuple(t.shape)} vs {shape}")
                blob[o:o + n] = t.contiguous().view(-1).view(torch.uint8)
            if source == "copy":
                blob = blob.cpu().pin_memory()
            elif source in ("pull", "push"):
                blob = self.feed.register(blob)
            self.blobs.append(blob)
        views = {name: self.live_blob[o:o + n].view(dt).view(shape) for name, o, n, dt, shape in self.layout}
        self.live = {k: v for k, v in t0.items() if k not in self.TOP + ("steps",)}
        for k in self.TOP:
            self.live[k] = views[k]
        self.live["steps"] = [{k: views[f"{i}.{k}"] for k in self.STEP} for i in range(len(t0["steps"]))]

    def _items(self, tp):
        for k in self.HEAD_TOP:
            yield k, tp[k]
        for i, s in enumerate(tp["steps"]):
            for k in self.HEAD_STEP:
                yield f"{i}.{k}", s[k]
        for k in self.TOP:
            if k not in self.HEAD_TOP:
                yield k, tp[k]
        for i, s in enumerate(tp["steps"]):
            for k in self.STEP:
                if k not in self.HEAD_STEP:
                    yield f"{i}.{k}", s[k]

    def load(self, k):
        if self.feed is not None:
            self.feed.select(self.blobs[k % len(self.blobs)])       # one host store; the iteration's first launch pulls the blob
            if self.source == "push" and self.send_ahead:           # the loop visits the batches in order: batch k + 1 starts travelling now
                self.feed.send_ahead(self.blobs[(k + 1) % len(self.blobs)])
        else:
            self.live_blob.copy_(self.blobs[k % len(self.blobs)], non_blocking=True)
        return self.live

    def fetch(self):
        """Top of the iteration (eager or inside a capture): the pull of the selected batch, if this LiveBatch pulls."""
        if self.feed is not None:
            self.feed.fetch()

    def launched(self):
        if self.feed is not None:
            self.feed.launched()


class GpuAgent:
    """The caller side of the drop-in modules: the reference's rollout/optimizer sequence for IL."""

    def __init__(self, vln, dev, dtype, world, arena=False, rollout_ce=True, side_gather=False, fused_gather=True):
        self.vln, self.world, self.dtype = vln, world, dtype
        # (A/B option, off by default: measured slower) the step's feature gather reads only the resident table + index
        # vectors, so it can be issued on a side stream beside the encoder / the previous step's kernels
        self.side = torch.cuda.Stream(device=dev) if side_gather else None
        self.copy_stream, self._copy_fenced, self._host_drop = None, False, 0
        self._branch_stream = None
        self._gen_done, self._iter_no, self.prefetch_under_backward = [None, None], 0, True
        self._one = None
        # store-fed steps: the decoder gathers its own rows from the resident table inside its first launch (forward(gather=...))
        # instead of a separate store.gather_step launch in front of every step
        self.fused_gather = bool(fused_gather) and not side_gather
        # teacher forcing: every step's rows are known up front -> ONE gather launch per rollout (store.gather_rollout), A/B option
        self.rollout_gather = False
        self.clear_grads_in_step = False
        self.rollout_ce = rollout_ce
        self.enc = vln.EncoderLSTM(992, 256, 512, 0, 0.5, True, 1, compute_dtype=dtype).to(dev)
        self.dec = vln.EnvDropDecoder(512, 0.5, 0.3, 64, 128, 2176, compute_dtype=dtype).to(dev)
        self.enc.train(); self.dec.train()
        # teacher forcing: nothing reads the logits before the loss, so the decoder leaves them to be formed for the whole
        # rollout at once when losses.RolloutCE evaluates (one GEMM over steps x batch + one dot launch instead of two
        # launches on every step's dependent chain)
        self.dec.defer_logits = bool(rollout_ce)
        # ... and consecutive steps are chained: a step's last elementwise stage rides in the next step's first launch, forward
        # and backward (vln_envdrop_step.chain; needs the deferred logits: nothing reads a step's h_tilde but the next step)
        self.dec.chain_steps = bool(rollout_ce)
        # trainer.py:380-381,423-427: RMSprop(lr) + clip_grad_norm(40) per module -- fused over flat buffers; the flat
        # gradient buffer doubles as the RCCL all-reduce bucket (optim.FusedRMSprop)
        self.opt = vln.optim.FusedRMSprop([list(self.enc.parameters()), list(self.dec.parameters())], lr=LR, clip_norm=CLIP)
        if world > 1:   # the decoder's 34.7 MB of gradients are final before the encoder's BPTT starts: reduce them under it
            self.dec.grads_ready_hook = lambda: self.opt.start_allreduce(1)
        # A training loop allocates the same sequence of buffers every iteration: with the arena they come back at the
        # same device addresses, so each decoder step (13 forward / 15 backward launches) replays as one hipGraph.
        self.arena = None
        self.use_arena(arena)
        # runtime.DeviceClock: dropout offsets and the recurrence's launch sequence come from device words that one tick
        # launch bumps per iteration -> the iteration's launch arguments repeat and it can be captured whole (graphs.IterationGraph)
        self.clock = None
        self.graph = None
        # Segmented form of the iteration (graphs.SegmentedIterationGraph): the backward is cut at the encoder's outputs so that
        # the host can start the decoder slice's all-reduce between the two halves -- the data-parallel path (N > 1, --dp-path)
        self.segmented = False
        self._cut = None
        self.batch_fetch = None         # LiveBatch.fetch / .launched when the batches are pulled from pinned host memory
        self.batch_launched = None
        self.batch_feed = None
        self.use_prologue = True        # pull + tick + shadow refreshes as one launch (runtime.DeviceClock.prologue)
        self.split_pull = True          # the decoder-only part of a pulled batch crosses PCIe under the encoder's recurrence (ride_gather only)
        self._live_split, self._live_tape = 0, None
        self.gather_branch = False      # graph mode A/B: the rollout-wide gather as a captured branch beside the encoder
        self.ride_gather = False        # the rollout-wide gather as passenger workgroups of the encoder's recurrence launch
        self.ride_shadows = False       # ... which then also refresh the decoder's weight shadows, out of the prologue launch (--ride-shadows: measured neutral)

    def _probe(self):
        n = getattr(self, "probe_trivial", 0)
        if n:
            if getattr(self, "_probe_buf", None) is None:
                self._probe_buf = torch.zeros(2, 64 * 512, device=next(self.enc.parameters()).device)
            lib = self.vln._lib.load()
            self.vln._lib.check(lib.vln_debug_trivial_chain(self._probe_buf[0].data_ptr(), self._probe_buf[1].data_ptr(), 64 * 512, n, 256,
                                                            self.vln._lib.raw_stream()), "vln_debug_trivial_chain")

    def use_clock(self, store=None):
        self.clock = self.vln.DeviceClock(next(self.enc.parameters()).device)
        self.clock.attach(self.enc, self.dec)
        if store is not None:
            self.clock.attach(store)
        return self.clock

    def capture(self, tape):
        """Record one iteration over `tape` (buffers at fixed addresses: LiveBatch.live) as ONE hipGraph; `replay()` then runs
        an iteration on whatever those buffers hold."""
        if self.clock is None:
            raise RuntimeError("GpuAgent.capture: use_clock() first (a captured iteration reads its dropout offsets from device words)")
        if self.segmented:
            self.graph = self.vln.SegmentedIterationGraph(self.segments(tape), self.clock).capture()
            return self.graph
        self.graph = self.vln.IterationGraph(lambda: self.iteration(tape), self.clock).capture(
            debug_dump=getattr(self, "dump_graph", None), capture_error_mode=getattr(self, "capture_error_mode", "global"))
        return self.graph

    def replay(self):
        out = self.graph.replay()
        if self.batch_launched is not None:
            self.batch_launched()
        return out

    def use_live(self, live):
        """A LiveBatch whose batches are PULLED from pinned host memory: the iteration's first launch is the pull
        (staging.HostBatchFeed); after every iteration / replay an event bounds how far the host may run ahead."""
        if live.feed is not None:
            self.batch_fetch, self.batch_launched, self.batch_feed = live.fetch, live.launched, live.feed
            self._live_split, self._live_tape = live.split, live.live

    def use_arena(self, on: bool):
        self.arena = self.vln.ops.RolloutArena() if on else None
        self.dec.step_graphs = bool(on)

    def step_features(self, tape, s):
        """Per-step marshalling (agent/base.py:141-157 + the EnvDrop feature dropout, policy.py:226-231).
        store path: ONE gather pass per tensor from the HBM-resident table (indices in, dropped features + bf16 stream
        copy out); tensor path: fresh copies of pre-built feature tensors, the decoder applies the dropout in place."""
        store = tape.get("store")
        if "img_host" in s:
            return self.stage_from_host(s)
        if store is None:
            return s["img"].clone(), s["cand"].clone(), {}
        if self.fused_gather:
            return None, None, dict(gather=(store, s["rows"], s["vidx"], s["crow"], s["cview"], s["chead"], s["celev"]))
        lp = self.dtype != torch.float32
        pf = self.dec.feat_drop_ratio if self.dec.training else 0.0
        # bf16 decoder: only the bf16 rows exist (nothing on this path reads fp32 features)
        (img, img_lp), (cand, cand_lp), _ = store.gather_step(s["rows"], s["vidx"], s["crow"], s["cview"], s["chead"], s["celev"],
                                                              pf, want_bf16=lp, want_f32=not lp, stream=self.side)
        kw = dict(already_dropfeat=True)
        return (img_lp, cand_lp, kw) if lp else (img, cand, kw)

    def stage_from_host(self, s):
        """Host-resident (pinned) features: hipMemcpyAsync on a copy stream into per-step device buffers, the compute stream
        waits for the step's copy only -- the copies of later steps run under the encoder / earlier steps (north star:
        'pinned and hipMemcpyAsync-streamed to HBM overlapped').  bf16 host features get the feature dropout here (the
        decoder only takes non-fp32 features that are already dropped)."""
        ops = self.vln.ops
        if self.copy_stream is None:
            self.copy_stream = torch.cuda.Stream()
            self.copy_events = {}
        main = torch.cuda.current_stream()
        if ops.current_arena() is None or not self._copy_fenced:   # buffers may still be in use by earlier work on `main`
            self.copy_stream.wait_stream(main)
            self._copy_fenced = True
        ih, ch = s["img_host"], s["cand_host"]
        img = ops.empty(ih.shape, dtype=ih.dtype, device=main.device)
        cand = ops.empty(ch.shape, dtype=ch.dtype, device=main.device)
        with torch.cuda.stream(self.copy_stream):
            img.copy_(ih, non_blocking=True)
            cand.copy_(ch, non_blocking=True)
        ev = self.copy_events.get(id(s))
        if ev is None:
            ev = self.copy_events[id(s)] = torch.cuda.Event()
        ev.record(self.copy_stream)
        main.wait_event(ev)
        if ih.dtype == torch.float32:
            return img, cand, {}                                   # the decoder drops in place + writes its bf16 stream copies
        pf = self.dec.feat_drop_ratio if self.dec.training else 0.0
        if pf > 0:
            F, ANG = self.dec.feature_size, self.dec.angle_feat_size
            self._host_drop += 2
            ops.feat_dropout_inplace(img, F - ANG, ANG, 0x51A6E, self._host_drop, pf)
            ops.feat_dropout_inplace(cand, F - ANG, ANG, 0x51A6E, self._host_drop + 1, pf)
        return img, cand, dict(already_dropfeat=True)

    def iteration(self, tape):
        out = self._iteration_eager(tape)
        if self.batch_launched is not None and not torch.cuda.is_current_stream_capturing():
            self.batch_launched()
        return out

    def _iteration_eager(self, tape):
        if self.segmented:             # the same five pieces a SegmentedIterationGraph captures / replays, issued eagerly
            out = None
            for _, fn in self.segments(tape):
                r = fn()
                out = r if r is not None else out
            return out
        self._copy_fenced = False
        self.vln.ops.set_arena(self.arena)
        if self.arena is not None:
            self.arena.begin()
        try:
            return self._iteration(tape)
        finally:
            self.vln.ops.set_arena(None)

    def segments(self, tape):
        """The iteration cut at its two exchange points (trainer.py:421-427 with the gradient all-reduce in it):
        [graph A: forward, loss, the decoder's backward] [host: start the decoder slice's all-reduce] [graph B: the encoder's
        backward] [host: reduce the rest, wait] [graph C: clip + update]."""
        def in_arena(fn, begin=False):
            def run():
                self.vln.ops.set_arena(self.arena)
                if begin and self.arena is not None:
                    self.arena.begin()
                try:
                    return fn()
                finally:
                    self.vln.ops.set_arena(None)
            return run

        def part_a():
            self._copy_fenced = False
            return self._iteration(tape)

        def part_b():
            self._cut.resume()

        def part_c():
            self.opt.step(zero_grads=self.clear_grads_in_step)
            self._iter_no += 1

        return [("graph", in_arena(part_a, begin=True)), ("host", lambda: self.opt.start_allreduce(1)),
                ("graph", in_arena(part_b)), ("host", lambda: self.opt.allreduce()), ("graph", in_arena(part_c))]

    def _shadows_ride(self, tape):
        """The decoder's weight shadows are refreshed by the gather ride's passengers (staging.RolloutRide.carry_shadows) instead of
        the prologue launch: whenever there is a ride and a prologue to take them out of."""
        return bool(self.ride_shadows and self.ride_gather and tape.get("store") is not None and self.clock is not None and self.use_prologue)

    def _iteration(self, tape):
        B = tape["B"]
        # the decoder-only part of a pulled batch crosses PCIe under the encoder's recurrence (one passenger workgroup of that launch)
        # when the rollout's gather rides there too; decided BEFORE the head fetch of this iteration is issued
        carry_tail = bool(self.batch_feed is not None and self.split_pull and self._live_split and tape is self._live_tape and
                          self.ride_gather and tape.get("store") is not None)
        if self.batch_feed is not None:
            self.batch_feed.split_at(self._live_split if carry_tail else 0)
        if self.clock is not None and self.use_prologue:
            # ONE launch: the GPU pulls the selected batch out of pinned host memory (LiveBatch "pull"), the device clock ticks (this
            # iteration's dropout offsets / launch sequence) and both modules' weight shadows follow the last optimizer step
            # (the decoder's shadows ride in the encoder's recurrence launch instead when the gather does: carry_shadows below)
            self.clock.prologue(self.batch_feed, (self.enc,) if self._shadows_ride(tape) else (self.enc, self.dec))
        else:
            if self.batch_fetch is not None:
                self.batch_fetch()     # one launch: the pull
            if self.clock is not None:
                self.clock.tick()      # one launch: the tick
        self._probe()
        if self.side is not None:      # once per iteration: the side stream's gathers write buffers last read two iterations ago
            self.side.wait_stream(torch.cuda.current_stream())
        if self.copy_stream is not None and self.arena is not None:
            # Fence for the H2D copy stream (host features).  The arena alternates between two buffer generations, so this
            # iteration's staging buffers were last read TWO iterations ago: the copies only wait for the end of that
            # iteration and run under the previous iteration's backward (the link is busy for the whole iteration instead
            # of the forward only: 'streamed to HBM overlapped with backward', north star / base.py:141-157).
            ev = self._gen_done[self._iter_no & 1]
            if ev is not None and self.prefetch_under_backward:
                self.copy_stream.wait_event(ev)
            else:
                self.copy_stream.wait_stream(torch.cuda.current_stream())
            self._copy_fenced = True
        self.opt.zero_grad()
        pre, branch = None, None

        def gather_all():
            lp = self.dtype != torch.float32
            pf = self.dec.feat_drop_ratio if self.dec.training else 0.0
            return tape["store"].gather_rollout([(s["rows"], s["vidx"], s["crow"], s["cview"], s["chead"], s["celev"]) for s in tape["steps"]],
                                                pf, want_bf16=lp, want_f32=not lp)

        ride = None
        if self.ride_gather and tape.get("store") is not None:
            # the rollout's gather rides in the encoder's persistent recurrence launch (passenger workgroups on its idle CUs)
            lp = self.dtype != torch.float32
            pf = self.dec.feat_drop_ratio if self.dec.training else 0.0
            ride = tape["store"].rollout_ride([(s["rows"], s["vidx"], s["crow"], s["cview"], s["chead"], s["celev"]) for s in tape["steps"]],
                                              pf, want_bf16=lp, want_f32=not lp)
            pre = ride.outputs
            if carry_tail:
                ride.carry_batch_tail(self.batch_feed)
            if self._shadows_ride(tape):
                ride.carry_shadows((self.dec,))
        elif self.rollout_gather and self.gather_branch and tape.get("store") is not None:
            # the gather reads only the resident table + index vectors: as a branch of the captured graph it runs beside the
            # instruction encoder (whose 0.2 ms recurrence keeps half of the CUs idle) and joins before the first decoder step
            main = torch.cuda.current_stream()
            if self._branch_stream is None:
                self._branch_stream = torch.cuda.Stream()
            branch = self._branch_stream
            branch.wait_stream(main)
            with torch.cuda.stream(branch):
                pre = gather_all()
        ctx, h_t, c_t = self.enc(tape["tokens"], tape["lengths32"], ride=ride) if ride is not None else self.enc(tape["tokens"], tape["lengths32"])
        if self.segmented:
            # the decoder's backward ends at these leaves; the encoder's backward starts from their .grad (segments(): part_b)
            self._cut = self.vln.dp.BackwardCut()
            ctx, h_t, c_t = self._cut.at(ctx, h_t, c_t)
        h_tilde = h_t
        terms = []
        ce = self.vln.losses.RolloutCE() if self.rollout_ce else None
        if branch is not None:
            torch.cuda.current_stream().wait_stream(branch)
        elif pre is None and self.rollout_gather and tape.get("store") is not None:
            pre = gather_all()
        for t, s in enumerate(tape["steps"]):
            if pre is not None:
                (im, im_lp), (cd, cd_lp) = pre[t]
                img, cand, kw = (im_lp, cd_lp, dict(already_dropfeat=True)) if im_lp is not None else (im, cd, dict(already_dropfeat=True))
            else:
                img, cand, kw = self.step_features(tape, s)
            logits, (h_t, c_t), h_tilde = self.dec(s["angle"], img, cand, h_tilde, h_t, c_t, ctx, tape["seq_mask"], **kw)
            # envdrop.py:173-179: masked_fill_(-inf) + CrossEntropyLoss(ignore_index=-1, reduction="none").sum() (SURVEY §8 row
            # A9): recorded per step, evaluated for the whole rollout in ONE launch (losses.RolloutCE) -- or, --ce per-step,
            # one fused launch per step
            if ce is not None:
                ce.add(logits, s["target"], s["cand_mask"])
            else:
                terms.append(self.vln.losses.masked_cross_entropy(logits, s["target"], s["cand_mask"], "sum"))
        w = ML_WEIGHT / (B * self.world)                         # envdrop.py:268; global batch normalisation under DP
        if ce is not None:
            loss = ce.sum(scale=w)                               # ml_loss summed over the steps (envdrop.py:179), scaled in the launch
        else:
            loss = torch.stack(terms).sum() * w
        if self._one is None or self._one.device != loss.device:
            self._one = torch.ones((), dtype=loss.dtype, device=loss.device)
        self._probe()
        loss.backward(self._one)                                 # the root gradient is a constant: no ones_like fill per iteration
        if self.segmented:
            return loss
        self.opt.allreduce()
        # bench: the update clears the gradients it consumed (the next zero_grad() is free); tests keep them to look at
        self.opt.step(zero_grads=self.clear_grads_in_step)
        if self.copy_stream is not None and self.arena is not None:
            g = self._iter_no & 1
            if self._gen_done[g] is None:
                self._gen_done[g] = torch.cuda.Event()
            self._gen_done[g].record()                          # this generation's buffers are free again from here
        self._iter_no += 1
        return loss


def cpu_baseline(tape, P_enc, P_dec):
    """The CPU oracle driven identically (dropout sampled with bernoulli_ like nn.Dropout).  Returns run(warm, iters, budget_s)
    -> (median seconds per iteration, iterations timed)   (2 warm-ups, median)."""
    from oracle import torch_port as O
    B, ANG = tape["B"], tape["ANG"]
    params = [p.requires_grad_(True) for p in list(P_enc.values()) + list(P_dec.values())]
    opt = torch.optim.RMSprop(params, lr=LR)

    def mask(shape, p):
        return torch.empty(shape).bernoulli_(1 - p).div_(1 - p)

    def one():
        opt.zero_grad()
        L = tape["L"]
        ctx, h_t, c_t = O.encoder_forward(P_enc, tape["tokens"], tape["lengths"].tolist(), num_layers=1, bidirectional=True,
                                          emb_mask=mask((B, L, 256), 0.5), ctx_mask_drop=mask((B, L, 512), 0.5))
        h_tilde, ml = h_t, 0.
        for s in tape["steps"]:
            img = O.feature_dropout(s["img"], mask(s["img"][..., :-ANG].shape, 0.3), ANG)
            cand = O.feature_dropout(s["cand"], mask(s["cand"][..., :-ANG].shape, 0.3), ANG)
            drop = {"act": mask((B, 64), 0.5), "hprev": mask((B, 512), 0.5), "h1": mask((B, 512), 0.5), "htilde": mask((B, 512), 0.5)}
            logit, (h_t, c_t), h_tilde, _ = O.envdrop_step(P_dec, s["angle"], img, cand, h_tilde, c_t, ctx, tape["seq_mask"], drop=drop)
            ml = ml + O.masked_cross_entropy(logit, s["target"], s["cand_mask"], "sum")
        (ml * ML_WEIGHT / B).backward()
        torch.nn.utils.clip_grad_norm_(list(P_enc.values()), CLIP)
        torch.nn.utils.clip_grad_norm_(list(P_dec.values()), CLIP)
        opt.step()

    def run(warm, iters, budget):
        t0 = time.perf_counter()
        for _ in range(warm):
            one()
            if time.perf_counter() - t0 > budget:           # pathological host (e.g. CPU quota): keep the bench bounded
                return time.perf_counter() - t0, 0
        ts = []
        t0 = time.perf_counter()
        while len(ts) < iters and (time.perf_counter() - t0) < budget:
            t1 = time.perf_counter()
            one()
            ts.append(time.perf_counter() - t1)
        ts.sort()
        return ts[len(ts) // 2], len(ts)

    return run


def usable_cores() -> int:
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // per))
        except (OSError, ValueError, IndexError):
            pass
    return max(1, n)


def read_prof(lib, nk):
    rows = []
    for k in range(nk):
        n, ms, by = C.c_int64(), C.c_double(), C.c_double()
        lib.vln_prof_read(k, C.byref(n), C.byref(ms), C.byref(by))
        if n.value:
            rows.append(dict(kernel=lib.vln_prof_kernel_name(k).decode(), launches=n.value, ms=ms.value, bytes=by.value))
    return rows


def launch_ranks(n: int) -> int:
    """Start `n` ranks of this script under torch.distributed.run (one process per GPU) and return its exit code.  Called
    before anything in this process has initialised the GPU; the children are ordinary subprocesses (no exec)."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--len", type=int, default=80, dest="L")
    ap.add_argument("--T", type=int, default=7)
    ap.add_argument("--cpu-iters", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary numbers (fp32, host features, IL+A2C, Self-Monitor)")
    ap.add_argument("--viewpoints", type=int, default=N_VIEWPOINTS, help="rows of the resident ResNet table (R2R: 10,567)")
    ap.add_argument("--tapes", type=int, default=N_TAPES, help="distinct episode batches rotated through the timed loop")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-arena", action="store_true", help="allocate per-iteration buffers with torch.empty (no address-stable "
                                                           "arena, hence no decoder-step hipGraph replay)")
    ap.add_argument("--features", default="store", choices=["store", "tensor", "host", "host-bf16"],
                    help="store: ResNet table resident in HBM, a step ships indices (DeviceFeatureStore); "
                         "tensor: pre-built per-step feature tensors, cloned each step; host / host-bf16: per-step features in "
                         "pinned HOST memory (fp32 like the reference / bf16 converted once), hipMemcpyAsync per step on a copy "
                         "stream -- the PCIe-inclusive rate (DESIGN.md §6), never the headline value")
    ap.add_argument("--ce", default="rollout", choices=["rollout", "per-step"],
                    help="rollout: the IL loss of all T steps in one launch after the last step (losses.RolloutCE); per-step: one "
                         "fused CE launch per decoder step")
    ap.add_argument("--gather-stream", default="main", choices=["side", "main"],
                    help="main: the per-step feature gather in line on the compute stream; side: on a side stream (it depends on "
                         "no decoder output) -- measured SLOWER (2.28 vs 2.11 ms/iteration: the gathers land beside the persistent "
                         "recurrence and slow its hand-offs, and the per-step event pair costs host time), kept for A/B")
    ap.add_argument("--wgrad", default="bf16", choices=["split", "bf16"],
                    help="bf16 mode: weight gradients from split-bf16 operands (three MFMAs per product, fp32-grade) or from plain "
                         "bf16 operands (one MFMA, mixed-precision standard)")
    ap.add_argument("--no-backward-prefetch", action="store_true",
                    help="host features A/B: the H2D copies of an iteration wait for the END of the previous iteration (forward-only "
                         "overlap, round 1) instead of the end of the one before it (they then run under the previous backward)")
    ap.add_argument("--iteration-graph", default="auto", choices=["auto", "on", "off"],
                    help="the WHOLE iteration (encoder, decoder steps, loss, backward, clip + RMSprop) captured as one hipGraph and "
                         "replayed (graphs.IterationGraph; dropout offsets and the recurrence's launch sequence come from device words, "
                         "runtime.DeviceClock).  auto: on for one GPU with the resident feature store, off otherwise (the gradient "
                         "all-reduce of N > 1 stays a stream operation between launches)")
    ap.add_argument("--tunable", action="append", default=[], metavar="ID=VALUE",
                    help="(A/B) vln_set_tunable(ID, VALUE) before anything runs, e.g. --tunable 0=256 (csrc/vln_internal.h lists them)")
    ap.add_argument("--dump-graph", default=None, metavar="PATH",
                    help="write the captured iteration graph's nodes (kernel names, memcpy / memset nodes, edges) as graphviz text")
    ap.add_argument("--probe-trivial", type=int, default=0,
                    help="(measurement) N trivial dependent launches (vln_debug_trivial_chain) at the top of every iteration and "
                         "N more between the forward and the backward: (ms with N - ms without) / 2N = the price of a kernel "
                         "boundary inside this very graph (rocprofv3 reports a ~4.7 us floor for ANY short kernel; unprofiled: 1.95 us)")
    ap.add_argument("--inject-capture-failure", action="store_true",
                    help="(test) make the whole-iteration graph capture fail: the run must fall back to eager launches and say so")
    ap.add_argument("--inject-timeout", type=int, default=0, metavar="K",
                    help="(test) raise the sticky timeout word before untimed iteration K, as a persistent recurrence whose "
                         "workgroups were not co-resident would: exercises the fallback to per-step launches")
    ap.add_argument("--ride-gather", default="auto", choices=["auto", "on", "off"],
                    help="store features, teacher forcing: the rollout's feature gather as PASSENGER workgroups of the encoder's "
                         "persistent recurrence launch (the 128 CUs that launch leaves idle at B = 64); the decoder steps then "
                         "start with their prep launch only.  auto = on (profiles/round3_notes.md: 1.661 vs 1.687 ms)")
    ap.add_argument("--ride-shadows", action="store_true",
                    help="(A/B) the decoder's weight shadows are refreshed by the gather ride's passengers under the encoder's recurrence "
                         "(staging.RolloutRide.carry_shadows) instead of the prologue launch: the prologue is bound by its PCIe pull, not "
                         "by the refresh, so this only pays together with --batch-source push -- and measured the same "
                         "(profiles/round5_notes.md section 11)")
    ap.add_argument("--gather-branch", action="store_true",
                    help="with --rollout-gather and the iteration graph: the rollout-wide gather as a captured BRANCH beside the encoder")
    ap.add_argument("--rollout-gather", action="store_true",
                    help="store features: ONE gather launch for all T steps ahead of the rollout (teacher forcing: the path is known), "
                         "A/B against the gather inside every step's first launch")
    ap.add_argument("--separate-gather", action="store_true",
                    help="store features: one store.gather_step launch in front of every decoder step (A/B) instead of the gather "
                         "inside the step's first launch")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL); 'gloo' only to smoke-test "
                                                      "the N>1 code path on a single-GPU box")
    ap.add_argument("--batch-source", default="pull", choices=["push", "pull", "copy", "device"],
                    help="where the packed episode batches (tokens, masks, per-step index vectors, targets: ~0.4 MB each) wait: "
                         "push = pinned HOST memory (what a data loader hands over), sent ahead by an asynchronous H2D copy on a copy "
                         "stream into a device ring slot under the previous iteration, the iteration's first launch moves it into the "
                         "live buffers; pull = pinned host memory, the iteration's first launch pulls the batch through PCIe itself "
                         "(round 4); copy = pinned host memory, one hipMemcpyAsync H2D in front of the iteration; device = "
                         "device memory, one device-to-device copy (round 3's form)")
    ap.add_argument("--no-prologue", action="store_true", help="(A/B) the batch pull, the clock tick and the shadow refreshes as separate launches")
    ap.add_argument("--no-ride-wgrads", action="store_true", help="(A/B) the decoder's weight / bias gradients as their own launches in front of "
                    "the encoder's BPTT instead of passengers of its launch (ops.GradRide); N > 1 and --dp-path never ride")
    ap.add_argument("--no-dx-post", action="store_true", help="(A/B) the encoder backward's d x product as its own launch behind the weight "
                    "gradients instead of extra workgroups of their pack launch (EncoderLSTM.dx_with_wgrads, vln_linear_fwd_post)")
    ap.add_argument("--no-split-pull", action="store_true", help="(A/B) the iteration's first launch pulls the WHOLE batch blob instead of leaving the "
                    "decoder-only part to a passenger workgroup of the encoder's recurrence launch")
    ap.add_argument("--no-chain", action="store_true", help="(A/B) decoder steps not chained: every step issues its own last stage")
    ap.add_argument("--no-project-context", action="store_true", help="(A/B) the decoder projects its text-attention query every step "
                    "(round 4's step: 8 dependent launches per direction) instead of scoring on K = ctx W_in formed once per rollout")
    ap.add_argument("--dp-segments", action="store_true", help="with --dp-path / N > 1: (A/B, round 4's form) the iteration as three graph segments with "
                    "host-issued collectives between them instead of ONE graph with the process group's collectives captured inside")
    ap.add_argument("--dp-path", action="store_true",
                    help="N = 1 only: run the DATA-PARALLEL form of the iteration -- three hipGraph segments with the gradient "
                         "exchange issued between them (graphs.SegmentedIterationGraph) on a ONE-rank RCCL group, collectives "
                         "forced on -- so that the path the N > 1 runs take is timed against the single graph on one GPU")
    ap.add_argument("--one-device", action="store_true", help="(testing) map every rank to cuda:0")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="(testing, runs without a GPU) launch / join the N ranks, all-reduce one scalar over --backend, rank 0 "
                         "prints {n_gpus, world_size, backend} and every rank leaves: checks the launch path of --gpus N")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU, RCCL rendezvous
        # on 127.0.0.1) BEFORE this process touches the GPU, and leave with the launcher's exit code -- rank 0 of the
        # children prints the one JSON line.
        sys.exit(launch_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    local = 0 if args.one_device else int(os.environ.get("LOCAL_RANK", "0"))
    if args.rendezvous_only:
        import torch.distributed as dist
        if world > 1:
            dist.init_process_group(args.backend)
            one = torch.ones(1)
            dist.all_reduce(one)
            assert int(one.item()) == world == dist.get_world_size()
        if rank == 0:
            print(json.dumps({"n_gpus": world, "rendezvous_only": True,
                              "config": {"world_size": world, "backend": args.backend if world > 1 else None}}))
        if world > 1:
            dist.destroy_process_group()
        return
    if world > 1 or args.dp_path:
        os.environ.setdefault("NCCL_DEBUG", "WARN")          # RCCL's version banner goes to STDOUT: rank 0 owes the driver one JSON line there
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.backend)
        assert dist.get_world_size() == args.gpus and dist.get_backend() == args.backend, (dist.get_world_size(), dist.get_backend())
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    import vln_amd as vln
    lib = vln._lib.load()                                        # fails loudly if the HIP extension is missing
    if args.dp_path:
        if world != 1:
            raise SystemExit("--dp-path is the one-GPU rehearsal of the N > 1 path: use it with --gpus 1")
        import socket
        import torch.distributed as dist
        if "MASTER_PORT" not in os.environ:
            sk = socket.socket(); sk.bind(("127.0.0.1", 0)); os.environ["MASTER_PORT"] = str(sk.getsockname()[1]); sk.close()
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        vln.dp._dp_active = lambda group=None: True              # a one-rank group: issue the collectives anyway
    vln.ops.set_wgrad_precision(args.wgrad)
    for tv in args.tunable:
        tid, val = tv.split("=")
        vln._lib.check(lib.vln_set_tunable(int(tid), int(val)), "vln_set_tunable")
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    if args.features == "host-bf16" and dtype != torch.bfloat16:
        raise SystemExit("--features host-bf16 needs --dtype bf16")
    torch.manual_seed(2020)
    agent = GpuAgent(vln, dev, dtype, world, arena=not args.no_arena, rollout_ce=args.ce == "rollout",
                     side_gather=args.gather_stream == "side" and args.features == "store", fused_gather=not args.separate_gather)
    agent.clear_grads_in_step = True
    agent.prefetch_under_backward = not args.no_backward_prefetch
    agent.rollout_gather = bool(args.rollout_gather)
    agent.gather_branch = bool(args.gather_branch)
    agent.ride_gather = args.features == "store" and args.ride_gather != "off" and not args.rollout_gather
    agent.ride_shadows = bool(args.ride_shadows)
    agent.enc.dx_with_wgrads = not args.no_dx_post
    agent.enc.layout_with_bridge = not args.no_dx_post
    agent.probe_trivial = int(args.probe_trivial)
    agent.dump_graph = args.dump_graph
    if args.no_chain:
        agent.dec.chain_steps = False
    if args.no_project_context:
        agent.dec.project_context = False
    if args.no_prologue:
        agent.use_prologue = False
    if args.no_split_pull:
        agent.split_pull = False
    use_graph = args.iteration_graph == "on" or (args.iteration_graph == "auto" and args.features == "store" and not args.no_arena)
    # N > 1 (and --dp-path): the iteration as three graph segments with the gradient exchange issued between them -- the same
    # kernels in the same order as the single graph of N = 1 (graphs.SegmentedIterationGraph)
    args.dp_capture = not args.dp_segments
    agent.segmented = bool(use_graph and (world > 1 or args.dp_path) and not args.dp_capture)
    if args.dp_capture and (world > 1 or args.dp_path):
        agent.dec.grads_ready_hook = lambda: agent.opt.start_allreduce(1)      # (N = 1 rehearsal: the hook GpuAgent sets for world > 1)
        agent.capture_error_mode = "thread_local"
    # one GPU: the decoder's parameter gradients ride in the encoder's BPTT launch (a data-parallel rank wants them final before it)
    agent.dec.ride_wgrads = bool(world == 1 and not args.dp_path and not args.no_ride_wgrads and args.dtype != "fp32")
    if agent.segmented:
        agent.dec.grads_ready_hook = None                        # the early slice goes out between segments A and B instead
    if use_graph and (args.features != "store" or args.ce != "rollout"):
        raise SystemExit("--iteration-graph on needs --features store and --ce rollout (inputs at fixed addresses, no host sync)")
    # The resident feature table is the FULL-size one (10,567 viewpoints x 36 x 2048: 1.56 GB bf16 / 3.1 GB fp32), and the
    # timed loop rotates through N_TAPES different episode batches (new tokens, new viewpoints every iteration): the gather
    # reads rows that were last touched 8 iterations ago out of a table six times the Infinity Cache, i.e. from HBM.
    t_setup = time.perf_counter()
    store = build_store(vln, dev, dtype, args.viewpoints)
    cpu_tapes = [make_tape(args.batch, args.L, args.T, 8, seed=2020 + 97 * rank + k, n_rows=store.N) for k in range(args.tapes)]
    if args.features == "store":
        tapes = [tape_to(t, dev, store=store) for t in cpu_tapes]
    else:                          # explicit per-step feature tensors, built from the same table rows
        hd = {"host": torch.float32, "host-bf16": torch.bfloat16}.get(args.features)
        tapes = []
        for t in cpu_tapes:
            for s in t["steps"]:
                s.update(materialize_step(s, store.table))
            tapes.append(tape_to(t, dev, host_dtype=hd))
            if hd is not None:
                for s in t["steps"]:
                    del s["img"], s["cand"]
    live = LiveBatch(tapes, source=args.batch_source) if args.features == "store" else None
    if live is not None:
        agent.use_live(live)
    if rank == 0:
        print(f"[bench] setup: {store.N}-viewpoint table ({store.table.numel() * store.table.element_size() / 2**30:.2f} GiB {args.dtype}), "
              f"{len(tapes)} tapes, {time.perf_counter() - t_setup:.1f} s", file=sys.stderr, flush=True)
    it_no = [0]
    if use_graph:
        agent.use_clock(store)

    def iterate_eager():
        k = it_no[0]
        it_no[0] = k + 1
        return agent.iteration(live.load(k) if live is not None else tapes[k % len(tapes)])

    def iterate():
        if agent.graph is None:
            return iterate_eager()
        k = it_no[0]
        it_no[0] = k + 1
        live.load(k)                  # the new batch: its address into the pinned slot ring (or one copy), then ONE graph launch
        return agent.replay()

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    raised = [0]

    def warm_iterate():
        """`iterate()` of the untimed phases.  A bounded in-kernel wait that timed out (the persistent recurrence's workgroups
        were not co-resident) surfaces as VlnError from the NEXT library entry on this rank only -- the library has already
        switched this process to per-step launches, whose results are the same.  The rank completes the iteration's gradient
        exchange (the other ranks are inside it), remembers, and goes on; at the end of the phase every rank learns of it
        through one all-reduced flag and takes the fallback together."""
        try:
            if args.inject_timeout and it_no[0] == args.inject_timeout:
                vln._lib.check(lib.vln_debug_raise_sticky(0), "vln_debug_raise_sticky")
            return iterate()
        except vln.VlnError as e:
            if "timed out" not in str(e):
                raise
            print(f"[bench] rank {rank}: {e}", file=sys.stderr, flush=True)
            raised[0] = 1
            agent.opt.abandon_iteration(early_groups=(1,))
            return None

    # Warm-up runs like the timed loop: iterations back to back, no per-iteration sync (the first time the host gets
    # many launches ahead of the GPU the runtime grows its in-flight pools: a one-time cost that belongs here).
    # Initialisation (not warm-up): the first four iterations build what later iterations only replay -- library load,
    # weight shadows, the arena's two buffer generations, one hipGraph capture and one step plan per decoder step and
    # generation.  Like a JIT compile this happens once per process, whatever W is.
    tw = time.perf_counter()
    for i in range(4):
        warm_iterate()
        if i == 0:
            torch.cuda.synchronize()
            if rank == 0:
                print(f"[bench] first iteration (module init, captures): {(time.perf_counter() - tw) * 1e3:.1f} ms", file=sys.stderr, flush=True)
    torch.cuda.synchronize()
    if use_graph:
        tg = time.perf_counter()
        try:
            if args.inject_capture_failure:
                raise RuntimeError("injected capture failure (--inject-capture-failure)")
            agent.capture(live.live)
        except vln.VlnError as e:                  # a timeout in the four iterations above: recorded after the fallback below
            if "timed out" not in str(e):
                raise
            raised[0] = 1
        except Exception as e:                     # noqa: BLE001 -- stream capture itself failed on this box / runtime: the bench
            # line is still owed.  The eager path (per-step graphs, same kernels, same results) is what runs instead, and the
            # JSON line says so (config.iteration_graph false).
            print(f"[bench] the iteration could not be captured as a hipGraph ({type(e).__name__}: {e}); eager launches instead",
                  file=sys.stderr, flush=True)
            agent.graph = None
            use_graph = False
            torch.cuda.synchronize()
        for _ in range(2):
            warm_iterate()
        torch.cuda.synchronize()
        if rank == 0 and use_graph:
            print(f"[bench] iteration captured as one hipGraph: {(time.perf_counter() - tg) * 1e3:.0f} ms", file=sys.stderr, flush=True)
    # Python's cyclic GC: a full pass over the (static) module/object graph costs tens of ms and would land in the
    # timed region at random; collect now and move the survivors out of the collector's reach.  (Before the warm-up
    # iterations, not after them: tens of ms of idle GPU right in front of the timed region let the clocks fall back, and
    # a 20-step region -- 36 ms -- then read 1.85-1.92 ms per step instead of 1.80.)
    import gc
    gc.collect()
    gc.freeze()
    for i in range(args.warmup):
        warm_iterate()
    barrier()
    timed_out = int(agent.enc.persistent_status() != 0 or raised[0] or lib.vln_persistent_check() != 0)
    if world > 1:                                 # the fallback below contains collectives: every rank takes it or none does
        flag = torch.tensor([timed_out], device=dev, dtype=torch.int32)
        torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MAX)
        timed_out = int(flag.item())
    if timed_out:                                 # a bounded in-kernel wait timed out during warm-up: fall back
        print("[bench] persistent recurrence reported a timeout; using per-step launches", file=sys.stderr, flush=True)
        lib.vln_set_persistent(0)
        if use_graph:                             # the recorded iteration contains the persistent launches: record it again
            agent.capture(live.live)
        for i in range(max(1, args.warmup)):      # the warm-up again, on the path that will be timed
            iterate()
        barrier()
    marks = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        iterate()
        marks.append(time.perf_counter())
    barrier()
    dt = time.perf_counter() - t0
    if rank == 0 and len(marks) >= 10:      # host submit time per block of iterations (diagnostic, stderr only)
        q = max(1, len(marks) // 5)
        blk = [(marks[min(i + q, len(marks)) - 1] - (marks[i - 1] if i else t0)) / (min(i + q, len(marks)) - i) * 1e3
               for i in range(0, len(marks), q)]
        print("[bench] host submit ms/iter by block: " + " ".join(f"{b:.2f}" for b in blk), file=sys.stderr, flush=True)
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = float(tmax.item())
    ms_per_step = dt / args.steps * 1e3
    value = world * args.steps / dt
    if rank == 0:
        print(f"[bench] timed region: {ms_per_step:.3f} ms/step on {world} GPU(s)", file=sys.stderr, flush=True)

    roofline = None
    if not args.no_roofline:
        # EVERY rank replays the K steps (an iteration contains the gradient all-reduce: a collective only rank 0 entered
        # would never complete); the per-kernel timers are switched on and read on rank 0 only
        nk = 0
        while lib.vln_prof_kernel_name(nk):
            nk += 1
        if rank == 0:
            for k in range(nk):
                lib.vln_prof_enable(k, 1)
            read_prof(lib, nk)
        torch.cuda.synchronize()
        for _ in range(args.steps):
            iterate_eager()           # per-kernel event pairs ride on plain launches (a captured graph has none)
        torch.cuda.synchronize()
        rows = read_prof(lib, nk) if rank == 0 else []
        if rank == 0:
            for k in range(nk):
                lib.vln_prof_enable(k, 0)
        if rows:
            rows.sort(key=lambda r: -r["ms"])
            top = rows[0]
            ach = top["bytes"] / (top["ms"] * 1e-3) / 1e9
            traffic, mfma, pmc_note = pmc_figures(top["kernel"], args.dtype)
            roofline = dict(bound="hbm", kernel=top["kernel"], achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                            frac=round(ach / HBM_PEAK_GBS, 4), traffic=traffic, mfma_util=mfma, pmc=pmc_note,
                            avg_launch_us=round(top["ms"] * 1e3 / top["launches"], 2),
                            algo_bytes_per_launch=round(top["bytes"] / top["launches"]),
                            kernels=[dict(kernel=r["kernel"], launches_per_step=r["launches"] / args.steps,
                                          us_per_step=round(r["ms"] * 1e3 / args.steps, 1),
                                          GBps=round(r["bytes"] / (r["ms"] * 1e-3) / 1e9, 1)) for r in rows])
    if world > 1:
        torch.distributed.barrier()

    # Secondary, driver-timed numbers in the same line (never `value`): the fp32 path, the PCIe-inclusive path, the same
    # workload at 128 episodes per GPU (the iteration is bound by its dependent chain, not by bytes: twice the episodes cost
    # about a third more), and the two
    # other single-GPU workloads BASELINE.json configures (IL + A2C at the reference's episode cap 35; Self-Monitor B=128;
    # the Speaker-Follower agent of config 0 at a GPU batch).
    secondary = None
    if rank == 0 and world == 1 and not args.no_secondary:
        secondary = {}
        gc.unfreeze()
        short = [make_tape(args.batch, args.L, 3, 8, seed=5050 + k, n_rows=store.N) for k in range(4)]

        def per_step():      # marginal cost of one decoder step (forward + backward + its share of the weight gradients)
            if args.T <= 3:
                return {"error": "needs --T > 3"}
            a = secondary_envdrop(vln, dev, store, cpu_tapes, dtype, "store", args, graph=use_graph)
            b = secondary_envdrop(vln, dev, store, short, dtype, "store", args, graph=use_graph)
            return {"us": round((a - b) / (args.T - 3) * 1e3, 1), "how": f"(ms at T={args.T} - ms at T=3) / {args.T - 3}, same path as the headline"}

        def long_run(n=1000):      # the headline path over a region 50x the driver's: what a 20-step region cannot show (clock ramps, drift)
            if agent.graph is None:
                return {"error": "no iteration graph"}
            gc.collect(); gc.freeze()
            try:
                for _ in range(8):
                    iterate()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(n):
                    iterate()
                torch.cuda.synchronize()
                return {"ms_per_step": round((time.perf_counter() - t1) / n * 1e3, 3), "steps": n}
            finally:
                gc.unfreeze()

        for name, fn in (("headline_long_run", long_run),
                         ("eager_ms_per_step", lambda: secondary_envdrop(vln, dev, store, cpu_tapes, dtype, "store", args)),
                         ("dropin_unchanged_caller_ms_per_step", lambda: secondary_envdrop(vln, dev, store, cpu_tapes, dtype, "tensor", args, dropin=True)),
                         ("split_wgrad_ms_per_step", lambda: secondary_envdrop(vln, dev, store, cpu_tapes, dtype, "store", args, graph=use_graph, wgrad="split")),
                         ("all_bf16_weights_ms_per_step", lambda: secondary_envdrop(vln, dev, store, cpu_tapes, dtype, "store", args, graph=use_graph,
                                                                                    fp32_weights=())),
                         ("il_host_in_loop", lambda: secondary_host_in_loop(vln, dev, store, cpu_tapes, dtype, args)),
                         ("decoder_step_fwd_bwd", per_step),
                         ("phases", lambda: secondary_envdrop(vln, dev, store, cpu_tapes, dtype, "store", args, phases=True)),
                         ("fp32_ms_per_step", lambda: secondary_envdrop(vln, dev, store, cpu_tapes, torch.float32, "store", args, graph=use_graph)),
                         ("features_host_fp32_ms_per_step", lambda: secondary_envdrop(vln, dev, store, cpu_tapes, dtype, "host", args)),
                         ("batch128_ms_per_step", lambda: secondary_envdrop(
                             vln, dev, store, [make_tape(128, args.L, args.T, 8, seed=4040 + k, n_rows=store.N) for k in range(4)],
                             dtype, "store", args, graph=use_graph)),
                         # BASELINE config 3's per-rank iteration as 36 graph segments, the host reading every sampled action between them
                         # (envdrop.py:196-206); beside it the same iteration with the actions left on the device (round 3's form of the figure)
                         ("il_plus_a2c_T35", lambda: secondary_agents(dev, args, "a2c", store, read_actions="handshake")),
                         ("il_plus_a2c_T35_graph_per_step", lambda: secondary_agents(dev, args, "a2c", store, read_actions="poll")),
                         ("il_plus_a2c_T35_stream_sync_per_step", lambda: secondary_agents(dev, args, "a2c", store)),
                         ("il_plus_a2c_T35_actions_on_device", lambda: secondary_agents(dev, args, "a2c", store, read_actions=False)),
                         # BASELINE config 2 does not ask for bf16: the Self-Monitor's figure is the fp32 one; bf16 beside it
                         ("self_monitor_B128", lambda: secondary_agents(dev, args, "monitor", store, dtype="fp32")),
                         ("self_monitor_B128_bf16", lambda: secondary_agents(dev, args, "monitor", store, dtype="bf16")),
                         ("speaker_follower_B64", lambda: secondary_agents(dev, args, "follower", store, dtype="bf16")),
                         ("speaker_follower_B64_fp32", lambda: secondary_agents(dev, args, "follower", store, dtype="fp32")),
                         ("speaker_teacher_forcing_B64", lambda: secondary_agents(dev, args, "speaker", store, dtype="bf16")),
                         ("speaker_teacher_forcing_B64_fp32", lambda: secondary_agents(dev, args, "speaker", store, dtype="fp32"))):
            t1 = time.perf_counter()
            try:
                secondary[name] = fn()
            except Exception as e:          # a secondary number never takes the headline line down
                secondary[name] = {"error": f"{type(e).__name__}: {e}"[:300]}
            print(f"[bench] secondary {name}: {secondary[name]} ({time.perf_counter() - t1:.1f} s)", file=sys.stderr, flush=True)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        ncores = min(usable_cores(), 64)
        print(f"[bench] cpu baseline on {ncores} threads, then 1 ...", file=sys.stderr, flush=True)
        P_enc = {k: v.detach().float().cpu().clone() for k, v in agent.enc.state_dict().items()}
        P_dec = {k: v.detach().float().cpu().clone() for k, v in agent.dec.state_dict().items()}
        t_cpu = cpu_tapes[0]
        for s_ in t_cpu["steps"]:           # tape 0 with its explicit feature tensors, gathered from the same table rows
            m = s_ if "img" in s_ else materialize_step(s_, store.table)
            s_["img"], s_["cand"] = m["img"].float().cpu(), m["cand"].float().cpu()
        run = cpu_baseline(t_cpu, P_enc, P_dec)
        torch.set_num_threads(ncores)
        sec_n, done_n = run(2, args.cpu_iters, 14.0)
        torch.set_num_threads(1)
        sec_1, done_1 = run(1, 3, 14.0)
        torch.set_num_threads(ncores)
        cpu = dict(value=round(1.0 / sec_n, 4), unit="steps/s", cores=ncores, kind="port",
                   value_1thread=round(1.0 / sec_1, 4),
                   sample=f"tape 0 of the same workload (B={args.batch}, L={args.L}, T={args.T}), fp32, dropout on: {ncores} threads = "
                          f"median of {done_n} iterations after 2 warm-ups; 1 thread = median of {done_1} after 1 warm-up")

    if rank == 0:
        print(json.dumps({
            "metric": "agent train steps/sec (EnvDrop IL, batch 64/GPU, 36x2048 feats)", "value": round(value, 3),
            "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"envdrop_il_fwd_bwd_clip_rmsprop_B{args.batch}_L{args.L}_T{args.T}", "features": args.features,
                       "feature_table": f"{store.N}x36x2048 {args.dtype} resident in HBM", "episode_batches_rotated": len(tapes),
                       "batch_source": ({"push": "pinned host memory, sent one iteration ahead by an async H2D copy on a copy stream into a device ring slot; the iteration's first launch moves it into the live buffers (HostBatchFeed prefetch)",
                                         "pull": "pinned host memory, pulled by the iteration's first launch (vln_host_fetch)",
                                         "copy": "pinned host memory, one hipMemcpyAsync H2D per iteration",
                                         "device": "device memory, one D2D copy per iteration"}[args.batch_source] if live is not None else "per-step tensors"),
                       "global_batch": args.batch * world, "seq_len": args.L, "decoder_steps": args.T,
                       "parallelism": f"dp{world}", "world_size": world,
                       "iteration_graph": ("3 segments + host-issued gradient exchange" if agent.segmented else ("one graph, gradient exchange captured inside" if (args.dp_capture and (world > 1 or args.dp_path)) else True)) if use_graph else False,
                       "decoder_fp32_weights": sorted(agent.dec.fp32_weights), "chained_steps": bool(agent.dec.chain_steps), "projected_context": bool(agent.dec.last_projected), "decoder_wgrad_ride": (vln.ops.GradRide.stats() if agent.dec.ride_wgrads else False), "prologue_launch": bool(agent.use_prologue and use_graph), "batch_tail_under_recurrence": bool(agent.split_pull and agent.batch_feed is not None and agent.ride_gather), "decoder_shadows_under_recurrence": bool(agent.ride_shadows and agent.ride_gather and agent.use_prologue and use_graph), "gather": "recurrence passengers" if agent.ride_gather else ("rollout launch" if agent.rollout_gather else "per step"),
                       "wgrad": vln.ops.get_wgrad_precision(),
                       "backend": (args.backend + ("=rccl" if args.backend == "nccl" else "")) if (world > 1 or args.dp_path) else None},
            "roofline": roofline, "cpu_baseline": cpu, "secondary": secondary}))
    if world > 1 or args.dp_path:
        torch.distributed.destroy_process_group()


def csrc_sha():
    """Hash of the kernel sources: PMC figures are only quoted for the code they were measured on."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "curriculum-learning-for-vln_amd", "csrc", "*.h*"))):
        if f.endswith((".hip", ".h")):
            h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def pmc_figures(kernel, dtype):
    """(HBM-side bytes per launch, MFMA issue-slot utilisation, note) of `kernel` from the committed rocprofv3 --pmc passes
    (profiles/round5_pmc.json, written by scripts/pmc_stamp.py from separate FETCH_SIZE / WRITE_SIZE / MFMA_BUSY runs of this
    same command).  The file carries the hash of the kernel sources it was taken on: a mismatch means the numbers describe
    OTHER code, and they are refused (null) rather than quoted stale."""
    f = os.path.join(ROOT, "profiles", "round5_pmc.json")
    if not os.path.exists(f):
        return None, None, "no PMC passes committed for this round yet"
    d = json.load(open(f))
    if d.get("csrc_sha") != csrc_sha():
        return None, None, f"profiles/round5_pmc.json was taken on kernel sources {d.get('csrc_sha')}, this is {csrc_sha()}: refused"
    t = d.get("traffic", {}).get(dtype, {}).get(kernel)
    m = d.get("mfma_util", {}).get(dtype, {}).get(kernel)
    return (t["bytes_per_launch"] if t else None), m, f"profiles/round5_pmc.json, kernel sources {d['csrc_sha']}"


def secondary_envdrop(vln, dev, store, cpu_tapes, dtype, features, args, steps=20, warmup=6, graph=False, dropin=False,
                      wgrad=None, phases=False, fp32_weights=None):
    """ms per iteration of the headline workload under another precision / feature path / caller (own agent, own arena).
    graph: the whole iteration as one hipGraph (store features only).  dropin: the reference's UNCHANGED caller -- feature
    tensors handed in every step, `logits.masked_fill_` + per-step cross entropy (envdrop.py:173-179), no arena, no deferred
    logits; only the fused clip + RMSprop is kept.  phases: instead of the ms, GPU microseconds per phase of an eager iteration
    (hip events on the stream: encoder forward, the decoder steps' forward, loss, backward, clip + optimizer)."""
    torch.manual_seed(2020)
    prev_w = vln.ops.get_wgrad_precision()
    if wgrad is not None:
        vln.ops.set_wgrad_precision(wgrad)
    try:
        ag = GpuAgent(vln, dev, dtype, 1, arena=not dropin, rollout_ce=not dropin)
        ag.clear_grads_in_step = True
        ag.ride_gather = features == "store" and args.ride_gather != "off" and not args.rollout_gather
        if fp32_weights is not None:       # None: the module's default (the two attention query projections in fp32)
            ag.dec.fp32_weights = frozenset(fp32_weights)
        if features == "store":
            st = store if store.table.dtype == dtype else vln.DeviceFeatureStore(store.table.to(dtype), device=dev, dtype=dtype)
            tapes = [tape_to(t, dev, store=st) for t in cpu_tapes]
            live = LiveBatch(tapes, source=args.batch_source)
            ag.use_live(live)
            get = live.load
        else:                    # host: pinned host fp32 features (what the reference's ImageFeatures holds); tensor: device tensors
            tapes = []
            for t in cpu_tapes[:2]:
                t = dict(t, steps=[dict(s) for s in t["steps"]])
                for s in t["steps"]:
                    s.update(materialize_step(s, store.table))
                tapes.append(tape_to(t, dev, host_dtype=torch.float32) if features == "host" else tape_to(t, dev))
            get = lambda k: tapes[k % len(tapes)]
        if graph:
            ag.use_clock(st)
        for k in range(4):
            ag.iteration(get(k))
        torch.cuda.synchronize()
        if phases:
            return _phase_times(ag, get, steps, len(cpu_tapes[0]["steps"]))
        if graph:
            ag.capture(live.live)
            run = lambda k: (live.load(k), ag.replay())
        else:
            run = lambda k: ag.iteration(get(k))
        for k in range(warmup):
            run(k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            run(k)
        torch.cuda.synchronize()
        return round((time.perf_counter() - t0) / steps * 1e3, 3)
    finally:
        vln.ops.set_wgrad_precision(prev_w)


class LiveSteps:
    """Host-in-the-loop marshalling: like LiveBatch, but the per-STEP inputs (viewpoint rows, view / candidate indices, candidate
    mask, angle feature of the previous action, teacher action) live in one pinned host blob PER STEP and are copied to that step's
    fixed device buffers only when the step is about to run -- the shape of the reference's rollout, whose every step marshals the
    simulator's new observation on the host (agent/base.py:141-178) after the previous action has reached it (envdrop.py:198-204)."""

    def __init__(self, tapes, dev):
        t0 = tapes[0]
        T = len(t0["steps"])
        top = [(k, t0[k]) for k in LiveBatch.TOP]
        self.top_layout, self.top_bytes = self._layout(top)
        self.step_layout, self.step_bytes = self._layout([(k, t0["steps"][0][k]) for k in LiveBatch.STEP])
        self.top_host = [self._pack(self.top_layout, self.top_bytes, [(k, tp[k]) for k in LiveBatch.TOP]) for tp in tapes]
        self.step_host = [[self._pack(self.step_layout, self.step_bytes, [(k, s[k]) for k in LiveBatch.STEP]) for s in tp["steps"]]
                          for tp in tapes]
        self.top_dev = torch.zeros(self.top_bytes, dtype=torch.uint8, device=dev)
        self.step_dev = [torch.zeros(self.step_bytes, dtype=torch.uint8, device=dev) for _ in range(T)]
        self.live = {k: v for k, v in t0.items() if k not in LiveBatch.TOP + ("steps",)}
        self.live.update(self._views(self.top_layout, self.top_dev))
        self.live["steps"] = [self._views(self.step_layout, b) for b in self.step_dev]
        # what the fake environment keeps on the host: every step's teacher actions, to be compared with what the agent sent
        self.host_targets = [[s["target"].cpu().numpy() for s in tp["steps"]] for tp in tapes]

    @staticmethod
    def _layout(items):
        out, off = [], 0
        for name, t in items:
            n = t.numel() * t.element_size()
            out.append((name, off, n, t.dtype, tuple(t.shape)))
            off = (off + n + 15) & ~15
        return out, off

    @staticmethod
    def _pack(layout, nbytes, items):
        blob = torch.zeros(nbytes, dtype=torch.uint8)
        for (name, o, n, dt, shape), (name2, t) in zip(layout, items):
            if name != name2 or tuple(t.shape) != shape or t.dtype != dt:
                raise ValueError(f"LiveSteps: tape layouts differ at {name}")
            blob[o:o + n] = t.detach().cpu().contiguous().view(-1).view(torch.uint8)
        return blob.pin_memory()

    @staticmethod
    def _views(layout, blob):
        return {name: blob[o:o + n].view(dt).view(shape) for name, o, n, dt, shape in layout}

    def load_top(self, k):
        self.top_dev.copy_(self.top_host[k % len(self.top_host)], non_blocking=True)
        return self.live

    def load_step(self, k, t):
        self.step_dev[t].copy_(self.step_host[k % len(self.step_host)][t], non_blocking=True)
        return self.live["steps"][t]


def secondary_host_in_loop(vln, dev, store, cpu_tapes, dtype, args, steps=20, warmup=6):
    """ms per iteration of the headline workload with THE HOST IN THE LOOP, in the reference's loop shape (envdrop.py:151-220):
    per decoder step the step's index vectors arrive by a pinned H2D copy (the observation the simulator just produced,
    base.py:141-178), the step runs (per-step hipGraph, candidate logits formed in the step, CE term per step), the chosen action
    a_t goes back to the host (`a_t.cpu()`, envdrop.py:198: one D2H + stream synchronisation per step) and a fake environment
    steps on it (checks the action against its own teacher tape, picks the next observation's blob).  Features stay in the
    resident table; the iteration is eager launches + per-step graphs: nothing of it can be captured whole."""
    import numpy as np
    torch.manual_seed(2020)
    ag = GpuAgent(vln, dev, dtype, 1, arena=True, rollout_ce=False)
    ag.clear_grads_in_step = True
    st = store if store.table.dtype == dtype else vln.DeviceFeatureStore(store.table.to(dtype), device=dev, dtype=dtype)
    tapes = [tape_to(t, dev, store=st) for t in cpu_tapes]
    ls = LiveSteps(tapes, dev)
    B = tapes[0]["B"]
    mismatches = [0]

    def iteration(k):
        vln.ops.set_arena(ag.arena); ag.arena.begin()
        try:
            tape = ls.load_top(k)
            ag.opt.zero_grad()
            ctx, h_t, c_t = ag.enc(tape["tokens"], tape["lengths32"])
            h_tilde = h_t
            terms = []
            for t in range(len(tape["steps"])):
                s = ls.load_step(k, t)                                      # this step's observation: pinned host -> device
                logits, (h_t, c_t), h_tilde = ag.dec(s["angle"], None, None, h_tilde, h_t, c_t, ctx, tape["seq_mask"],
                                                     gather=(st, s["rows"], s["vidx"], s["crow"], s["cview"], s["chead"], s["celev"]))
                terms.append(vln.losses.masked_cross_entropy(logits, s["target"], s["cand_mask"], "sum"))
                a_t = s["target"]                                           # teacher forcing: a_t = target (envdrop.py:183)
                cpu_a_t = a_t.cpu().numpy()                                 # envdrop.py:198: the action reaches the simulator
                # fake environment: episodes whose action is -1 have ended (envdrop.py:199-203); it answers with the next blob
                mismatches[0] += int((cpu_a_t != ls.host_targets[k % len(tapes)][t]).sum())
            loss = torch.stack(terms).sum() * (ML_WEIGHT / B)
            loss.backward()
            ag.opt.step(zero_grads=True)
            return loss
        finally:
            vln.ops.set_arena(None)

    for k in range(4 + warmup):
        iteration(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        iteration(k)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    return {"ms_per_step": round(ms, 3), "per_step_host_round_trips": len(tapes[0]["steps"]), "action_mismatches": mismatches[0],
            "how": "per step: pinned H2D of the step's index vectors, decoder step (per-step hipGraph, logits in the step, CE per step), "
                   "D2H of a_t + fake-env host step; features from the resident table"}


def _phase_times(ag, get, steps, n_dec_steps):
    """GPU time per phase of an eager iteration (per-decoder-step fwd / bwd and the optimizer reported apart): hip
    events recorded on the stream between the phases; the host runs ahead, so the differences are device time."""
    marks = {}

    def mark(name):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        marks.setdefault(name, []).append(ev)

    enc_fwd, dec_call, opt_step, opt_zero = ag.enc.forward, ag.dec.forward, ag.opt.step, ag.opt.zero_grad
    state = {"t": 0}
    T = n_dec_steps

    def enc_w(*a, **k):
        mark("start")
        out = enc_fwd(*a, **k)
        mark("enc_done")
        state["t"] = 0
        return out

    def dec_w(*a, **k):
        out = dec_call(*a, **k)
        state["t"] += 1
        if state["t"] == T:
            mark("dec_done")
        return out

    def opt_w(*a, **k):
        mark("bwd_done")
        out = opt_step(*a, **k)
        mark("opt_done")
        return out

    hook = ag.dec.grads_ready_hook

    def hook_w():                        # fires when the decoder's weight gradients have been issued, before the encoder's backward
        mark("dec_bwd_done")
        if hook is not None:
            hook()

    ag.enc.forward, ag.dec.forward, ag.opt.step, ag.dec.grads_ready_hook = enc_w, dec_w, opt_w, hook_w
    try:
        for k in range(steps + 3):
            ag.iteration(get(k))
    finally:
        ag.enc.forward, ag.dec.forward, ag.opt.step, ag.dec.grads_ready_hook = enc_fwd, dec_call, opt_step, hook
    torch.cuda.synchronize()

    def span(a, b):
        v = sorted(x.elapsed_time(y) * 1e3 for x, y in list(zip(marks[a], marks[b]))[3:])
        return round(v[len(v) // 2], 1)

    return {"encoder_fwd_us": span("start", "enc_done"), "decoder_fwd_us_per_step": round(span("enc_done", "dec_done") / T, 1),
            "loss_and_backward_us": span("dec_done", "bwd_done"), "clip_and_optimizer_us": span("bwd_done", "opt_done"),
            "decoder_bwd_us_per_step": round(span("dec_done", "dec_bwd_done") / T, 1),
            "encoder_bwd_us": span("dec_bwd_done", "bwd_done"),
            "decoder_steps": T, "note": "eager launches; backward = loss + decoder steps + encoder BPTT + weight gradients; "
                                        "decoder_bwd per step includes 1/T of the rollout loss, the logit branch and the decoder's weight gradients"}


def secondary_agents(dev, args, which, store, dtype=None, read_actions=True):
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import bench_agents as W
    # warm-up: the first iterations of a workload in a process grow the allocator's pools and load its kernels' code objects;
    # with 8 of them the Self-Monitor number read 5.7 ms against 5.05 ms for a second run in the same process
    W.configure(steps=20, warmup=30, dtype=dtype or args.dtype, arena=False, device=dev)
    W.args.roofline = bool(which == "a2c" and read_actions == "handshake")     # the cfg3 entry carries its own roofline block
    W.vln.functional.set_grad_in_place(True)
    W.vln.functional.set_rollout_wgrads(which in ("monitor", "follower"))     # parameter gradients once per rollout (functional.RolloutWgrads)
    import gc
    gc.collect()
    gc.freeze()                         # the bench's own objects (agent, tapes, store) out of the cyclic collector's way, as in the timed loop
    try:
        r = W.run_a2c(T_rl=35, store=store, read_actions=read_actions) if which == "a2c" else (W.run_follower() if which == "follower" else
                                                                    (W.run_speaker() if which == "speaker" else W.run_monitor()))
    finally:
        W.vln.functional.set_rollout_wgrads(False)
        W.vln.functional.set_grad_in_place(False)
        gc.unfreeze()
    out = {"workload": r["workload"], "ms_per_iteration": r["ms_per_iteration"], "dtype": r.get("dtype")}
    for k in ("iteration", "per_step_action_read", "roofline"):       # a2c: how the iteration was issued, that the host read every sampled action, its dominant kernel
        if k in r:
            out[k] = r[k]
    return out


if __name__ == "__main__":
    main()
