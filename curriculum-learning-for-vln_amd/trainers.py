"""Training ITERATIONS of the R2R agents on the HIP modules -- the reference's loop bodies (engine/trainer.py:405-427 EnvDrop,
:207-240 Self-Monitor, :60-110 Speaker-Follower; agent/speaker.py:75-87 the speaker) as objects a trainer drives:

    EnvDropILIteration      teacher-forced EnvDrop rollout + loss + backward + clip + RMSprop   (BASELINE config 1: the headline)
    EnvDropA2CIteration     IL rollout + SAMPLED rollout with the critic, mixed loss            (config 3 / 4, one rank's share)
    SelfMonitorIteration    co-grounding decoder + progress monitor, one Adam                   (config 2)
    FollowerIteration       Speaker-Follower agent, two Adam                                    (config 0's model)
    SpeakerIteration        the speaker's teacher-forcing iteration, clip + two Adam            (config 4's back-translation model)

Every class offers the same three calls: `iteration(...)` issues the launches eagerly, `capture(...)` records the same
iteration as ONE hipGraph (graphs.IterationGraph; the A2C iteration with its per-step host turns as in-graph waits,
graphs.HandshakeIterationGraph, or as graph segments), `replay()` runs it on whatever the fixed-address batch buffers hold.
Results of the three forms are bit-identical (tests/test_hip_graphs.py).  The simulator, the curriculum samplers and the
episode ordering stay with the caller: an iteration takes the marshalled batch (batches.LiveBatch / LiveSteps or the caller's own
tensors with the same keys) and, where the rollout needs the host between steps, a `host_turn(t, actions)` callback.

`bench.py` and `scripts/bench_agents.py` time exactly these objects; tests/test_hip_headline_vs_oracle.py, test_hip_cfg3_cfg4.py
and test_hip_graphs.py hold them to the CPU oracle and to each other."""
from __future__ import annotations

import contextlib
import ctypes as C
import time

import torch

from . import _lib, dp, functional, losses, ops, optim
from .decoders import AttnDecoderLSTM, MonitorDecoder
from .encoder import EncoderLSTM
from .envdrop_decoder import Critic, EnvDropDecoder
from .graphs import HandshakeIterationGraph, IterationGraph, SegmentedIterationGraph
from .runtime import DeviceClock
from .speaker import Speaker, SpeakerDecoder, SpeakerEncoder

ML_WEIGHT = 0.2            # configs/envdrop/envdrop_config.yaml:45
CLIP = 40.0                # trainer.py:425-426
LR = 1e-4                  # envdrop_config.yaml:19
HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def time_iterations(fn, steps, warmup=0):
    """ms per call of `fn` over `steps` back-to-back calls after `warmup` untimed ones (device synchronised on both sides)."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def read_kernel_timers(lib=None):
    """Drain the library's per-kernel HIP-event timers (vln_prof_*): [{kernel, launches, ms, bytes}] (bytes = the ALGORITHMIC
    bytes the launches were issued with)."""
    lib = lib or _lib.load()
    rows, k = [], 0
    while lib.vln_prof_kernel_name(k):
        n, ms, by = C.c_int64(), C.c_double(), C.c_double()
        lib.vln_prof_read(k, C.byref(n), C.byref(ms), C.byref(by))
        if n.value:
            rows.append(dict(kernel=lib.vln_prof_kernel_name(k).decode(), launches=n.value, ms=ms.value, bytes=by.value))
        k += 1
    return rows


def kernel_roofline(fn, n_iterations, per="step", top=None, extra=None):
    """Run `fn` (EAGER launches: a replayed graph carries no event pairs) `n_iterations` times with the per-kernel timers on and
    return the `roofline` block of a bench line for the kernel with the largest total time: achieved = algorithmic bytes per launch
    / average launch duration against the HBM peak."""
    lib = _lib.load()
    nk = 0
    while lib.vln_prof_kernel_name(nk):
        nk += 1
    for k in range(nk):
        lib.vln_prof_enable(k, 1)
    read_kernel_timers(lib)
    torch.cuda.synchronize()
    try:
        for _ in range(n_iterations):
            fn()
        torch.cuda.synchronize()
        rows = sorted(read_kernel_timers(lib), key=lambda r: -r["ms"])
    finally:
        for k in range(nk):
            lib.vln_prof_enable(k, 0)
    if not rows:
        return None
    t = rows[0]
    ach = t["bytes"] / (t["ms"] * 1e-3) / 1e9
    out = dict(bound="hbm", kernel=t["kernel"], achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 4),
               traffic=None, avg_launch_us=round(t["ms"] * 1e3 / t["launches"], 2), algo_bytes_per_launch=round(t["bytes"] / t["launches"]))
    if extra:
        out.update(extra)
    out["kernels"] = [{"kernel": r["kernel"], f"launches_per_{per}": r["launches"] / n_iterations,
                       f"us_per_{per}": round(r["ms"] * 1e3 / n_iterations, 1), "GBps": round(r["bytes"] / (r["ms"] * 1e-3) / 1e9, 1)}
                      for r in (rows if top is None else rows[:top])]
    return out


@contextlib.contextmanager
def rollout_wgrads(on=True, in_place=True):
    """Parameter gradients of the fused decoder nodes written in place and formed ONCE per rollout (functional.RolloutWgrads) for
    the iterations issued inside the block; the module-level switches are restored on the way out."""
    functional.set_grad_in_place(in_place)
    functional.set_rollout_wgrads(on)
    try:
        yield
    finally:
        functional.set_rollout_wgrads(False)
        functional.set_grad_in_place(False)


class EnvDropILIteration:
    """One EnvDrop IMITATION-LEARNING training iteration (the reference's loop body, engine/trainer.py:405-427 with
    `feedback="teacher"`; rollout: agent/envdrop.py:86-278): instruction encoder, T teacher-forced decoder steps with the
    in-place candidate mask + cross entropy (envdrop.py:151-179), `ml_loss * ML_WEIGHT / B` (envdrop.py:268), backward,
    gradient all-reduce (world > 1), clip-norm 40 per module (trainer.py:425-426), RMSprop (trainer.py:380-381).

        it = EnvDropILIteration(device, torch.bfloat16)               # builds EncoderLSTM + EnvDropDecoder + FusedRMSprop
        it = EnvDropILIteration(device, dtype, enc=my_enc, dec=my_dec)  # or wraps modules a checkpoint was loaded into
        loss = it.iteration(batch)                                    # eager launches (per-step hipGraphs with arena=True)
        it.use_live(live); it.use_clock(store); it.capture(live.live) # the whole iteration as ONE hipGraph ...
        live.load(k); loss = it.replay()                              # ... replayed on whatever batch k holds

    `batch` is a tape (synthetic.make_tape / tape_to, or a trainer's own marshalling into the same keys): `tokens`,
    `lengths32`, `seq_mask`, `B`, `steps` = per step `angle`, `target`, `cand_mask` and either explicit `img` / `cand` feature
    tensors, pinned `img_host` / `cand_host`, or -- with `store` (a staging.DeviceFeatureStore) -- the index vectors `rows`,
    `vidx`, `crow`, `cview`, `chead`, `celev`.  Options (attributes, all A/B-measured in profiles/round*_notes.md):
    `ride_gather` (the rollout's feature gather as passengers of the encoder's recurrence launch), `dec.ride_wgrads`,
    `use_prologue`, `split_pull`, `segmented` (the data-parallel three-segment form), `clear_grads_in_step`."""

    def __init__(self, dev, dtype, world=1, arena=False, rollout_ce=True, side_gather=False, fused_gather=True, *, enc=None, dec=None,
                 vocab=992, embed=256, hidden=512, feature_size=2176, angle_size=128, action_embed=64, drop=0.5, feat_drop=0.3,
                 lr=LR, clip_norm=CLIP, ml_weight=ML_WEIGHT):
        self.world, self.dtype, self.ml_weight = world, dtype, ml_weight
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
        self.enc = enc if enc is not None else EncoderLSTM(vocab, embed, hidden, 0, drop, True, 1, compute_dtype=dtype).to(dev)
        self.dec = dec if dec is not None else EnvDropDecoder(hidden, drop, feat_drop, action_embed, angle_size, feature_size, compute_dtype=dtype).to(dev)
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
        self.opt = optim.FusedRMSprop([list(self.enc.parameters()), list(self.dec.parameters())], lr=lr, clip_norm=clip_norm)
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
            lib = _lib.load()
            _lib.check(lib.vln_debug_trivial_chain(self._probe_buf[0].data_ptr(), self._probe_buf[1].data_ptr(), 64 * 512, n, 256,
                                                            _lib.raw_stream()), "vln_debug_trivial_chain")

    def use_clock(self, store=None):
        self.clock = DeviceClock(next(self.enc.parameters()).device)
        self.clock.attach(self.enc, self.dec)
        if store is not None:
            self.clock.attach(store)
        return self.clock

    def capture(self, tape):
        """Record one iteration over `tape` (buffers at fixed addresses: LiveBatch.live) as ONE hipGraph; `replay()` then runs
        an iteration on whatever those buffers hold."""
        if self.clock is None:
            raise RuntimeError("EnvDropILIteration.capture: use_clock() first (a captured iteration reads its dropout offsets from device words)")
        if self.segmented:
            self.graph = SegmentedIterationGraph(self.segments(tape), self.clock).capture()
            return self.graph
        self.graph = IterationGraph(lambda: self.iteration(tape), self.clock).capture(
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
        self.arena = ops.RolloutArena() if on else None
        self.dec.step_graphs = True if on else bool(getattr(type(self.dec), "default_step_graphs", False))

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
        ops.set_arena(self.arena)
        if self.arena is not None:
            self.arena.begin()
        try:
            return self._iteration(tape)
        finally:
            ops.set_arena(None)

    def segments(self, tape):
        """The iteration cut at its two exchange points (trainer.py:421-427 with the gradient all-reduce in it):
        [graph A: forward, loss, the decoder's backward] [host: start the decoder slice's all-reduce] [graph B: the encoder's
        backward] [host: reduce the rest, wait] [graph C: clip + update]."""
        def in_arena(fn, begin=False):
            def run():
                ops.set_arena(self.arena)
                if begin and self.arena is not None:
                    self.arena.begin()
                try:
                    return fn()
                finally:
                    ops.set_arena(None)
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
            self._cut = dp.BackwardCut()
            ctx, h_t, c_t = self._cut.at(ctx, h_t, c_t)
        h_tilde = h_t
        terms = []
        ce = losses.RolloutCE() if self.rollout_ce else None
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
                terms.append(losses.masked_cross_entropy(logits, s["target"], s["cand_mask"], "sum"))
        w = self.ml_weight / (B * self.world)                         # envdrop.py:268; global batch normalisation under DP
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


class _GraphedIteration:
    """Shared plumbing of the teacher-forced agents below: the batch at fixed addresses (`load`), the eager iteration, its capture
    as ONE hipGraph over a runtime.DeviceClock, the replay."""

    clock = None
    graph = None
    live = None

    def _make_clock(self, dev, graph, *modules):
        self.clock = DeviceClock(dev).attach(*modules) if graph else None
        return self.clock

    def load(self, batch):
        """Copy `batch` (same keys, shapes and dtypes as the first one) into the fixed-address buffers the iteration reads."""
        if self.live is None:
            self.live = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items() if k != "steps"}
            self.live["steps"] = [{k: v.clone() for k, v in s.items()} for s in batch["steps"]]
            return self.live
        for k, v in batch.items():
            if k != "steps" and torch.is_tensor(v):
                self.live[k].copy_(v, non_blocking=True)
        for ls, bs in zip(self.live["steps"], batch["steps"]):
            for k in ls:
                ls[k].copy_(bs[k], non_blocking=True)
        return self.live

    def iteration(self, batch=None):
        if batch is not None and batch is not self.live:
            self.load(batch)
        with rollout_wgrads(self.rollout_wgrads):
            return self._iteration()

    def capture(self, warmup=3):
        if self.clock is None:
            raise RuntimeError(f"{type(self).__name__}.capture: built with graph=False (no device clock)")
        with rollout_wgrads(self.rollout_wgrads):
            for _ in range(warmup):
                self._iteration()
            self.graph = IterationGraph(self._iteration, self.clock).capture()
        return self.graph

    def replay(self):
        return self.graph.replay()


class SelfMonitorIteration(_GraphedIteration):
    """The Self-Monitoring agent's training iteration (BASELINE config 2; agent/monitor.py:66-196, engine/trainer.py:207-240):
    uni-directional encoder, T co-grounding decoder steps (policy.py:108-166, BN-MLP with train-mode statistics), the step loss
    of monitor.py:146-165 (CE at t = 0, then lam * MSE(progress) + (1 - lam) * CE, the progress target formed on the device) in
    one launch each way, one Adam over encoder + decoder (trainer.py:219-222).

    batch: `tokens` [B, L], `lens32` [B], `steps` = per step `cand` [B, C, F], `cmask`, `target`, `start`, `cur`, `ended`.
    `a_prev0`: the first step's previous-action rows (default: the reference's zeros, monitor.py:108)."""

    def __init__(self, dev, dtype, *, enc=None, dec=None, opt=None, vocab=992, embed=256, hidden=512, max_len=80, mlp=(1024,),
                 feature_size=2176, drop=0.5, lr=1e-4, lam=0.5, graph=True, rollout_wgrads=True, a_prev0=None, rollout_loss=True):
        self.dev, self.dtype, self.lam, self.rollout_wgrads, self.rollout_loss = dev, dtype, lam, rollout_wgrads, rollout_loss
        # MLP_HIDDEN (1024,): configs/monitor/selfmonitor_config.yaml:45
        self.enc = enc if enc is not None else EncoderLSTM(vocab, embed, hidden, 0, drop, False, 1, compute_dtype=dtype).to(dev).train()
        self.dec = dec if dec is not None else MonitorDecoder(hidden, drop, max_len, tuple(mlp), feature_size, feature_size, compute_dtype=dtype).to(dev).train()
        self.opt = opt if opt is not None else optim.FusedAdam([list(self.enc.parameters()) + list(self.dec.parameters())], lr=lr)
        self.a_prev0 = a_prev0
        if self._make_clock(dev, graph and getattr(self.dec, "c_step", True), self.enc, self.dec) is not None:
            self.opt.use_clock(self.clock)

    def _iteration(self):
        b = self.live
        if self.clock is not None:
            self.clock.tick()
        self.opt.zero_grad()
        ctx, h, c = self.enc(b["tokens"], b["lens32"])
        seq_mask = b["seq_mask"] if "seq_mask" in b else b["tokens"] == 0
        B = b["tokens"].shape[0]
        if self.a_prev0 is None:
            self.a_prev0 = torch.zeros(B, self.dec.feature_size if hasattr(self.dec, "feature_size") else b["steps"][0]["cand"].shape[-1], device=self.dev)
        a_prev, loss = self.a_prev0, 0.0
        rows = self._rows(B)
        rl = losses.RolloutMonitorLoss(self.lam) if self.rollout_loss else None      # every step's loss in ONE launch each way (round 6)
        # teacher forcing: every step's previous-action row is a function of the batch -- one launch up front instead of one per step
        nxt = ops.select_rows_multi([s["cand"] for s in b["steps"][:-1]], [s["target"] for s in b["steps"][:-1]]) if (self.rollout_loss and len(b["steps"]) > 1) else None
        for t, s in enumerate(b["steps"]):
            (logit, prog), (h, c), _ = self.dec(None, a_prev, s["cand"], h, c, ctx, seq_mask, s["cmask"])
            if rl is not None:
                rl.add(logit, s["target"], s["cmask"], prog, s["start"], s["cur"], s["ended"])
            else:
                # monitor.py:146-165 in one launch each way (CE + progress target + MSE + the lambda mix)
                loss_t, _ = losses.monitor_mixed_loss(logit, s["target"], s["cmask"], prog, s["start"], s["cur"], s["ended"], t, self.lam)
                loss = loss + loss_t
            if nxt is None:
                a_prev = s["cand"][rows, s["target"]].detach()                  # monitor.py:191
            elif t < len(nxt):
                a_prev = nxt[t]
        if rl is not None:
            loss = rl.sum()
        loss.backward()
        self.opt.step()
        return loss

    def _rows(self, B):
        if getattr(self, "_arange", None) is None or self._arange.numel() != B:
            self._arange = torch.arange(B, device=self.dev)
        return self._arange


class FollowerIteration(_GraphedIteration):
    """The Speaker-Follower agent's training iteration (BASELINE config 0's model; agent/follower.py:66-170, trainer.py:60-110):
    2-layer bi-directional encoder (E 300, H 256), AttnDecoderLSTM over 36 x 2176 views (policy.py:37-60), CE mean per step
    (follower.py:62,123-139), two Adam instances (trainer.py:65-67).

    batch: `tokens`, `lens32`, `steps` = per step `img` [B, 36, F], `cand` [B, C, F], `cmask`, `target`."""

    def __init__(self, dev, dtype, *, enc=None, dec=None, vocab=992, embed=300, hidden=256, feature_size=2176, drop=0.5, lr=1e-4,
                 graph=True, rollout_wgrads=True, fused=True, rollout_ce=True):
        self.dev, self.dtype, self.rollout_wgrads, self.rollout_ce = dev, dtype, rollout_wgrads, rollout_ce
        self.hoist_batch_only_work = rollout_ce        # previous-action rows and candidate projections of all steps up front (teacher forcing)
        self.enc = enc if enc is not None else EncoderLSTM(vocab, embed, hidden, 0, drop, True, 2, compute_dtype=dtype).to(dev).train()
        self.dec = dec if dec is not None else AttnDecoderLSTM(hidden, drop, feature_size, feature_size, compute_dtype=dtype).to(dev).train()
        self.dec.fused_step = fused
        self.opt_e = optim.FusedAdam([list(self.enc.parameters())], lr=lr)
        self.opt_d = optim.FusedAdam([list(self.dec.parameters())], lr=lr)
        if self._make_clock(dev, graph and fused and getattr(self.dec, "c_step", True), self.enc, self.dec) is not None:
            self.opt_e.use_clock(self.clock); self.opt_d.use_clock(self.clock)
        self._a0 = None

    def load(self, batch):
        """As _GraphedIteration.load; the steps' candidate tensors become slices of ONE [T, B, C, F] buffer (`cand_all`) when they
        share a shape, so that one product can project them all (AttnDecoderLSTM.project_candidates)."""
        first = self.live is None
        live = super().load(batch)
        if first:
            cs = [s["cand"] for s in live["steps"]]
            if all(c.shape == cs[0].shape and c.dtype == torch.float32 for c in cs):
                allc = torch.stack(cs, 0).contiguous()
                for t, s in enumerate(live["steps"]):
                    s["cand"] = allc[t]
                live["cand_all"] = allc
        return live

    def _iteration(self):
        b = self.live
        if self.clock is not None:
            self.clock.tick()
        self.opt_e.zero_grad(); self.opt_d.zero_grad()
        ctx, h, c = self.enc(b["tokens"], b["lens32"])
        seq_mask = b["seq_mask"] if "seq_mask" in b else b["tokens"] == 0
        B, F = b["tokens"].shape[0], b["steps"][0]["cand"].shape[-1]
        if self._a0 is None:
            self._a0, self._arange = torch.zeros(B, F, device=self.dev), torch.arange(B, device=self.dev)
        a_prev, loss = self._a0, 0.0                                             # follower.py:101: zeros
        ce = losses.RolloutCE() if self.rollout_ce else None                     # every step's mean CE in ONE launch each way (round 6)
        # ... and every step's previous-action row (a function of the batch under teacher forcing) in one launch up front
        nxt = ops.select_rows_multi([s["cand"] for s in b["steps"][:-1]], [s["target"] for s in b["steps"][:-1]]) if (self.hoist_batch_only_work and len(b["steps"]) > 1) else None
        # ... and the candidates' projection of ALL steps in one product (it depends on the batch only; `load` keeps the steps' candidate
        # tensors as slices of one buffer)
        pre = self.dec.project_candidates(b["cand_all"]) if (self.hoist_batch_only_work and "cand_all" in b and getattr(self.dec, "c_step", False) and self.dec.fused_step) else None
        for t, s in enumerate(b["steps"]):
            logit, (h, c), _ = self.dec(s["img"], a_prev, s["cand"], h, c, ctx, seq_mask, cand_context=None if pre is None else pre[t])
            if ce is not None:
                ce.add(logit, s["target"], s["cmask"])
            else:
                loss = loss + losses.masked_cross_entropy(logit, s["target"], s["cmask"], "mean")
            if nxt is None:
                a_prev = s["cand"][self._arange, s["target"]].detach()          # follower.py:164
            elif t < len(nxt):
                a_prev = nxt[t]
        if ce is not None:
            loss = ce.mean_per_step()
        loss.backward()
        self.opt_e.step(); self.opt_d.step()
        return loss


class SpeakerIteration(_GraphedIteration):
    """The speaker's training iteration (agent/speaker.py:75-87: teacher_forcing -> backward -> clip 40 per module -> two Adam) at
    the configured size: RNN_DIM 512, bidirectional encoder over paths of up to 7 viewpoints x 36 x 2176 views, WEMB 256,
    vocabulary 992, 80-token instructions, DROPOUT 0.6 / FEAT_DROPOUT 0.3.

    batch: `can` [B, Lp, F] (the taken views), `img` [B, Lp, 36, F], `lengths` [B] (CPU), `insts` [B, Lw].
    graph=True (round 6): dropout offsets and the three recurrences' launch sequences come from a runtime.DeviceClock, the batch
    lives at fixed addresses (`load`: also the path mask, which the eager form builds on the host every iteration), and
    `capture()` / `replay()` run the iteration as ONE hipGraph."""

    rollout_wgrads = False

    def __init__(self, dev, dtype, *, enc=None, dec=None, vocab=992, wemb=256, rnn=512, feature_size=2176, angle_size=128,
                 drop=0.6, feat_drop=0.3, lr=1e-4, clip_norm=CLIP, graph=False):
        self.dev, self.dtype = dev, dtype
        self.enc = enc if enc is not None else SpeakerEncoder(feature_size, rnn, drop, True, angle_size, feat_drop, compute_dtype=dtype).to(dev).train()
        self.dec = dec if dec is not None else SpeakerDecoder(vocab, wemb, 0, rnn, drop, compute_dtype=dtype).to(dev).train()
        self.speaker = Speaker(self.enc, self.dec)
        self.opt_e = optim.FusedAdam([list(self.enc.parameters())], lr=lr, clip_norm=clip_norm)
        self.opt_d = optim.FusedAdam([list(self.dec.parameters())], lr=lr, clip_norm=clip_norm)
        if self._make_clock(dev, graph, self.enc, self.dec) is not None:
            self.opt_e.use_clock(self.clock); self.opt_d.use_clock(self.clock)

    def load(self, batch):
        """The batch at fixed addresses + the path mask of `lengths` on the device (speaker.length2mask)."""
        from .speaker import length2mask
        mask = length2mask(batch["lengths"], self.dev, batch["can"].shape[1])
        if self.live is None:
            self.live = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}
            self.live["ctx_mask"] = mask
            return self.live
        for k in ("can", "img", "insts"):
            self.live[k].copy_(batch[k], non_blocking=True)
        self.live["lengths"] = batch["lengths"]
        self.live["ctx_mask"].copy_(mask, non_blocking=True)
        return self.live

    def iteration(self, batch=None):
        if batch is not None and batch is not self.live:
            if self.clock is None and self.live is None:          # the eager form of rounds 1-5: no copy of the batch, no clock
                with rollout_wgrads(False):
                    return self._eager(batch)
            self.load(batch)
        with rollout_wgrads(False):                               # (parameter gradients added into .grad by the launches themselves)
            return self._iteration()

    def _eager(self, b):
        self.opt_e.zero_grad(); self.opt_d.zero_grad()
        # the feature dropout works in place (units.py:322,331): a training loop hands over fresh feature tensors every batch
        loss = self.speaker.teacher_forcing(b["can"].clone(), b["img"].clone(), b["lengths"], b["insts"], train=True)
        loss.backward()
        self.opt_e.step(); self.opt_d.step()
        return loss

    def _iteration(self):
        b = self.live
        if self.clock is not None:
            self.clock.tick()
        self.opt_e.zero_grad(); self.opt_d.zero_grad()
        loss = self.speaker.teacher_forcing(b["can"].clone(), b["img"].clone(), b["lengths"], b["insts"], train=True, ctx_mask=b["ctx_mask"])
        loss.backward()
        self.opt_e.step(); self.opt_d.step()
        return loss


class _SplitRows(torch.autograd.Function):
    """x [2B, ...] -> (x[:B], x[B:]) as views; backward = ONE concatenation of the two halves' gradients (autograd's own slice nodes
    would each fill a full-size zero tensor, copy their half in and add the two: five launches over 2 x the bytes)."""

    @staticmethod
    def forward(ctx, x, B):
        ctx.B, ctx.shape = B, x.shape
        ctx.set_materialize_grads(False)
        return x[:B], x[B:]

    @staticmethod
    def backward(ctx, g1, g2):
        if g1 is None and g2 is None:
            return None, None
        B = ctx.B
        z = lambda g, n: g if g is not None else torch.zeros((n,) + tuple(ctx.shape[1:]), dtype=(g1 if g1 is not None else g2).dtype,
                                                             device=(g1 if g1 is not None else g2).device)
        return torch.cat((z(g1, B), z(g2, ctx.shape[0] - B)), 0), None


class EnvDropA2CIteration:
    """BASELINE config 3's per-rank iteration -- the reference's DEFAULT EnvDrop iteration (engine/trainer.py:411-427): a
    teacher-forced IL rollout (T_il steps) and a SAMPLED rollout of up to T_rl steps (the reference caps episodes at
    MAX_EPISODE_LEN = 35, configs/envdrop/envdrop_config.yaml:31) scored by A2C with the critic (agent/envdrop.py:186-264), one
    backward over both, clip 40 on encoder and decoder only (trainer.py:425-426), one RMSprop over encoder / decoder / critic.

    The sampled rollout needs the HOST between steps (envdrop.py:196-206: the sampled action goes to the simulator, which answers
    with the next observation).  `read_actions` chooses how:
      True         eager / segmented: D2H of a_t into pinned memory + stream synchronise per step (the reference's loop shape)
      "poll"       segmented graph: the host spins on the pinned action words instead of synchronising (wake-up ~17 us per step)
      "handshake"  ONE hipGraph for the iteration, every host turn a `vln_host_wait` inside it (graphs.HandshakeIterationGraph)
      False        (A/B) the actions never leave the device
    `host_turn(t, actions)` -- the caller's simulator step: `actions` is the pinned int64 row of step t; it may refill the live
    batch's step t + 1 (batches.LiveSteps) and the `rewards` / `masks` rows before returning.  Default: bookkeeping of the STOP
    count (a stand-in for env.step that reads every action).

    tape: a synthetic.tape_to() batch with T_rl steps over a resident DeviceFeatureStore (`tape["store"]`); the IL rollout
    teacher-forces its first T_il steps.  rewards / masks / ended: [T_rl] lists of [B] device tensors and a [B] tensor at fixed
    addresses (what envdrop.py:209-217 computes from the simulator's distances)."""

    def __init__(self, dev, dtype, tape, *, T_il=7, rewards=None, masks=None, ended=None, graph=True, read_actions=True, host_turn=None,
                 enc=None, dec=None, critic=None, sampler_in_step=True, per_step_sampler=False, chain_il=True, chain_backward=True,
                 lr=LR, clip_norm=CLIP, ml_weight=ML_WEIGHT, gamma=0.9, normalize="total", stop_action=None, reward_seed=7, merge_encoders=True):
        self.dev, self.dtype, self.tape, self.T_il, self.T_rl = dev, dtype, tape, T_il, len(tape["steps"])
        self.read_actions, self.ml_weight, self.gamma, self.normalize = read_actions, ml_weight, gamma, normalize
        B, T_rl = tape["B"], self.T_rl
        self.B = B
        self.enc = enc if enc is not None else EncoderLSTM(992, 256, 512, 0, 0.5, True, 1, compute_dtype=dtype).to(dev).train()
        self.dec = dec if dec is not None else EnvDropDecoder(512, 0.5, 0.3, 64, 128, 2176, compute_dtype=dtype).to(dev).train()
        self.cri = critic if critic is not None else Critic(512, 0.5).to(dev).train()
        self.opt = optim.FusedRMSprop([list(self.enc.parameters()), list(self.dec.parameters()), list(self.cri.parameters())], lr=lr,
                                      clip_norm=[clip_norm, clip_norm, 0.0])
        self.store = tape["store"]
        if rewards is None:           # synthetic stand-ins for what the simulator's distances give (envdrop.py:209-217)
            g = torch.Generator().manual_seed(reward_seed)
            rewards = [torch.randn(B, generator=g).sign().to(dev) for _ in range(T_rl)]
            lens_rl = torch.randint(min(4, T_rl), T_rl + 1, (B,), generator=g)
            lens_rl[0] = T_rl
            masks = [(t < lens_rl).to(dev) for t in range(T_rl)]
            ended = (lens_rl < T_rl).to(dev)
        self.rewards, self.masks, self.ended = rewards, masks, ended
        self.clock = DeviceClock(dev).attach(self.enc, self.dec, self.cri) if graph else None
        self.a_host = torch.zeros(T_rl, B, dtype=torch.int64).pin_memory()
        self.a_np = self.a_host.numpy()         # the same pinned memory, for the polling form of the action read
        d = C.c_void_p()
        _lib.check(_lib.load().vln_host_device_pointer(self.a_host.data_ptr(), C.byref(d)), "vln_host_device_pointer")
        self._a_host_dev = int(d.value)         # the device-visible address of the pinned action words (the step's draw stores there itself)
        self.in_step = bool(sampler_in_step) and not per_step_sampler
        self.per_step_sampler, self.chain_il, self.chain_backward = per_step_sampler, chain_il, chain_backward
        # The two rollouts of an iteration encode the SAME instructions (trainer.py:413-416: the sampled rollout restarts the batch) with
        # independent dropout masks: ONE encoder call over 2B rows -- rows [0, B) feed the teacher-forced rollout, rows [B, 2B) the
        # sampled one, each row with its own Philox positions -- instead of two calls (round 6: the recurrence is bound by its L
        # dependent steps, not by its rows; B > 128 runs it in passes).  merge_encoders=False: the two calls (A/B).
        self.merge_encoders = merge_encoders
        self._tokens2 = self._lens2 = None
        self.poll = read_actions in ("poll", "handshake")
        self.stop_action = (tape["steps"][0]["cand_mask"].shape[1] - 1) if stop_action is None else stop_action
        self.host_ended = 0                     # what the stand-in for env.step keeps: episodes that chose STOP so far (read, never fed back)
        self.host_turn = host_turn if host_turn is not None else self._count_stops
        self.poll_deadline_s = 10.0
        self.arena = ops.RolloutArena()
        self.dec.step_graphs = True
        self._st = {}                           # the sampled rollout's running state between two segments
        self._polling = False
        self._run = None
        self.segments = self._segments()

    # ---- the pieces ------------------------------------------------------------------------------------------------------------------
    def _count_stops(self, t, actions):
        self.host_ended = int((actions == self.stop_action).sum())

    def _gather_of(self, s):
        return (self.store, s["rows"], s["vidx"], s["crow"], s["cview"], s["chead"], s["celev"])

    def _encode(self, which):
        """(ctx, h, c) of rollout `which` (0 = teacher-forced, 1 = sampled): two encoder calls, or the halves of one call over 2B rows."""
        tape = self.tape
        if not self.merge_encoders:
            return self.enc(tape["tokens"], tape["lengths32"])
        if which == 0:
            B = self.B
            if self._tokens2 is None:
                self._tokens2 = torch.empty((2 * B,) + tuple(tape["tokens"].shape[1:]), dtype=tape["tokens"].dtype, device=self.dev)
                self._lens2 = torch.empty(2 * B, dtype=tape["lengths32"].dtype, device=self.dev)
            torch.cat((tape["tokens"], tape["tokens"]), 0, out=self._tokens2)       # (the live batch's buffers may have new contents)
            torch.cat((tape["lengths32"], tape["lengths32"]), 0, out=self._lens2)
            ctx2, h2, c2 = self.enc(self._tokens2, self._lens2)
            lp = getattr(ctx2, "_vln_lp", None)
            halves = []
            parts = [_SplitRows.apply(t, B) for t in (ctx2, h2, c2)]
            for k in range(2):
                cx, hh, cc = parts[0][k], parts[1][k], parts[2][k]
                if lp is not None:
                    cx._vln_lp = lp[:B] if k == 0 else lp[B:]
                ops.stamp(cx, hh, cc)
                halves.append((cx, hh, cc))
            self._st["enc_rl"] = halves[1]
            return halves[0]
        return self._st.pop("enc_rl")

    def _il_rollout(self):
        tape, dec = self.tape, self.dec
        ctx, h, c = self._encode(0)
        ht = h
        dec.defer_logits = True            # teacher forcing: the logits are only needed by the loss (formed once per rollout)
        dec.chain_steps = self.chain_il    # ... and nothing reads a step's h_tilde but the next step: consecutive steps share launches
        ce = losses.RolloutCE()
        for s in tape["steps"][:self.T_il]:
            logit, (h, c), ht = dec(s["angle"], None, None, ht, h, c, ctx, tape["seq_mask"], gather=self._gather_of(s))
            ce.add(logit, s["target"], s["cand_mask"])
        return ce.sum(scale=self.ml_weight / self.B)

    def _rl_begin(self):
        tape, dec = self.tape, self.dec
        ctx, h, c = self._encode(1)
        dec.defer_logits = False
        dec.chain_steps = False               # the sampled rollout reads every step's logits
        dec.chain_backward = self.chain_backward     # ... but nothing except the next step consumes its h_tilde
        self._st.update(ctx=ctx, h=h, c=c, ht=h, hidden=[], logps=[], ents=[],
                        sampler=None if self.per_step_sampler else losses.RolloutSampler(clock=self.clock))

    def _rl_step(self, t):
        tape, dec, st = self.tape, self.dec, self._st
        s = tape["steps"][t]
        if self.in_step and st["sampler"] is not None:
            # mask + softmax + draw + log-prob + entropy inside the step's logits launch; the action goes to the pinned words itself
            logit, (h, c), ht = dec(s["angle"], None, None, st["ht"], st["h"], st["c"], st["ctx"], tape["seq_mask"], gather=self._gather_of(s),
                                    sampler=(st["sampler"], s["cand_mask"], None, (self._a_host_dev + 8 * self.B * t) if self.read_actions else 0))
            st.update(h=h, c=c, ht=ht)
            st["hidden"].append(h)
            return
        logit, (h, c), ht = dec(s["angle"], None, None, st["ht"], st["h"], st["c"], st["ctx"], tape["seq_mask"], gather=self._gather_of(s))
        st.update(h=h, c=c, ht=ht)
        st["hidden"].append(h)
        if st["sampler"] is not None:
            a = st["sampler"].step(logit, s["cand_mask"])                       # envdrop.py:186-195 as one launch per step ...
        else:
            a, lp_a, en_a = losses.sample_action(logit, s["cand_mask"])
            st["logps"].append(lp_a); st["ents"].append(en_a)
        if self.read_actions:
            self.a_host[t].copy_(a, non_blocking=True)                          # envdrop.py:198: cpu_a_t = a_t.cpu().numpy()

    def _host_step(self, t):
        if self.poll and self._polling and not torch.cuda.is_current_stream_capturing():
            # the store of a_t is the step's last action: the host spins on the pinned words (armed with -1 before the launch) instead
            # of paying a stream synchronisation's wake-up; sampled actions are >= 0.  Bounded: a device-side timeout (the sticky word)
            # or a deadline ends the wait with an exception instead of a hang.
            row = self.a_np[t]
            if (row < 0).any():
                t0 = time.perf_counter()
                lib = _lib.load()
                while (row < 0).any():
                    if time.perf_counter() - t0 > self.poll_deadline_s:
                        _lib.check(lib.vln_persistent_check(), "vln_persistent_check (while the host waited for the sampled actions)")
                        raise TimeoutError(f"EnvDropA2CIteration: the sampled actions of step {t} did not arrive within {self.poll_deadline_s} s")
            self.host_turn(t, row)
        elif self.read_actions:
            torch.cuda.current_stream().synchronize()
            self.host_turn(t, self.a_np[t])                                     # env.step(cpu_a_t): the host reads the actions

    def _rl_end(self):
        tape, dec, st = self.tape, self.dec, self._st
        if st["sampler"] is not None:
            logps, ents = st["sampler"].stats()                                 # ... and ONE backward node for all steps
        else:
            logps, ents = st["logps"], st["ents"]
        hidden = st["hidden"]
        sl = tape["steps"][len(hidden) - 1]
        _, (last_h, _), _ = dec(sl["angle"], None, None, st["ht"], st["h"], st["c"], st["ctx"], tape["seq_mask"], gather=self._gather_of(sl))
        with torch.no_grad():
            last_v = self.cri(last_h).detach()
        # the critic is row-wise: V of all T steps in ONE call over (steps x batch) rows instead of T calls (the reference
        # loops `self.critic(hidden_states[t])`, envdrop.py:246 -- same function of the same rows)
        vals = list(self.cri(torch.cat(hidden, 0)).view(len(hidden), self.B).unbind(0))
        T = len(hidden)
        rl, _ = losses.a2c_loss(logps, ents, vals, self.rewards[:T], self.masks[:T], last_v, self.ended, self.gamma, self.normalize)
        st.clear()
        return rl

    def _in_arena(self, fn, begin=False):
        def run():
            ops.set_arena(self.arena)
            if begin:
                self.arena.begin()
            try:
                return fn()
            finally:
                ops.set_arena(None)
        return run

    def _first(self):
        if self.clock is not None:
            self.clock.prologue(modules=(self.enc, self.dec))
        self.opt.zero_grad()               # no launch after a step(zero_grads=True): the update cleared the buffer while it read it
        self._st["il"] = self._il_rollout()
        self._rl_begin()
        self._rl_step(0)

    def _last(self):
        il = self._st.pop("il")
        loss = il + self._rl_end()
        loss.backward()
        self.opt.step(zero_grads=True)
        return loss

    def _segments(self):
        """[graph: prologue, the IL rollout, the RL encoder, RL step 0 + draw + a_0 to the host] [host: turn 0] [graph: RL step 1] ...
        [graph: last step, critic, A2C loss, the backward of BOTH rollouts, clip + RMSprop]."""
        segs = [("graph", self._in_arena(self._first, begin=True)), ("host", lambda: self._host_step(0))]
        for t in range(1, self.T_rl):
            segs += [("graph", self._in_arena(lambda t=t: self._rl_step(t))), ("host", lambda t=t: self._host_step(t))]
        segs.append(("graph", self._in_arena(self._last)))
        return segs

    # ---- the three forms ------------------------------------------------------------------------------------------------------------
    def iteration(self):
        out = None
        for _, fn in self.segments:
            r = fn()
            out = r if r is not None else out
        return out

    def capture(self, warmup=0):
        """Record the iteration: read_actions == "handshake" -> ONE graph whose host turns are waits inside it; otherwise T_rl + 1
        graph segments with the host's turns between them.  Returns `self.replay`."""
        if self.clock is None:
            raise RuntimeError("EnvDropA2CIteration.capture: built with graph=False (no device clock)")
        for _ in range(warmup):
            self.iteration()
        if self.read_actions == "handshake":
            g = HandshakeIterationGraph(self.segments, self.clock).capture()
        else:
            g = SegmentedIterationGraph(self.segments, self.clock).capture()
        self._polling = self.poll
        self.graph = g
        return self.replay

    def replay(self):
        if self._polling:
            self.a_np[:] = -1              # arm the pinned words (every replayed step's store overwrites its row)
        return self.graph.replay()

    def describe(self):
        return dict(iteration=(("ONE hipGraph, the host's turns are waits inside it" if self.read_actions == "handshake" else f"{self.T_rl + 1} hipGraph segments")
                               if getattr(self, "graph", None) is not None else "per-step hipGraphs, Python-driven"),
                    per_step_action_read=("host spins on the pinned action words" if self._polling else bool(self.read_actions)),
                    plan_hits=self.dec.plan_hits, arena_misses=self.arena.misses)


class EnvDropHostLoopIteration:
    """The EnvDrop IL iteration with THE HOST IN THE LOOP, in the reference's loop shape (agent/envdrop.py:151-220): per decoder
    step the simulator's new observation is marshalled on the host (agent/base.py:141-178) -- here: the step's packed index vectors,
    a few KB, features stay in the resident table --, the step runs with its candidate logits and CE term formed in the step, and the
    chosen action a_t goes back to the host (`a_t.cpu()`, envdrop.py:198) where `env_step(t, actions)` takes it before the next
    observation exists.  Two forms with identical results:

      iteration(k)       eager launches + per-step hipGraphs; per step one pinned H2D copy and one D2H + stream synchronisation
      capture(); replay(k)   ONE hipGraph (graphs.HandshakeIterationGraph): every host turn is a wait INSIDE the graph whose launch
                         also pulls the observation the host just wrote out of pinned memory (vln_host_wait_fetch); the action words
                         are stored to pinned memory and polled by the host -- no launch, copy call or stream wake-up between steps

    `steps_feed` = batches.LiveSteps over the episode batches (batch k's blobs stand in for what the simulator produces)."""

    UNSET = -2              # the armed value of an action word (teacher actions are >= -1: -1 = the episode has ended, envdrop.py:199-203)

    def __init__(self, dev, dtype, steps_feed, store, *, env_step=None, enc=None, dec=None, lr=LR, clip_norm=CLIP, ml_weight=ML_WEIGHT,
                 spin_limit=0):
        self.dev, self.dtype, self.ls, self.store, self.ml_weight, self.spin_limit = dev, dtype, steps_feed, store, ml_weight, spin_limit
        self.enc = enc if enc is not None else EncoderLSTM(992, 256, 512, 0, 0.5, True, 1, compute_dtype=dtype).to(dev).train()
        self.dec = dec if dec is not None else EnvDropDecoder(512, 0.5, 0.3, 64, 128, 2176, compute_dtype=dtype).to(dev).train()
        self.opt = optim.FusedRMSprop([list(self.enc.parameters()), list(self.dec.parameters())], lr=lr, clip_norm=clip_norm)
        self.arena = ops.RolloutArena()
        self.dec.step_graphs = True
        self.dec.defer_logits = False           # the action is chosen from this step's logits (argmax / sample / teacher): formed in the step
        self.dec.chain_steps = False
        self.T = len(steps_feed.live["steps"])
        self.B = steps_feed.live["B"]
        self.a_host = torch.full((self.T, self.B), self.UNSET, dtype=torch.int64).pin_memory()
        self.a_np = self.a_host.numpy()
        from .graphs import _device_pointer
        self._a_dev = _device_pointer(self.a_host)
        self.action_store = "kernel"
        self.fetch_in_wait = True               # the step's observation pulled by the wait launch itself (vln_host_wait_fetch)
        self.use_feed = True                    # the top blob (tokens, lengths, sequence mask) pulled by the graph's first launch
        self.mismatches = 0
        self.env_step = env_step if env_step is not None else self._fake_env_step
        self.clock, self.graph, self._k = None, None, 0
        self.poll_deadline_s = 10.0

    def _fake_env_step(self, t, actions):
        """Stand-in for env.step(cpu_a_t): checks what the agent sent against the environment's own teacher tape (episodes whose action is
        -1 have ended, envdrop.py:199-203); the next observation is batch k's next blob."""
        self.mismatches += int((actions != self.ls.host_targets[self._k % len(self.ls.host_targets)][t]).sum())

    def _gather_of(self, s):
        return (self.store, s["rows"], s["vidx"], s["crow"], s["cview"], s["chead"], s["celev"])

    def _step(self, t, tape, st):
        s = tape["steps"][t]
        logits, (h, c), ht = self.dec(s["angle"], None, None, st["ht"], st["h"], st["c"], st["ctx"], tape["seq_mask"], gather=self._gather_of(s))
        st.update(h=h, c=c, ht=ht)
        # envdrop.py:173-179: the step's CE term -- recorded here (the logits exist, the action below depends on them), evaluated for
        # the whole rollout in one launch each way (losses.RolloutCE: nothing in the rollout reads the loss VALUE)
        st["ce"].add(logits, s["target"], s["cand_mask"])
        return s["target"]                                                      # teacher forcing: a_t = target (envdrop.py:183)

    def _finish(self, st):
        loss = st["ce"].sum(scale=self.ml_weight / self.B)                      # envdrop.py:179,268
        loss.backward()
        self.opt.step(zero_grads=True)
        return loss

    def _in_arena(self, fn, begin=False):
        def run():
            ops.set_arena(self.arena)
            if begin:
                self.arena.begin()
            try:
                return fn()
            finally:
                ops.set_arena(None)
        return run

    # ---- eager: the reference's loop, one host round trip per step ------------------------------------------------------------------
    def iteration(self, k):
        self._k = k
        ls = self.ls
        ops.set_arena(self.arena); self.arena.begin()
        try:
            tape = ls.load_top(k)
            if self.clock is not None:
                self.clock.prologue(None, (self.enc, self.dec))
            self.opt.zero_grad()
            ctx, h, c = self.enc(tape["tokens"], tape["lengths32"])
            st = dict(ctx=ctx, h=h, c=c, ht=h, ce=losses.RolloutCE())
            for t in range(self.T):
                ls.load_step(k, t)                                              # this step's observation: pinned host -> device
                a_t = self._step(t, tape, st)
                self.env_step(t, a_t.cpu().numpy())                             # envdrop.py:198: the action reaches the simulator
            return self._finish(st)
        finally:
            ops.set_arena(None)

    # ---- ONE graph, the host's turns inside it ----------------------------------------------------------------------------------------
    def capture(self, warmup=3):
        from .staging import HostBatchFeed
        ls = self.ls
        if self.clock is None:
            self.clock = DeviceClock(self.dev).attach(self.enc, self.dec)
            self.clock.attach(self.store)
            warmup = max(warmup, 2)              # the modules' first calls on a clock register their sequence words: never inside a capture
        for k in range(warmup):
            self.iteration(k)
        torch.cuda.synchronize()
        self.feed = HostBatchFeed(ls.top_dev) if self.use_feed else None
        self._top = [self.feed.register(b) for b in ls.top_host] if self.use_feed else None
        self._mail = [torch.zeros(ls.step_bytes, dtype=torch.uint8).pin_memory() for _ in range(self.T)]
        # numpy views for the host turns: a torch CPU op there (even a 4 KB copy_) wakes the intra-op thread pool, whose workers then
        # spin-wait for their block time -- 16 spinning threads burn a container's CPU quota and the cgroup is throttled for the rest
        # of every 100 ms period (measured: 2.0 ms iterations with an 88 ms stall every sixth, scripts/hostloop_probe.py)
        mail_np = [m.numpy() for m in self._mail]
        step_np = [[b.numpy() for b in tp] for tp in ls.step_host]
        tape, st = ls.live, {}

        def first():
            self.clock.prologue(self.feed, (self.enc, self.dec))               # ONE launch: the top blob's pull, the tick, the shadows
            self.opt.zero_grad()
            ctx, h, c = self.enc(tape["tokens"], tape["lengths32"])
            st.update(ctx=ctx, h=h, c=c, ht=h, ce=losses.RolloutCE())

        def step(t):
            if not self.fetch_in_wait:                                          # (A/B) the observation as an H2D memcpy node behind a plain wait
                ls.step_dev[t].copy_(self._mail[t], non_blocking=True)
            a_t = self._step(t, tape, st)
            if self.action_store == "kernel":                                   # the action reaches the host's pinned words
                _lib.check(_lib.load().vln_store_to_host(a_t.data_ptr(), self._a_dev + 8 * self.B * t, 8 * self.B, _lib.raw_stream()), "vln_store_to_host")
            else:                                                               # (A/B) a D2H memcpy node
                self.a_host[t].copy_(a_t, non_blocking=True)
            if t == self.T - 1:
                return self._finish(st)

        def turn(t):
            if t > 0:
                self._await_action(t - 1)
            mail_np[t][:] = step_np[self._k % len(step_np)][t]                  # the simulator's observation for step t, marshalled
        segs = [("graph", self._in_arena(first, begin=True))]
        for t in range(self.T):
            segs += [("host", lambda t=t: turn(t), (self._mail[t], ls.step_dev[t]) if self.fetch_in_wait else None),
                     ("graph", self._in_arena(lambda t=t: step(t)))]
        self.graph = HandshakeIterationGraph(segs, self.clock, spin_limit=self.spin_limit).capture()
        return self.graph

    def _await_action(self, t):
        row = self.a_np[t]
        if (row == self.UNSET).any():
            t0, lib = time.perf_counter(), _lib.load()
            while (row == self.UNSET).any():
                if time.perf_counter() - t0 > self.poll_deadline_s:
                    _lib.check(lib.vln_persistent_check(), "vln_persistent_check (while the host waited for the actions)")
                    raise TimeoutError(f"EnvDropHostLoopIteration: the actions of step {t} did not arrive within {self.poll_deadline_s} s")
        self.env_step(t, row)

    def replay(self, k):
        self._k = k
        self.a_np[:] = self.UNSET
        if self.feed is not None:
            self.feed.select(self._top[k % len(self._top)])
        else:                                    # (A/B) one H2D copy in front of the graph
            self.ls.load_top(k)
        out = self.graph.replay()
        if self.feed is not None:
            self.feed.launched()
        self._await_action(self.T - 1)           # the last action reaches the simulator too (envdrop.py:198)
        return out
