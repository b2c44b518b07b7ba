"""One training iteration as ONE hipGraph.

The reference's training loop (engine/trainer.py:411-427: rollout, `loss.backward()`, clip, `optimizer.step()`) issues ~170
launches per EnvDrop iteration at B = 64; on this path every one of them is a pure function of device memory once

  * the batch lives at fixed addresses (a trainer marshals each new batch into the same buffers: bench.LiveBatch),
  * per-iteration buffers come from an address-stable arena (ops.RolloutArena) or from the capture's own memory pool,
  * everything that used to be a HOST value per launch -- dropout offsets, the launch sequence of the recurrence's tagged
    hand-offs -- is read from device words that a tick launch bumps (runtime.DeviceClock).

`IterationGraph` captures such an iteration (forward, loss, backward, optimizer step) with torch's stream capture -- PyTorch
supplies the capture plumbing and the memory pool, every captured node is one of this library's kernels -- and replays it
with one `hipGraphLaunch`: the host's share of an iteration drops from ~1.5 ms of Python to the batch copy plus one call, so
a slow host (or eight ranks sharing one) no longer bounds the step, and the graph's internal edges cost less than stream
launches (scripts/boundary_probe.hip: 1.9 us per dependent trivial kernel inside a long graph against 2.3-2.5 us inside a
12-node one and ~3 us eager).

Contract of `fn` (the iteration): it calls `clock.tick()` first, touches only device-resident inputs at fixed addresses,
does not synchronise with the host, and leaves its outputs (e.g. the loss) in tensors that the NEXT replay overwrites --
read or copy them before replaying again.  Sampled rollouts (whose next observation depends on a drawn action that the
simulator must see) cannot be captured; teacher-forced iterations can.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch

import contextlib
import gc

from . import _lib
from .runtime import DeviceClock


@contextlib.contextmanager
def _no_gc_while_capturing():
    """Python's cyclic collector must not run inside a stream capture: it may destroy objects left over from EARLIER work (an old
    CUDAGraph, tensors with cross-stream uses, autograd nodes kept alive by a traceback) whose destructors issue device calls that
    are illegal while a stream captures -- an exception in a destructor aborts the process (seen once in the round-5 test suite:
    `Fatal Python error: Aborted` with the interpreter "Garbage-collecting" inside an autograd node's apply during a capture).
    Collect first, keep the collector off for the capture, restore its state after."""
    gc.collect()
    was = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was:
            gc.enable()


class IterationGraph:
    def __init__(self, fn: Callable[[], object], clock: DeviceClock):
        self.fn, self.clock = fn, clock
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.out = None
        self.replays = 0

    def capture(self, warmup: int = 0, debug_dump: Optional[str] = None, capture_error_mode: str = "global"):
        """Run `warmup` eager iterations (first-use allocations, weight shadows, step plans), then record one.
        capture_error_mode="thread_local": for iterations that contain collectives of a process group -- its watchdog thread polls
        events while the capture is open (see SegmentedIterationGraph.capture)."""
        for _ in range(warmup):
            self.fn()
        torch.cuda.synchronize()
        lib = _lib.load()
        _lib.check(lib.vln_persistent_check(), "vln_persistent_check")
        self.clock.restart_sequences()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        if debug_dump:
            g.enable_debug_mode()
        epoch0 = self.clock.epoch
        try:
            with _no_gc_while_capturing(), torch.cuda.graph(g, capture_error_mode=capture_error_mode):
                self.out = self.fn()
        finally:
            # The captured tick did not run: the device words still hold the pre-capture values.  Also when `fn` RAISED inside the
            # capture: without this the clock's host mirror would stay one tick ahead of the device words and a
            # caller that falls back to eager launches on the same clock would draw every dropout offset one iteration off.
            # Only a tick that WAS counted is taken back (`fn` may raise before it issues its tick / prologue).
            while self.clock.epoch > epoch0:
                self.clock.uncount()
        if debug_dump:
            g.debug_dump(debug_dump)   # graphviz text of the captured nodes (kernel names, memcpy / memset nodes, edges)
        self.graph = g
        self._check = lib.vln_persistent_check
        return self

    def replay(self):
        # A replay has no host code between its launches: what an EARLIER replay raised on the device (a timed-out bounded
        # wait, an out-of-range gather index -- host-mapped words, no synchronisation to read them) is reported here, loudly;
        # that replay's update has already been applied, the caller restores from its last good state.
        st = self._check()
        if st:
            _lib.check(st, "vln_persistent_check (raised by an earlier replay)")
        self.clock.replayed()          # host mirror of the tick launch inside the graph (+ the launch-sequence wrap guard)
        self.graph.replay()
        self.replays += 1
        return self.out


class SegmentedIterationGraph:
    """One training iteration as a SEQUENCE of hipGraphs with host code between them -- the data-parallel form of
    `IterationGraph` (round 4).

    A data-parallel iteration has two points where the HOST must act (the reference's loop is
    engine/trainer.py:411-427): the decoder's gradients are final before the encoder's BPTT starts, so their RCCL all-reduce
    is started there and runs under the BPTT; the rest of the bucket is reduced after the backward, before clip + update.  A
    single captured graph cannot contain those calls, so round 3's N > 1 path issued all ~170 launches from Python (1.5 ms of
    host time per iteration against 1.7 ms of GPU time).  Here the iteration is cut AT the exchange points:

        graph A   tick, zero_grad, encoder forward, decoder steps, loss, the decoder's backward + its weight gradients
        host      start_allreduce(decoder slice)                       (asynchronous, the process group's stream)
        graph B   the encoder's backward (BPTT, its weight gradients)
        host      allreduce(rest) + wait
        graph C   clip + optimizer step

    `segments` is a list of ("graph", fn) / ("host", fn) pairs; the graph segments are captured one after the other INTO ONE
    memory pool on one stream (tensors made in one segment -- the autograd graph, the loss -- are consumed by the next), the host
    segments run between them during capture too (so the ranks' collective sequences stay matched) and on every replay.
    N = 1 and N > 1 replay the same kernels in the same order as the single graph: bit-identical results
    (tests/test_hip_graphs.py).  Same contract for the iteration as `IterationGraph`: the first graph segment ticks the clock."""

    def __init__(self, segments, clock: DeviceClock):
        self.segments, self.clock = list(segments), clock
        self.plan = None           # [("graph", CUDAGraph) | ("host", fn)]
        self.replays = 0
        self.out = None

    def capture(self):
        torch.cuda.synchronize()
        lib = _lib.load()
        _lib.check(lib.vln_persistent_check(), "vln_persistent_check")
        self.clock.restart_sequences()
        torch.cuda.synchronize()
        pool = torch.cuda.graph_pool_handle()
        stream = torch.cuda.Stream()
        plan = []
        epoch0 = self.clock.epoch
        try:
            for sg in self.segments:
                kind, fn = sg[0], sg[1]
                if kind == "graph":
                    g = torch.cuda.CUDAGraph()
                    # thread_local: a process group's watchdog thread polls the events of collectives that are still in flight
                    # (the early slice issued right before this segment) -- under the default "global" mode such a call from
                    # another thread is an error while ANY stream captures.  Launches from the autograd thread are captured all
                    # the same: capturing is a property of the stream.
                    with _no_gc_while_capturing(), torch.cuda.graph(g, pool=pool, stream=stream, capture_error_mode="thread_local"):
                        r = fn()
                    if r is not None:
                        self.out = r
                    plan.append(("graph", g))
                else:
                    # Host code between two captures: it runs for real (a collective on whatever the buffers hold: the captured
                    # kernels have not run) -- every rank does the same, so the collective sequences stay matched.  It must be
                    # ordered after the capture stream's (empty) work like a replay would order it: nothing to wait for here.
                    fn()
                    plan.append(("host", fn))
        finally:
            while self.clock.epoch > epoch0:       # the captured tick did not run (also when a segment raised: see IterationGraph.capture)
                self.clock.uncount()
        self.plan = plan
        self._check = lib.vln_persistent_check
        return self

    def run_eager(self):
        """The same pieces issued as plain launches, in order (warm-up iterations; the CPU / gloo tests of the segment order)."""
        out = None
        for sg in self.segments:
            r = sg[1]()
            out = r if r is not None else out
        return out

    def replay(self):
        st = self._check()
        if st:
            _lib.check(st, "vln_persistent_check (raised by an earlier replay)")
        self.clock.replayed()
        for kind, x in self.plan:
            if kind == "graph":
                x.replay()
            else:
                x()
        self.replays += 1
        return self.out



class HandshakeIterationGraph:
    """One training iteration whose rollout needs the HOST between steps, as ONE hipGraph (round 5).

    `SegmentedIterationGraph` replays one graph per step and runs the host's turn (read the sampled action, step the simulator:
    envdrop.py:196-206) between two graph launches: ~40 us of launch latency and stream wake-up per step stand between the
    steps' kernels (36 segments, 1.4 ms of a 7.7 ms IL + A2C iteration).  Here the `("graph", fn)` segments are captured back to
    back into ONE graph and every `("host", fn)` segment becomes a one-wave launch that WAITS for the host (`vln_host_wait`: spins
    on a pinned flag word until it holds this iteration's device-clock value).  `replay()` launches the graph once and then plays
    the host's part: for each host segment in order, `fn()` (it polls what the step before it stored to pinned memory -- the
    sampled actions --, does the host's work and leaves the next step's inputs in place), then the flag is written and the device
    goes on.  Same kernels in the same order as the segmented form: bit-identical results.  The host segments are NOT run
    while capturing (there is nothing to read yet)."""

    POISON = 0xFFFFFFFFFFFFFFFF        # VLN_HOST_WAIT_POISON: "the host has given this iteration up"

    def __init__(self, segments, clock: DeviceClock, spin_limit: int = 0):
        """segments: ("graph", fn) | ("host", fn) | ("host", fn, (src, dst)) -- the third form's wait also PULLS `src` (a pinned host
        uint8 tensor the host turn fills: the next step's packed observation) into the device tensor `dst` in the same launch
        (vln_host_wait_fetch).  spin_limit: vln_host_wait's bound (0 = 2 s of wall clock, > 0 polls, < 0 microseconds)."""
        self.segments, self.clock, self.spin_limit = list(segments), clock, int(spin_limit)
        n = sum(1 for sg in self.segments if sg[0] == "host")
        self.flags = torch.zeros(max(n, 1), dtype=torch.int64).pin_memory()
        self._flags_dev = _device_pointer(self.flags)
        self._flags_np = self.flags.numpy().view("uint64")
        # one acknowledgement word per host turn: the wait of turn i stores the iteration's clock value once it is over (and its mailbox
        # read); the host starts turn i of the NEXT replay only after it has seen the previous replay's value there
        self.acks = torch.zeros(max(n, 1), dtype=torch.int64).pin_memory()
        self._acks_dev = _device_pointer(self.acks)
        self._acks_np = self.acks.numpy().view("uint64")
        self._prev_want = None
        self.ack_deadline_s = 10.0
        self.graph = None
        self.host_fns = []
        self.replays = 0
        self.out = None
        self.poisoned = 0

    def capture(self):
        torch.cuda.synchronize()
        lib = _lib.load()
        _lib.check(lib.vln_persistent_check(), "vln_persistent_check")
        self.clock.restart_sequences()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        epoch0 = self.clock.epoch
        host_fns = []
        try:
            with _no_gc_while_capturing(), torch.cuda.graph(g):
                for sg in self.segments:
                    kind, fn = sg[0], sg[1]
                    if kind == "graph":
                        r = fn()
                        if r is not None:
                            self.out = r
                        continue
                    flag, ack = self._flags_dev + 8 * len(host_fns), self._acks_dev + 8 * len(host_fns)
                    if len(sg) > 2 and sg[2] is not None:
                        src, dst = sg[2]
                        _lib.check(lib.vln_host_wait_fetch(flag, self.clock.ptr, self.spin_limit, _device_pointer(src), dst.data_ptr(),
                                                           src.numel() * src.element_size(), ack, _lib.raw_stream()), "vln_host_wait_fetch")
                    else:
                        _lib.check(lib.vln_host_wait(flag, self.clock.ptr, self.spin_limit, ack, _lib.raw_stream()), "vln_host_wait")
                    host_fns.append(fn)
        finally:
            while self.clock.epoch > epoch0:
                self.clock.uncount()
        self.graph, self.host_fns = g, host_fns
        self._check = lib.vln_persistent_check
        return self

    def _await_ack(self, i, prev):
        _await_ack_impl(self._acks_np, i, prev, self.ack_deadline_s, self._check)

    def replay(self):
        st = self._check()
        if st:
            _lib.check(st, "vln_persistent_check (raised by an earlier replay)")
        self.clock.replayed()
        want = self.clock.host                # what the device clock's word holds once this replay's tick has run
        self.graph.replay()
        i, prev = 0, self._prev_want
        try:
            for i, fn in enumerate(self.host_fns):
                if prev is not None and self._acks_np[i] != prev:
                    self._await_ack(i, prev)  # the host is a whole iteration ahead: turn i of the PREVIOUS replay is not over yet
                fn()
                self._flags_np[i] = want      # the device's wait for host turn i ends here
            i = len(self.host_fns)
            self._prev_want = want
        finally:
            if i < len(self.host_fns):
                # A host turn raised (simulator error, KeyboardInterrupt, a poll deadline): the device must not sit in the remaining
                # waits until their bound.  Every flag not yet released gets the POISON value: each wait ends at once and raises the
                # sticky word, so the NEXT library entry (vln_persistent_check at the top of the next replay, the optimizer step) reports
                # this iteration as invalid -- its update ran on stale inputs; the caller restores its last good state.
                self._flags_np[i:len(self.host_fns)] = self.POISON
                self.poisoned += 1
                self._prev_want = None        # (the abandoned waits acknowledge nothing: the next replay starts from a drained queue)
                torch.cuda.current_stream().synchronize()
        self.replays += 1
        return self.out


def _await_ack_impl(acks, i, prev, deadline_s, check):
    import time
    t0 = time.perf_counter()
    while acks[i] != prev:
        if time.perf_counter() - t0 > deadline_s:
            st = check()
            if st:
                _lib.check(st, "vln_persistent_check (while the host waited for the previous iteration's turn to end)")
            raise TimeoutError(f"HandshakeIterationGraph: host turn {i} of the previous replay was not acknowledged within {deadline_s} s")


def _device_pointer(pinned: torch.Tensor) -> int:
    """The device-visible address of a pinned host tensor (vln_host_device_pointer)."""
    import ctypes as C_
    d = C_.c_void_p()
    _lib.check(_lib.load().vln_host_device_pointer(pinned.data_ptr(), C_.byref(d)), "vln_host_device_pointer")
    return int(d.value)
