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

from . import _lib
from .runtime import DeviceClock


class IterationGraph:
    def __init__(self, fn: Callable[[], object], clock: DeviceClock):
        self.fn, self.clock = fn, clock
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.out = None
        self.replays = 0

    def capture(self, warmup: int = 0):
        """Run `warmup` eager iterations (first-use allocations, weight shadows, step plans), then record one."""
        for _ in range(warmup):
            self.fn()
        torch.cuda.synchronize()
        lib = _lib.load()
        _lib.check(lib.vln_persistent_check(), "vln_persistent_check")
        self.clock.restart_sequences()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self.out = self.fn()
        self.clock.uncount()           # the captured tick did not run: the device words still hold the pre-capture values
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
