"""The caller's episode batch at FIXED device addresses (agent/base.py:114-178 marshals every batch on the host; here the
marshalled bytes land in the same device buffers every iteration, so step plans and hipGraphs keyed by addresses keep
replaying while the data changes).

  LiveBatch   the whole teacher-forced batch (tokens, masks, every step's index vectors, targets) as one blob: pulled from
              pinned host memory by the iteration's first launch (staging.HostBatchFeed), pushed ahead, copied, or on device
  LiveSteps   host-in-the-loop rollouts: one pinned blob PER STEP, copied when the simulator has produced the observation
"""
from __future__ import annotations

import torch

from .staging import HostBatchFeed


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
            self.feed = HostBatchFeed(self.live_blob, prefetch=source == "push")
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

    def put(self, k, tape):
        """A trainer's NEW batch into blob slot k (same keys, shapes and dtypes as the batches this LiveBatch was built from; CPU
        tensors or anything `.cpu()` accepts): the marshalling of agent/base.py:114-178 ends here.  Rotate at least two slots: slot k
        must not be rewritten while an iteration that was `load(k)`-ed is still in flight.  Packing goes through numpy views of the
        blob (no torch CPU op per field)."""
        blob = self.blobs[k % len(self.blobs)]
        if blob.is_cuda:                                   # source "device": stage through a host buffer, one H2D copy
            host = torch.empty(self.nbytes, dtype=torch.uint8)
            self._pack(host.numpy(), tape)
            blob.copy_(host, non_blocking=False)
            return
        self._pack(blob.numpy(), tape)

    def _pack(self, dst, tape):
        for (name, o, n, dt, shape), (name2, t) in zip(self.layout, self._items(tape)):
            if name != name2 or tuple(t.shape) != shape or t.dtype != dt:
                raise ValueError(f"LiveBatch.put: the batch's layout differs at {name}: {tuple(t.shape)} {t.dtype} vs {shape} {dt}")
            dst[o:o + n] = t.detach().cpu().contiguous().view(-1).view(torch.uint8).numpy()

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
