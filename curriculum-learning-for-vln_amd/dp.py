"""Data-parallel plumbing: one process per GPU, episodes sharded across ranks, ONE RCCL all-reduce of a flat
fp32 gradient bucket per optimizer step.

The reference has no distributed code (single process, single GPU; trainer.py:421-427 does
zero_grad -> backward -> clip -> step).  Episodes are independent, so the only exchange is the gradient
sum.  xGMI is point-to-point: one large bucket (EnvDrop: 10.5 M floats = 42 MB) keeps every link busy and
costs one collective launch instead of ~30 per-tensor ones.

`GradBucket` makes every `p.grad` a VIEW into one contiguous buffer, so autograd accumulates straight into
the bucket (no gather/scatter copies), `zero()` is one memset and `allreduce()` is one collective.
Device-agnostic: the same code runs under `gloo` on CPU (tests) and `nccl` (= RCCL) on MI355X.
"""
from __future__ import annotations

from typing import Iterable, List, Sequence

import torch
import torch.distributed as dist


def stride_shard(n_items: int, rank: int, world: int) -> List[int]:
    """Rows rank, rank+world, ... : keeps a length-sorted batch sorted and length-balanced on every rank
    (the reference sorts each minibatch by instruction length, common_env.py:204-205)."""
    return list(range(rank, n_items, world))


def _dp_active(group=None) -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1


class BucketReducer:
    """Overlapped exchange of ONE flat gradient buffer.  `start(lo, hi)` hands a finished slice (e.g. the decoder's
    gradients, complete before the encoder's BPTT begins) to an asynchronous all-reduce -- with nccl (= RCCL) it runs on
    the process group's own stream behind an event on the caller's stream, so it overlaps the rest of backward;
    `finish()` reduces whatever was not started early and waits for everything.  Every rank must call start/finish in
    the same order with the same ranges (they are collectives)."""

    def __init__(self, flat: torch.Tensor):
        self.flat = flat
        self.pending = []          # (lo, hi, work)

    def start(self, lo: int, hi: int, group=None):
        if hi <= lo or not _dp_active(group):
            return
        w = dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=group, async_op=True)
        self.pending.append((lo, hi, w))

    def finish(self, group=None):
        if not _dp_active(group):
            self.pending = []
            return
        pos, n = 0, self.flat.numel()
        late = []
        for lo, hi, _ in sorted(self.pending, key=lambda t: t[0]):
            if lo > pos:
                late.append(dist.all_reduce(self.flat[pos:lo], op=dist.ReduceOp.SUM, group=group, async_op=True))
            pos = max(pos, hi)
        if pos < n:
            late.append(dist.all_reduce(self.flat[pos:n], op=dist.ReduceOp.SUM, group=group, async_op=True))
        works = [w for _, _, w in self.pending] + late
        if works and dist.get_backend(group) == "nccl":
            # RCCL runs every collective of the group on ONE stream, in issue order: waiting for the LAST one joins them all with a
            # single cross-stream edge (each edge costs tens of microseconds on this stack: profiles/round4_notes.md)
            works[-1].wait()
        else:
            for w in works:
                w.wait()
        self.pending = []


class BackwardCut:
    """Cuts ONE backward pass in two at a set of tensors (the instruction encoder's outputs): the data-parallel iteration starts
    the all-reduce of the decoder's gradients -- final when the decoder's backward is done -- BEFORE the encoder's backward
    (BPTT) runs, and a captured iteration must end its first hipGraph segment there (graphs.SegmentedIterationGraph).

        ctx, h, c = cut.at(*encoder(tokens, lengths))      # detached leaves that stand in for the encoder's outputs
        loss = decoder_rollout(ctx, h, c); loss.backward()   # stops at the leaves: every decoder gradient is final
        ... start_allreduce(decoder slice) ...
        cut.resume()                                         # the encoder's backward, fed with the leaves' gradients

    Same arithmetic as one `loss.backward()`: each leaf's gradient is what autograd would have handed to the encoder's node."""
    carry = ("_vln_lp", "_vln_born")        # attributes the modules hang on their outputs (bf16 stream copy, arena stamp)

    def __init__(self):
        self.outs, self.leaves = None, None

    def at(self, *tensors):
        self.outs = tuple(tensors)
        self.leaves = tuple(t.detach().requires_grad_(True) for t in tensors)
        for new, old in zip(self.leaves, self.outs):
            for a in self.carry:
                if hasattr(old, a):
                    setattr(new, a, getattr(old, a))
        return self.leaves

    def resume(self):
        outs, leaves = self.outs, self.leaves
        self.outs = self.leaves = None
        pairs = [(o, l.grad) for o, l in zip(outs, leaves) if l.grad is not None and o.requires_grad]
        if pairs:
            torch.autograd.backward([o for o, _ in pairs], [g for _, g in pairs])


class GradBucket:
    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("GradBucket: no trainable parameters")
        dev, dt = self.params[0].device, self.params[0].dtype
        total = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(total, dtype=dt, device=dev)
        off = 0
        self.views, self._span = [], {}
        for p in self.params:
            v = self.flat[off:off + p.numel()].view_as(p)
            p.grad = v
            self.views.append(v)
            self._span[id(p)] = (off, off + p.numel())
            off += p.numel()
        self.reducer = BucketReducer(self.flat)

    def zero(self):
        """Replacement for optimizer.zero_grad(): keeps the .grad views alive."""
        self.flat.zero_()
        for p, v in zip(self.params, self.views):
            if p.grad is not v:      # someone replaced .grad (e.g. zero_grad(set_to_none=True)): re-attach
                p.grad = v

    def span_of(self, params: Iterable[torch.nn.Parameter]):
        """[lo, hi) of the bucket covering `params` (they must be adjacent in the bucket, e.g. one module's parameters)."""
        spans = sorted(self._span[id(p)] for p in params if id(p) in self._span)
        if not spans:
            return 0, 0
        for (_, h0), (l1, _) in zip(spans, spans[1:]):
            if h0 != l1:
                raise ValueError("GradBucket.span_of: parameters are not adjacent in the bucket")
        return spans[0][0], spans[-1][1]

    def start_allreduce(self, params: Iterable[torch.nn.Parameter], group=None):
        """Begin reducing the gradients of `params` now (call when they are final, e.g. from the decoder's
        `grads_ready_hook`); `allreduce()` later covers the rest and waits."""
        lo, hi = self.span_of(params)
        self.reducer.start(lo, hi, group)

    def allreduce(self, group=None, average: bool = False):
        """Sum (or mean) the bucket over all ranks.  No-op without an initialised process group."""
        if not _dp_active(group):
            return
        self.reducer.finish(group)
        if average:
            self.flat.div_(dist.get_world_size(group))


def allreduce_scalar(x: torch.Tensor, group=None) -> torch.Tensor:
    """Global count for the A2C `total` normalisation (envdrop.py:258-262) under data parallelism."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(x, op=dist.ReduceOp.SUM, group=group)
    return x


def gather_item_losses(item_index: torch.Tensor, item_loss: torch.Tensor, group=None):
    """SELF-PACE bookkeeping under data parallelism (curriculum.py:311-314: `loss_for_item[cur_batch_idx] = ...`): every rank
    holds the per-episode losses of ITS shard; the curriculum's weight update (curriculum.py:428-448) needs all of them on
    every replica.  -> (index [B_global] int64, loss [B_global]) in rank order; shards must have equal sizes (stride
    sharding of a batch divisible by the world size).  Two tiny all-gathers per iteration; identity without a group."""
    if not _dp_active(group):
        return item_index, item_loss
    world = dist.get_world_size(group)
    idx = [torch.empty_like(item_index) for _ in range(world)]
    los = [torch.empty_like(item_loss) for _ in range(world)]
    dist.all_gather(idx, item_index.contiguous(), group=group)
    dist.all_gather(los, item_loss.detach().contiguous(), group=group)
    return torch.cat(idx), torch.cat(los)


def self_pace_batch_loss(weight_rows: torch.Tensor, per_sample_loss: torch.Tensor) -> torch.Tensor:
    """`torch.dot(self.weight[cur_batch_idx], cur_loss)` (curriculum.py:296) on this rank's shard: the all-reduce of the
    gradients sums the shards' terms, so no extra normalisation is needed for the EnvDrop form (the non-EnvDrop form divides
    by the GLOBAL weight sum, curriculum.py:301 -> `allreduce_scalar(weight_rows.sum())`)."""
    return torch.dot(weight_rows.to(per_sample_loss.dtype), per_sample_loss)


def clip_grad_norm_groups(groups: Sequence[Sequence[torch.nn.Parameter]], max_norm: float) -> List[torch.Tensor]:
    """trainer.py:425-426 clips encoder and decoder separately (norm 40 each); applied AFTER the all-reduce
    so every replica computes the same scale."""
    return [torch.nn.utils.clip_grad_norm_(list(g), max_norm) for g in groups]
