#!/usr/bin/env python3
"""bench.py -- EnvDrop agent training steps/sec on MI355X (BASELINE.json metric, config 1).

One "step" = one agent training iteration of the reference's EnvDrop IL path
(trainer.py:411-427 with feedback="teacher"): instruction encoder forward, T teacher-forced decoder steps
with the in-place candidate mask + cross-entropy (envdrop.py:151-179), `ml_loss * ML_WEIGHT / B`, full
backward, gradient all-reduce (N > 1), clip-norm 40 on encoder and decoder, RMSprop step.  Batch 64 episodes
per GPU, 36 x (2048+128) view features, <= 80 instruction tokens, synthetic data, inputs
resident in HBM before the timed region; dropout ON (training mode).

    python bench.py                       # N=1, prints ONE JSON line
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

The `cpu_baseline` leg times the CPU oracle (oracle/torch_port.py, kind "port") on a bounded sample of the same
workload on rank 0 at N=1 only.  The `roofline` leg replays the K timed steps with per-kernel HIP-event timers
(vln_prof_*), and reports the kernel with the largest total time.
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


def make_tape(B, L, T, C_max, seed, vocab=992, V=36, IMG=2048, ANG=128):
    """Synthetic episode batch (BASELINE.md §3 / SURVEY.md §8d), CPU tensors.  Features are defined the way the
    reference's environment builds them (common_env.py:272,287-291,307-308): a per-viewpoint ResNet table
    [T*B viewpoints, 36 views, 2048] (post-ReLU, non-negative), the agent's viewIndex selecting the static angle
    table, and each candidate = (view of the current panorama, its relative heading/elevation).  The explicit
    img/cand tensors of every step are materialised from those for the tensor path and the CPU baseline."""
    import math
    g = torch.Generator().manual_seed(seed)
    F = IMG + ANG
    lens = torch.sort(torch.randint(8, L + 1, (B,), generator=g), descending=True).values
    lens[0] = L
    tokens = torch.zeros(B, L, dtype=torch.long)
    for i, n in enumerate(lens.tolist()):
        tokens[i, 0] = 3                                        # <BOS>
        tokens[i, 1:n - 1] = torch.randint(4, vocab, (n - 2,), generator=g)
        tokens[i, n - 1] = 2                                    # <EOS>
    seq_mask = tokens == 0

    def angle_feat(h, e):                                       # utils/misc.py:285-293
        return torch.stack([h.sin(), h.cos(), e.sin(), e.cos()], -1).repeat_interleave(ANG // 4, dim=-1)

    import vln_amd
    loc_table = vln_amd.staging.loc_embedding_table(ANG, V)    # static location embeddings [V, V, ANG], misc.py:296-317

    table = torch.randn(T * B, V, IMG, generator=g).abs() * 0.5
    T_i = torch.randint(min(4, T), T + 1, (B,), generator=g)
    T_i[0] = T
    steps = []
    for t in range(T):
        rows = torch.arange(B) + t * B
        vidx = torch.randint(0, V, (B,), generator=g).int()
        ncand = torch.randint(3, C_max + 1, (B,), generator=g)      # candidates incl. the STOP slot
        Ct = int(ncand.max())
        cmask = torch.arange(Ct)[None, :] >= ncand[:, None]
        real = torch.arange(Ct)[None, :] < (ncand - 1)[:, None]     # STOP slot + padding are all-zero rows
        crow = torch.where(real, rows[:, None].expand(B, Ct), torch.full((B, Ct), -1))
        cview = torch.randint(0, V, (B, Ct), generator=g).int()
        chead = (torch.rand(B, Ct, generator=g) - 0.5) * 6.0
        celev = (torch.rand(B, Ct, generator=g) - 0.5) * 1.04
        img = torch.cat((table[rows], loc_table[vidx.long()]), -1)
        cand = torch.cat((table[rows[:, None].expand(B, Ct), cview.long()], angle_feat(chead, celev)), -1) * real[..., None]
        ended = t >= T_i
        tgt = torch.where(t == T_i - 1, ncand - 1, (torch.rand(B, generator=g) * (ncand - 1).float()).long())
        tgt = torch.where(ended, torch.full_like(tgt, -1), tgt)
        ah = torch.rand(B, generator=g) * 6.283 - 3.1415
        steps.append(dict(img=img, cand=cand, cand_mask=cmask, angle=angle_feat(ah, torch.zeros(B)), target=tgt, rows=rows,
                          vidx=vidx, crow=crow, cview=cview, chead=chead, celev=celev))
    return dict(tokens=tokens, lengths=lens, seq_mask=seq_mask, steps=steps, table=table, B=B, L=L, T=T, IMG=IMG, ANG=ANG)


def tape_to(tape, dev, store_dtype=None, host_dtype=None):
    """Device copy.  With `store_dtype` the ResNet table becomes a resident DeviceFeatureStore and the per-step img/cand
    tensors are NOT uploaded (a step only needs its index vectors).  With `host_dtype` the per-step img/cand tensors stay on
    the HOST, pinned, in that dtype (fp32 = what the reference's ImageFeatures holds, utils/misc.py:253-279; bf16 = converted
    once at load time): every step then pays its H2D copy (PCIe-inclusive mode, never the headline value)."""
    skip = ("steps", "table")
    out = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in tape.items() if k not in skip}
    out["lengths32"] = tape["lengths"].to(dev, torch.int32)
    drop = ("img", "cand") if (store_dtype is not None or host_dtype is not None) else ()
    out["steps"] = [{k: v.to(dev) for k, v in s.items() if k not in drop} for s in tape["steps"]]
    if host_dtype is not None:
        for so, si in zip(out["steps"], tape["steps"]):
            so["img_host"] = si["img"].to(host_dtype).contiguous().pin_memory()
            so["cand_host"] = si["cand"].to(host_dtype).contiguous().pin_memory()
    if store_dtype is not None:
        import vln_amd
        out["store"] = vln_amd.DeviceFeatureStore(tape["table"], device=dev, dtype=store_dtype, angle_size=tape["ANG"])
    return out


class GpuAgent:
    """The caller side of the drop-in modules: the reference's rollout/optimizer sequence for IL."""

    def __init__(self, vln, dev, dtype, world, arena=False, rollout_ce=True, side_gather=False):
        self.vln, self.world, self.dtype = vln, world, dtype
        # (A/B option, off by default: measured slower) the step's feature gather reads only the resident table + index
        # vectors, so it can be issued on a side stream beside the encoder / the previous step's kernels
        self.side = torch.cuda.Stream(device=dev) if side_gather else None
        self.copy_stream, self._copy_fenced, self._host_drop = None, False, 0
        self.rollout_ce = rollout_ce
        self.enc = vln.EncoderLSTM(992, 256, 512, 0, 0.5, True, 1, compute_dtype=dtype).to(dev)
        self.dec = vln.EnvDropDecoder(512, 0.5, 0.3, 64, 128, 2176, compute_dtype=dtype).to(dev)
        self.enc.train(); self.dec.train()
        # teacher forcing: nothing reads the logits before the loss, so the decoder leaves them to be formed for the whole
        # rollout at once when losses.RolloutCE evaluates (one GEMM over steps x batch + one dot launch instead of two
        # launches on every step's dependent chain)
        self.dec.defer_logits = bool(rollout_ce)
        # trainer.py:380-381,423-427: RMSprop(lr) + clip_grad_norm(40) per module -- fused over flat buffers; the flat
        # gradient buffer doubles as the RCCL all-reduce bucket (optim.FusedRMSprop)
        self.opt = vln.optim.FusedRMSprop([list(self.enc.parameters()), list(self.dec.parameters())], lr=LR, clip_norm=CLIP)
        if world > 1:   # the decoder's 34.7 MB of gradients are final before the encoder's BPTT starts: reduce them under it
            self.dec.grads_ready_hook = lambda: self.opt.start_allreduce(1)
        # A training loop allocates the same sequence of buffers every iteration: with the arena they come back at the
        # same device addresses, so each decoder step (13 forward / 15 backward launches) replays as one hipGraph.
        self.arena = None
        self.use_arena(arena)

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
        self._copy_fenced = False
        self.vln.ops.set_arena(self.arena)
        if self.arena is not None:
            self.arena.begin()
        try:
            return self._iteration(tape)
        finally:
            self.vln.ops.set_arena(None)

    def _iteration(self, tape):
        B = tape["B"]
        if self.side is not None:      # once per iteration: the side stream's gathers write buffers last read two iterations ago
            self.side.wait_stream(torch.cuda.current_stream())
        if self.copy_stream is not None and self.arena is not None:     # same fence for the H2D copy stream (host features)
            self.copy_stream.wait_stream(torch.cuda.current_stream())
            self._copy_fenced = True
        self.opt.zero_grad()
        ctx, h_t, c_t = self.enc(tape["tokens"], tape["lengths32"])
        h_tilde = h_t
        terms = []
        ce = self.vln.losses.RolloutCE() if self.rollout_ce else None
        for s in tape["steps"]:
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
        loss.backward()
        self.opt.allreduce()
        self.opt.step()
        return loss


def cpu_baseline(tape, iters, P_enc, P_dec):
    """The CPU oracle driven identically (dropout sampled with bernoulli_ like nn.Dropout)."""
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

    t0 = time.perf_counter()
    one()                                   # warm-up
    warm = time.perf_counter() - t0
    if warm > 20.0:                         # pathological host (e.g. CPU quota): keep the bench bounded
        return warm, 1
    t0 = time.perf_counter()
    done = 0
    while done < iters and (time.perf_counter() - t0) < 20.0:
        one()
        done += 1
    return (time.perf_counter() - t0) / done, done


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
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL); 'gloo' only to smoke-test "
                                                      "the N>1 code path on a single-GPU box")
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
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    torch.manual_seed(2020)
    agent = GpuAgent(vln, dev, dtype, world, arena=not args.no_arena, rollout_ce=args.ce == "rollout",
                     side_gather=args.gather_stream == "side" and args.features == "store")
    tape_cpu = make_tape(args.batch, args.L, args.T, 8, seed=2020 + rank)   # weak scaling: 64 episodes per rank
    tape = tape_to(tape_cpu, dev, store_dtype=(dtype if args.features == "store" else None),
                   host_dtype={"host": torch.float32, "host-bf16": torch.bfloat16}.get(args.features))
    if args.features == "host-bf16" and dtype != torch.bfloat16:
        raise SystemExit("--features host-bf16 needs --dtype bf16")

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # Warm-up runs like the timed loop: iterations back to back, no per-iteration sync (the first time the host gets
    # many launches ahead of the GPU the runtime grows its in-flight pools: a one-time cost that belongs here).
    # Initialisation (not warm-up): the first four iterations build what later iterations only replay -- library load,
    # weight shadows, the arena's two buffer generations, one hipGraph capture and one step plan per decoder step and
    # generation.  Like a JIT compile this happens once per process, whatever W is.
    tw = time.perf_counter()
    for i in range(4):
        agent.iteration(tape)
        if i == 0:
            torch.cuda.synchronize()
            if rank == 0:
                print(f"[bench] first iteration (module init, captures): {(time.perf_counter() - tw) * 1e3:.1f} ms", file=sys.stderr, flush=True)
    torch.cuda.synchronize()
    for i in range(args.warmup):
        agent.iteration(tape)
    barrier()
    # Python's cyclic GC: a full pass over the (static) module/object graph costs tens of ms and would land in the
    # timed region at random; collect now and move the survivors out of the collector's reach.
    import gc
    gc.collect()
    gc.freeze()
    timed_out = int(agent.enc.persistent_status() != 0)
    if world > 1:                                 # the fallback below contains collectives: every rank takes it or none does
        flag = torch.tensor([timed_out], device=dev, dtype=torch.int32)
        torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MAX)
        timed_out = int(flag.item())
    if timed_out:                                 # a bounded in-kernel wait timed out during warm-up: fall back
        print("[bench] persistent recurrence reported a timeout; using per-step launches", file=sys.stderr, flush=True)
        lib.vln_set_persistent(0)
        agent.iteration(tape)
        barrier()
    marks = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        agent.iteration(tape)
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
            agent.iteration(tape)
        torch.cuda.synchronize()
        rows = read_prof(lib, nk) if rank == 0 else []
        if rank == 0:
            for k in range(nk):
                lib.vln_prof_enable(k, 0)
        if rows:
            rows.sort(key=lambda r: -r["ms"])
            top = rows[0]
            ach = top["bytes"] / (top["ms"] * 1e-3) / 1e9
            traffic = None
            tfile = os.path.join(ROOT, "profiles", "traffic.json")   # PMC-derived bytes/launch (scripts/pmc_traffic.py)
            if os.path.exists(tfile):
                t = json.load(open(tfile)).get(args.dtype, {}).get(top["kernel"])
                traffic = t["bytes_per_launch"] if t else None
            mfma = None                                               # MFMA issue slots busy, SQ_VALU_MFMA_BUSY_CYCLES pass
            mfile = os.path.join(ROOT, "profiles", "round1_mfma_util.json")      # (scripts/pmc_mfma.py), bf16 run
            if os.path.exists(mfile) and args.dtype == "bf16":
                mfma = json.load(open(mfile)).get(top["kernel"] + "_kernel", {}).get("mfma_util")
            roofline = dict(bound="hbm", kernel=top["kernel"], achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                            frac=round(ach / HBM_PEAK_GBS, 4), traffic=traffic, mfma_util=mfma,
                            avg_launch_us=round(top["ms"] * 1e3 / top["launches"], 2),
                            algo_bytes_per_launch=round(top["bytes"] / top["launches"]),
                            kernels=[dict(kernel=r["kernel"], launches_per_step=r["launches"] / args.steps,
                                          us_per_step=round(r["ms"] * 1e3 / args.steps, 1),
                                          GBps=round(r["bytes"] / (r["ms"] * 1e-3) / 1e9, 1)) for r in rows])
    if world > 1:
        torch.distributed.barrier()

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        ncores = min(usable_cores(), 64)
        torch.set_num_threads(ncores)
        print(f"[bench] cpu baseline on {ncores} threads ...", file=sys.stderr, flush=True)
        P_enc = {k: v.detach().float().cpu().clone() for k, v in agent.enc.state_dict().items()}
        P_dec = {k: v.detach().float().cpu().clone() for k, v in agent.dec.state_dict().items()}
        sec, done = cpu_baseline(tape_cpu, args.cpu_iters, P_enc, P_dec)
        cpu = dict(value=round(1.0 / sec, 4), unit="steps/s", cores=torch.get_num_threads(), kind="port",
                   sample=f"{done} iterations of the same tape (B={args.batch}, L={args.L}, T={args.T}), fp32, after 1 warm-up")

    if rank == 0:
        print(json.dumps({
            "metric": "agent train steps/sec (EnvDrop IL, batch 64/GPU, 36x2048 feats)", "value": round(value, 3),
            "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"envdrop_il_fwd_bwd_clip_rmsprop_B{args.batch}_L{args.L}_T{args.T}", "features": args.features,
                       "global_batch": args.batch * world, "seq_len": args.L, "decoder_steps": args.T,
                       "parallelism": f"dp{world}", "world_size": world,
                       "backend": (args.backend + ("=rccl" if args.backend == "nccl" else "")) if world > 1 else None},
            "roofline": roofline, "cpu_baseline": cpu}))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
