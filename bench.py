#!/usr/bin/env python3
"""bench.py -- EnvDrop agent training steps/sec on MI355X (BASELINE.json metric, config 1).

One "step" = one agent training iteration of the reference's EnvDrop IL path (trainer.py:411-427 with feedback="teacher"):
`vln_amd.trainers.EnvDropILIteration` -- instruction encoder forward, T teacher-forced decoder steps with the in-place candidate
mask + cross-entropy (envdrop.py:151-179), `ml_loss * ML_WEIGHT / B`, full backward, gradient all-reduce (N > 1), clip-norm 40 on
encoder and decoder, RMSprop step.  This file holds NO training logic: it builds the synthetic workload (vln_amd.synthetic,
BASELINE.md section 3), drives the package's iteration objects, times them and prints the line.  Batch 64 episodes per GPU,
36 x (2048+128) view features, <= 80 instruction tokens, dropout ON.
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
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import vln_amd as vln                                                    # noqa: E402  (no GPU call at import: the library loads lazily)
from vln_amd.batches import LiveBatch, LiveSteps                         # noqa: E402,F401
from vln_amd.synthetic import (N_TAPES, N_VIEWPOINTS, build_store, make_tape, materialize_step, tape_to)  # noqa: E402,F401
from vln_amd.trainers import (CLIP, HBM_PEAK_GBS, LR, ML_WEIGHT, EnvDropILIteration, EnvDropHostLoopIteration,  # noqa: E402,F401
                              read_kernel_timers, time_iterations)

PMC_FILE = "round6_pmc.json"       # profiles/<this>: the committed rocprofv3 --pmc passes (hash-checked against csrc/)

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
    agent = EnvDropILIteration(dev, dtype, world, arena=not args.no_arena, rollout_ce=args.ce == "rollout",
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
        agent.dec.grads_ready_hook = lambda: agent.opt.start_allreduce(1)      # (N = 1 rehearsal: the hook EnvDropILIteration sets for world > 1)
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
    store = build_store(dev, dtype, args.viewpoints)
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
            read_kernel_timers(lib)
        torch.cuda.synchronize()
        for _ in range(args.steps):
            iterate_eager()           # per-kernel event pairs ride on plain launches (a captured graph has none)
        torch.cuda.synchronize()
        rows = read_kernel_timers(lib) if rank == 0 else []
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
                         ("il_host_in_loop", lambda: secondary_host_in_loop(vln, dev, store, cpu_tapes, dtype, args, handshake=True)),
                         ("il_host_in_loop_stream_sync_per_step", lambda: secondary_host_in_loop(vln, dev, store, cpu_tapes, dtype, args)),
                         ("decoder_step_fwd_bwd", per_step),
                         ("phases", lambda: secondary_envdrop(vln, dev, store, cpu_tapes, dtype, "store", args, phases=True)),
                         ("fp32_ms_per_step", lambda: secondary_envdrop(vln, dev, store, cpu_tapes, torch.float32, "store", args, graph=use_graph)),
                         ("features_host_fp32_ms_per_step", lambda: secondary_envdrop(vln, dev, store, cpu_tapes, dtype, "host", args)),
                         ("batch128_ms_per_step", lambda: secondary_envdrop(
                             vln, dev, store, [make_tape(128, args.L, args.T, 8, seed=4040 + k, n_rows=store.N) for k in range(4)],
                             dtype, "store", args, graph=use_graph)),
                         # 256 episodes per GPU: the persistent recurrence in two passes (round 6; B > 128 used to fall back to 2 x 80 launches)
                         ("batch256_ms_per_step", lambda: secondary_envdrop(
                             vln, dev, store, [make_tape(256, args.L, args.T, 8, seed=4140 + k, n_rows=store.N) for k in range(4)],
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
                         ("speaker_teacher_forcing_B64_fp32", lambda: secondary_agents(dev, args, "speaker", store, dtype="fp32")),
                         ("speaker_teacher_forcing_B64_eager_launches", lambda: secondary_agents(dev, args, "speaker", store, dtype="bf16", graph=False))):
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
    (profiles/round6_pmc.json, written by scripts/pmc_stamp.py from separate FETCH_SIZE / WRITE_SIZE / MFMA_BUSY runs of this
    same command).  The file carries the hash of the kernel sources it was taken on: a mismatch means the numbers describe
    OTHER code, and they are refused (null) rather than quoted stale."""
    f = os.path.join(ROOT, "profiles", "round6_pmc.json")
    if not os.path.exists(f):
        return None, None, "no PMC passes committed for this round yet"
    d = json.load(open(f))
    if d.get("csrc_sha") != csrc_sha():
        return None, None, f"profiles/round6_pmc.json was taken on kernel sources {d.get('csrc_sha')}, this is {csrc_sha()}: refused"
    t = d.get("traffic", {}).get(dtype, {}).get(kernel)
    m = d.get("mfma_util", {}).get(dtype, {}).get(kernel)
    return (t["bytes_per_launch"] if t else None), m, f"profiles/round6_pmc.json, kernel sources {d['csrc_sha']}"


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
        ag = EnvDropILIteration(dev, dtype, 1, arena=not dropin, rollout_ce=not dropin)
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


def secondary_host_in_loop(vln, dev, store, cpu_tapes, dtype, args, steps=20, warmup=6, handshake=False):
    """ms per iteration of the headline workload with THE HOST IN THE LOOP (trainers.EnvDropHostLoopIteration; reference loop shape
    envdrop.py:151-220): per decoder step the step's index vectors come from the host (the observation the simulator just produced),
    the action a_t goes back to it and a fake environment steps on it.  handshake=False: eager launches + per-step graphs, one pinned
    H2D copy and one D2H + stream synchronisation per step; True: ONE hipGraph whose host turns are waits inside it."""
    torch.manual_seed(2020)
    st = store if store.table.dtype == dtype else vln.DeviceFeatureStore(store.table.to(dtype), device=dev, dtype=dtype)
    tapes = [tape_to(t, dev, store=st) for t in cpu_tapes]
    ls = LiveSteps(tapes, dev)
    it = EnvDropHostLoopIteration(dev, dtype, ls, st)
    for k in range(4):
        it.iteration(k)
    if handshake:
        it.capture(warmup=0)
        run = it.replay
    else:
        run = it.iteration
    k0 = [0]

    def one():
        run(k0[0]); k0[0] += 1
    ms = time_iterations(one, steps, warmup)
    return {"ms_per_step": round(ms, 3), "per_step_host_round_trips": len(tapes[0]["steps"]), "action_mismatches": it.mismatches,
            "how": ("ONE hipGraph: per step an in-graph wait that also pulls the step's index vectors from pinned memory (vln_host_wait_fetch), "
                    "decoder step (logits + CE in the step), a_t stored to pinned words the host polls + fake-env host step" if handshake else
                    "per step: pinned H2D of the step's index vectors, decoder step (per-step hipGraph, logits in the step, CE per step), "
                    "D2H of a_t + fake-env host step") + "; features from the resident table"}


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


def secondary_agents(dev, args, which, store, dtype=None, read_actions=True, graph=True):
    """The other BASELINE workloads (scripts/bench_agents.py builds their synthetic batches; the iterations are vln_amd.trainers')."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import bench_agents as W
    # warm-up: the first iterations of a workload in a process grow the allocator's pools and load its kernels' code objects;
    # with 8 of them the Self-Monitor number read 5.7 ms against 5.05 ms for a second run in the same process
    W.configure(steps=20, warmup=30, dtype=dtype or args.dtype, arena=False, device=dev, graph=graph)
    W.args.roofline = bool(which == "a2c" and read_actions == "handshake")     # the cfg3 entry carries its own roofline block
    import gc
    gc.collect()
    gc.freeze()                         # the bench's own objects (agent, tapes, store) out of the cyclic collector's way, as in the timed loop
    try:
        r = W.run_a2c(T_rl=35, store=store, read_actions=read_actions) if which == "a2c" else (W.run_follower() if which == "follower" else
                                                                    (W.run_speaker() if which == "speaker" else W.run_monitor()))
    finally:
        gc.unfreeze()
    out = {"workload": r["workload"], "ms_per_iteration": r["ms_per_iteration"], "dtype": r.get("dtype")}
    for k in ("iteration", "per_step_action_read", "roofline"):       # a2c: how the iteration was issued, that the host read every sampled action, its dominant kernel
        if k in r:
            out[k] = r[k]
    return out


if __name__ == "__main__":
    main()
