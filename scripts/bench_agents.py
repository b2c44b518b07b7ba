#!/usr/bin/env python3
"""Timing of the OTHER configured workloads of BASELINE.json on one MI355X (not the headline metric: `bench.py` is).  The
iterations themselves are the package's (vln_amd.trainers); this file builds the synthetic batches and times them:

  monitor  (config 2)  trainers.SelfMonitorIteration   B=128, L=80, T teacher-forced steps, one Adam
  follower (config 0's model, GPU batch)  trainers.FollowerIteration   B=64, L=80, T steps, CE mean, two Adam
  speaker  (config 4's back-translation model)  trainers.SpeakerIteration   B=64, teacher forcing, clip + two Adam
  a2c      (config 3, one rank)  trainers.EnvDropA2CIteration   IL (teacher, T=7) + RL (sampled, T<=35, A2C with the critic)

Synthetic data of BASELINE.md's shapes, features resident in HBM; prints one JSON line per workload.
    python scripts/bench_agents.py [monitor|follower|speaker|a2c|all] [--steps K] [--warmup W] [--dtype bf16|fp32] [--T-rl 35]
(bench.py imports the run_* functions for the secondary numbers of its JSON line)
"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vln_amd as vln
from vln_amd import synthetic, trainers


class _Args:          # defaults when imported (bench.py's secondary numbers); overwritten by the command line below
    steps, warmup, dtype, arena, graph = 30, 8, "bf16", False, True


args = _Args()
dev = torch.device("cuda:0")
dt = torch.bfloat16
F = 2176


def configure(steps=30, warmup=8, dtype="bf16", arena=False, device=None, graph=True):
    global dt, dev
    args.steps, args.warmup, args.dtype, args.arena, args.graph = steps, warmup, dtype, arena, graph
    dt = torch.bfloat16 if dtype == "bf16" else torch.float32
    if device is not None:
        dev = device


def with_arena(fn):
    """Per-iteration buffers from an address-stable arena (ops.RolloutArena) instead of torch.empty."""
    arena = vln.ops.RolloutArena()

    def run():
        vln.ops.set_arena(arena); arena.begin()
        try:
            fn()
        finally:
            vln.ops.set_arena(None)
    return run


def timed(fn):
    return trainers.time_iterations(fn, args.steps, args.warmup)


def _instructions(g, B, L, vocab=992):
    tokens = torch.randint(4, vocab, (B, L), generator=g)
    lens = torch.sort(torch.randint(8, L + 1, (B,), generator=g), descending=True).values; lens[0] = L
    for i, n in enumerate(lens.tolist()):
        tokens[i, n:] = 0
    tokens = tokens.to(dev)
    return dict(tokens=tokens, lens32=lens.to(dev, torch.int32), seq_mask=tokens == 0)


def monitor_batch(B=128, L=80, T=7, C=8, seed=2020):
    g = torch.Generator().manual_seed(seed)
    batch = _instructions(g, B, L)
    steps = []
    for t in range(T):
        ncand = torch.randint(3, C + 1, (B,), generator=g)
        cand = (torch.randn(B, C, F, generator=g).abs() * 0.5)
        cmask = torch.arange(C)[None, :] >= ncand[:, None]
        cand = cand * (~cmask)[..., None]
        tgt = (torch.rand(B, generator=g) * ncand.float()).long()
        start = torch.rand(B, generator=g) * 15 + 4
        cur = (start - torch.rand(B, generator=g) * start).clamp_min(0.2)
        steps.append(dict(cand=cand.to(dev), cmask=cmask.to(dev), target=tgt.to(dev), start=start.to(dev), cur=cur.to(dev),
                          ended=(torch.rand(B, generator=g) < 0.1 * t).to(dev)))
    batch["steps"] = steps
    return batch


def run_monitor(B=128, L=80, T=7, C=8):
    use_graph = bool(args.graph and not getattr(args, "python_step", False) and not args.arena)
    it = trainers.SelfMonitorIteration(dev, dt, max_len=L, graph=use_graph, rollout_wgrads=not getattr(args, "per_step_wgrads", False))
    it.dec.c_step = not getattr(args, "python_step", False)
    it.dec.merge_projections = not getattr(args, "two_bn_mlp_calls", False)
    it.load(monitor_batch(B, L, T, C))
    run = it.iteration
    if args.arena:
        run = with_arena(run)
    if it.clock is not None:
        it.capture()
        run = it.replay
    ms = timed(run)
    return dict(workload=f"self_monitor_il_B{B}_L{L}_T{T}_adam" + ("_arena" if args.arena else "") + ("_graph" if it.clock is not None else ""),
                ms_per_iteration=round(ms, 3), iterations_per_s=round(1e3 / ms, 2), dtype=args.dtype)


def follower_batch(B=64, L=80, T=7, C=8, seed=2020):
    g = torch.Generator().manual_seed(seed)
    batch = _instructions(g, B, L)
    steps = []
    for t in range(T):
        ncand = torch.randint(3, C + 1, (B,), generator=g)
        cmask = torch.arange(C)[None, :] >= ncand[:, None]
        cand = torch.randn(B, C, F, generator=g).abs() * 0.5 * (~cmask)[..., None]
        img = torch.randn(B, 36, F, generator=g).abs() * 0.5
        tgt = (torch.rand(B, generator=g) * ncand.float()).long()
        steps.append(dict(img=img.to(dev), cand=cand.to(dev), cmask=cmask.to(dev), target=tgt.to(dev)))
    batch["steps"] = steps
    return batch


def run_follower(B=64, L=80, T=7, C=8, fused=True):
    """Speaker-Follower agent (BASELINE config 0's model at a GPU batch): trainers.FollowerIteration."""
    use_graph = bool(args.graph and fused and not getattr(args, "python_step", False) and not args.arena)
    it = trainers.FollowerIteration(dev, dt, graph=use_graph, fused=fused, rollout_wgrads=not getattr(args, "per_step_wgrads", False))
    it.dec.c_step = not getattr(args, "python_step", False)
    it.load(follower_batch(B, L, T, C))
    run = it.iteration
    if args.arena:
        run = with_arena(run)
    if it.clock is not None:
        it.capture()
        run = it.replay
    ms = timed(run)
    return dict(workload=f"follower_il_B{B}_L{L}_T{T}_adam" + ("" if fused else "_operator_path") + ("_graph" if it.clock is not None else ""),
                ms_per_iteration=round(ms, 3), iterations_per_s=round(1e3 / ms, 2), dtype=args.dtype)


def speaker_batch(B=64, Lp=7, Lw=80, V=36, vocab=992, seed=2020):
    g = torch.Generator().manual_seed(seed)
    can = (torch.randn(B, Lp, F, generator=g).abs() * 0.5).to(dev)
    img = (torch.randn(B, Lp, V, F, generator=g).abs() * 0.5).to(dev)
    lengths = torch.randint(3, Lp + 1, (B,), generator=g); lengths[0] = Lp
    wl = torch.randint(8, Lw + 1, (B,), generator=g); wl[0] = Lw
    insts = torch.zeros(B, Lw, dtype=torch.long)
    for b in range(B):
        n = int(wl[b])
        insts[b, 0] = 3; insts[b, 1:n - 1] = torch.randint(4, vocab, (n - 2,), generator=g); insts[b, n - 1] = 2
    return dict(can=can, img=img, lengths=lengths, insts=insts.to(dev))


def run_speaker(B=64, Lp=7, Lw=80, V=36, vocab=992):
    """The speaker's training iteration (agent/speaker.py:75-87) at the configured size: trainers.SpeakerIteration."""
    it = trainers.SpeakerIteration(dev, dt, vocab=vocab, graph=args.graph)
    batch = speaker_batch(B, Lp, Lw, V, vocab)
    if args.graph:
        it.load(batch)
        it.capture()
        ms = timed(it.replay)
    else:
        ms = timed(lambda: it.iteration(batch))
    return dict(workload=f"speaker_teacher_forcing_B{B}_Lp{Lp}_Lw{Lw}_adam" + ("_graph" if args.graph else ""), ms_per_iteration=round(ms, 3),
                iterations_per_s=round(1e3 / ms, 2), dtype=args.dtype)


def build_a2c(B=64, L=80, T_il=7, T_rl=10, C=8, store=None, graph=None, read_actions=True, seed=2020, chain_il=True):
    """The cfg3 iteration object over one synthetic tape of T_rl steps (trainers.EnvDropA2CIteration)."""
    graph = args.graph if graph is None else graph
    if store is None:
        tape = synthetic.tape_to(synthetic.make_tape(B, L, T_rl, C, seed), dev, store_dtype=dt)
    else:
        tape = synthetic.tape_to(synthetic.make_tape(B, L, T_rl, C, seed, n_rows=store.N), dev, store=store)
    return trainers.EnvDropA2CIteration(dev, dt, tape, T_il=T_il, graph=graph, read_actions=read_actions, chain_il=chain_il,
                                        sampler_in_step=not getattr(args, "separate_sampler", False),
                                        per_step_sampler=getattr(args, "per_step_sampler", False),
                                        chain_backward=not getattr(args, "no_chain_backward", False))


def run_a2c(B=64, L=80, T_il=7, T_rl=10, C=8, store=None, graph=None, read_actions=True, seed=2020, chain_il=True):
    """EnvDrop IL (teacher-forced rollout, T_il steps) + RL (sampled rollout up to T_rl steps, A2C with the critic) per optimizer step
    (trainer.py:411-427): ms per iteration of trainers.EnvDropA2CIteration, eager or captured (read_actions: see that class)."""
    graph = args.graph if graph is None else graph
    it = build_a2c(B, L, T_il, T_rl, C, store, graph, read_actions, seed, chain_il)
    if graph:
        for _ in range(3):
            it.iteration()
        run = it.capture()
    else:
        run = it.iteration
    ms = timed(run)
    roof = None
    if getattr(args, "roofline", False):
        # the dominant kernel of THIS workload against the HBM roofline: per-kernel hip-event timers ride on plain launches, so five
        # iterations are issued eagerly (no iteration graph, no per-step graphs while the timers are on)
        roof = trainers.kernel_roofline(it.iteration, 5, per="iteration", top=6)
    return dict(workload=f"envdrop_il_T{T_il}_plus_a2c_T{T_rl}_B{B}_L{L}_rmsprop_arena", ms_per_iteration=round(ms, 3), roofline=roof,
                iterations_per_s=round(1e3 / ms, 2), dtype=args.dtype, **it.describe())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("which", nargs="?", default="all")
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--T-rl", type=int, default=35, help="cap of the sampled rollout (reference: MAX_EPISODE_LEN 35)")
    ap.add_argument("--arena", action="store_true", help="monitor / follower: per-iteration buffers from ops.RolloutArena")
    ap.add_argument("--python-step", action="store_true", help="monitor: the step's launches driven from Python (functional.MonitorCoreFn) "
                                                               "instead of one C call each way")
    ap.add_argument("--separate-sampler", action="store_true", help="a2c: (A/B) candidate dots, draw and the action's D2H copy as three launches instead of inside the step's last launch")
    ap.add_argument("--no-chain-backward", action="store_true", help="a2c: (A/B) the sampled rollout's step backwards not chained")
    ap.add_argument("--per-step-sampler", action="store_true", help="a2c: losses.sample_action per step (A/B) instead of losses.RolloutSampler")
    ap.add_argument("--no-graph", action="store_true", help="monitor / follower / a2c: eager launches instead of one hipGraph (a2c: a sequence of graph segments) per iteration")
    ap.add_argument("--no-action-read", action="store_true", help="a2c: (A/B) the sampled actions never leave the device")
    ap.add_argument("--roofline", action="store_true", help="a2c: also time the workload's kernels with hip events (five eager iterations) and report its dominant kernel against the HBM roofline")
    ap.add_argument("--handshake", action="store_true", help="a2c: the iteration as ONE hipGraph whose per-step host turns are waits inside it (graphs.HandshakeIterationGraph); the host polls every action and releases the next step")
    ap.add_argument("--poll-actions", action="store_true", help="a2c: the host spins on the pinned action words instead of synchronising the stream after every step")
    ap.add_argument("--no-chain-il", action="store_true", help="a2c: (A/B) the teacher-forced rollout's steps not chained")
    ap.add_argument("--tunable", action="append", default=[], metavar="ID=VALUE", help="(A/B) vln_set_tunable(ID, VALUE) before anything runs")
    ap.add_argument("--two-bn-mlp-calls", action="store_true", help="monitor: the BN-MLP called twice per step like the reference (A/B) "
                                                                    "instead of once on both batches (MonitorDecoder.merge_projections)")
    ap.add_argument("--fused-only", action="store_true", help="follower: skip the operator-by-operator A/B run")
    ap.add_argument("--per-step-wgrads", action="store_true", help="monitor / follower: parameter gradients in every step's backward "
                                                                   "(A/B) instead of once per rollout (functional.RolloutWgrads)")
    a = ap.parse_args()
    configure(a.steps, a.warmup, a.dtype, a.arena, graph=not a.no_graph)
    args.python_step = a.python_step
    args.per_step_sampler = a.per_step_sampler
    args.separate_sampler, args.no_chain_backward, args.roofline = a.separate_sampler, a.no_chain_backward, a.roofline
    args.two_bn_mlp_calls, args.per_step_wgrads = a.two_bn_mlp_calls, a.per_step_wgrads
    for tv in a.tunable:
        tid, val = tv.split("=")
        vln._lib.check(vln._lib.load().vln_set_tunable(int(tid), int(val)), "vln_set_tunable")
    if a.which in ("monitor", "all"):
        print(json.dumps(run_monitor()), flush=True)
        print("grad sinks [in place, via autograd]:", vln.functional.GRAD_IN_PLACE_STATS, file=sys.stderr)
    if a.which in ("follower", "all"):
        print(json.dumps(run_follower()), flush=True)
        if not a.fused_only:
            print(json.dumps(run_follower(fused=False)), flush=True)
    if a.which in ("speaker", "all"):
        print(json.dumps(run_speaker()), flush=True)
    if a.which in ("a2c", "all"):
        print(json.dumps(run_a2c(T_rl=a.T_rl, read_actions=("handshake" if a.handshake else "poll" if a.poll_actions else not a.no_action_read), chain_il=not a.no_chain_il)), flush=True)


if __name__ == "__main__":
    main()
