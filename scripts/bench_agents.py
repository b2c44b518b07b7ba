#!/usr/bin/env python3
"""Timing of the OTHER configured workloads of BASELINE.json on one MI355X (not the headline metric: `bench.py` is):

  monitor  (config 2)  Self-Monitoring agent + progress-monitor head, B=128, L=80 (uni-directional encoder, H=512,
                       MLP 1024), T teacher-forced steps, loss = CE at t=0 then 0.5*MSE + 0.5*CE (monitor.py:146-165),
                       one Adam over encoder+decoder (trainer.py:219-222)
  follower (config 0's model, GPU batch)  Speaker-Follower agent, B=64, L=80, T teacher-forced steps, CE mean, two Adam
  a2c      (config 3, one rank)  EnvDrop IL (teacher, T=7) + RL (sampled actions, T=10, A2C with the critic) mixed loss
                       at B=64 per GPU, clip 40 + RMSprop over encoder / decoder / critic

Synthetic data of BASELINE.md's shapes, features resident in HBM; prints one JSON line per workload.
    python scripts/bench_agents.py [monitor|follower|speaker|a2c|all] [--steps K] [--warmup W] [--dtype bf16|fp32] [--T-rl 35]
(bench.py imports run_monitor / run_a2c for the secondary numbers of its JSON line)
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import vln_amd as vln


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


def graphed(fn, clock):
    """The whole iteration (forward, losses, backward, optimizer) as ONE hipGraph (graphs.IterationGraph): `fn` ticks `clock`
    first; the teacher-forced batch sits at fixed addresses."""
    for _ in range(3):
        fn()
    g = vln.IterationGraph(fn, clock).capture()
    return g.replay


def timed(fn):
    for _ in range(args.warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / args.steps * 1e3


def run_monitor(B=128, L=80, T=7, C=8):
    g = torch.Generator().manual_seed(2020)
    enc = vln.EncoderLSTM(992, 256, 512, 0, 0.5, False, 1, compute_dtype=dt).to(dev).train()
    dec = vln.MonitorDecoder(512, 0.5, L, (1024,), F, F, compute_dtype=dt).to(dev).train()      # MLP_HIDDEN (1024,): configs/monitor/selfmonitor_config.yaml:45
    dec.c_step = not getattr(args, "python_step", False)
    dec.merge_projections = not getattr(args, "two_bn_mlp_calls", False)
    opt = vln.optim.FusedAdam([list(enc.parameters()) + list(dec.parameters())], lr=1e-4)
    clock = None
    if args.graph and dec.c_step and not args.arena:
        clock = vln.DeviceClock(dev).attach(enc, dec)
        opt.use_clock(clock)
    tokens = torch.randint(4, 992, (B, L), generator=g)
    lens = torch.sort(torch.randint(8, L + 1, (B,), generator=g), descending=True).values; lens[0] = L
    for i, n in enumerate(lens.tolist()):
        tokens[i, n:] = 0
    tokens, lens32 = tokens.to(dev), lens.to(dev, torch.int32)
    seq_mask = tokens == 0
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

    def it():
        if clock is not None:
            clock.tick()
        opt.zero_grad()
        ctx, h, c = enc(tokens, lens32)
        a_prev = torch.zeros(B, F, device=dev)
        loss = 0.0
        for t, s in enumerate(steps):
            (logit, prog), (h, c), _ = dec(None, a_prev, s["cand"], h, c, ctx, seq_mask, s["cmask"])
            # monitor.py:146-165 in one launch each way (CE + progress target + MSE + the lambda mix)
            loss_t, _ = vln.losses.monitor_mixed_loss(logit, s["target"], s["cmask"], prog, s["start"], s["cur"], s["ended"], t, 0.5)
            loss = loss + loss_t
            a_prev = s["cand"][torch.arange(B, device=dev), s["target"]].detach()
        loss.backward()
        opt.step()

    if args.arena:
        it = with_arena(it)
    if clock is not None:
        it = graphed(it, clock)
    ms = timed(it)
    return dict(workload=f"self_monitor_il_B{B}_L{L}_T{T}_adam" + ("_arena" if args.arena else "") + ("_graph" if clock is not None else ""),
                ms_per_iteration=round(ms, 3), iterations_per_s=round(1e3 / ms, 2), dtype=args.dtype)


def run_follower(B=64, L=80, T=7, C=8, fused=True):
    """Speaker-Follower agent (BASELINE config 0's model at a GPU batch): 2-layer bi-directional encoder (E=300, H=256),
    AttnDecoderLSTM over 36 x 2176 views, CE mean per step (follower.py:62,123-139), two Adam instances (trainer.py:65-67)."""
    g = torch.Generator().manual_seed(2020)
    enc = vln.EncoderLSTM(992, 300, 256, 0, 0.5, True, 2, compute_dtype=dt).to(dev).train()
    dec = vln.AttnDecoderLSTM(256, 0.5, F, F, compute_dtype=dt).to(dev).train()
    dec.fused_step = fused
    dec.c_step = not getattr(args, "python_step", False)
    opt_e = vln.optim.FusedAdam([list(enc.parameters())], lr=1e-4)
    opt_d = vln.optim.FusedAdam([list(dec.parameters())], lr=1e-4)
    clock = None
    if args.graph and fused and dec.c_step and not args.arena:
        clock = vln.DeviceClock(dev).attach(enc, dec)
        opt_e.use_clock(clock); opt_d.use_clock(clock)
    tokens = torch.randint(4, 992, (B, L), generator=g)
    lens = torch.sort(torch.randint(8, L + 1, (B,), generator=g), descending=True).values; lens[0] = L
    for i, n in enumerate(lens.tolist()):
        tokens[i, n:] = 0
    tokens, lens32 = tokens.to(dev), lens.to(dev, torch.int32)
    seq_mask = tokens == 0
    steps = []
    for t in range(T):
        ncand = torch.randint(3, C + 1, (B,), generator=g)
        cmask = torch.arange(C)[None, :] >= ncand[:, None]
        cand = torch.randn(B, C, F, generator=g).abs() * 0.5 * (~cmask)[..., None]
        img = torch.randn(B, 36, F, generator=g).abs() * 0.5
        tgt = (torch.rand(B, generator=g) * ncand.float()).long()
        steps.append(dict(img=img.to(dev), cand=cand.to(dev), cmask=cmask.to(dev), target=tgt.to(dev)))

    def it():
        if clock is not None:
            clock.tick()
        opt_e.zero_grad(); opt_d.zero_grad()
        ctx, h, c = enc(tokens, lens32)
        a_prev = torch.zeros(B, F, device=dev)
        loss = 0.0
        for s in steps:
            logit, (h, c), _ = dec(s["img"], a_prev, s["cand"], h, c, ctx, seq_mask)
            loss = loss + vln.losses.masked_cross_entropy(logit, s["target"], s["cmask"], "mean")
            a_prev = s["cand"][torch.arange(B, device=dev), s["target"]].detach()
        loss.backward()
        opt_e.step(); opt_d.step()

    if args.arena:
        it = with_arena(it)
    if clock is not None:
        it = graphed(it, clock)
    ms = timed(it)
    return dict(workload=f"follower_il_B{B}_L{L}_T{T}_adam" + ("" if fused else "_operator_path") + ("_graph" if clock is not None else ""), ms_per_iteration=round(ms, 3),
                iterations_per_s=round(1e3 / ms, 2), dtype=args.dtype)


def run_speaker(B=64, Lp=7, Lw=80, V=36, vocab=992):
    """The speaker's training iteration (agent/speaker.py:75-87: teacher_forcing -> backward -> clip 40 per module -> two Adam) at
    the configured size: RNN_DIM 512, bidirectional encoder over paths of up to 7 viewpoints x 36 x 2176 views, WEMB 256, vocabulary
    992, 80-token instructions, DROPOUT 0.6 / FEAT_DROPOUT 0.3 on."""
    g = torch.Generator().manual_seed(2020)
    enc = vln.SpeakerEncoder(F, 512, 0.6, True, 128, 0.3, compute_dtype=dt).to(dev).train()
    dec = vln.SpeakerDecoder(vocab, 256, 0, 512, 0.6, compute_dtype=dt).to(dev).train()
    spk = vln.Speaker(enc, dec)
    opt_e = vln.optim.FusedAdam([list(enc.parameters())], lr=1e-4, clip_norm=40.0)
    opt_d = vln.optim.FusedAdam([list(dec.parameters())], lr=1e-4, clip_norm=40.0)
    can = (torch.randn(B, Lp, F, generator=g).abs() * 0.5).to(dev)
    img = (torch.randn(B, Lp, V, F, generator=g).abs() * 0.5).to(dev)
    lengths = torch.randint(3, Lp + 1, (B,), generator=g); lengths[0] = Lp
    wl = torch.randint(8, Lw + 1, (B,), generator=g); wl[0] = Lw
    insts = torch.zeros(B, Lw, dtype=torch.long)
    for b in range(B):
        n = int(wl[b])
        insts[b, 0] = 3; insts[b, 1:n - 1] = torch.randint(4, vocab, (n - 2,), generator=g); insts[b, n - 1] = 2
    insts = insts.to(dev)

    def it():
        opt_e.zero_grad(); opt_d.zero_grad()
        # the feature dropout works in place (units.py:322,331): a training loop hands over fresh feature tensors every batch
        loss = spk.teacher_forcing(can.clone(), img.clone(), lengths, insts, train=True)
        loss.backward()
        opt_e.step(); opt_d.step()

    ms = timed(it)
    return dict(workload=f"speaker_teacher_forcing_B{B}_Lp{Lp}_Lw{Lw}_adam", ms_per_iteration=round(ms, 3), iterations_per_s=round(1e3 / ms, 2),
                dtype=args.dtype)


def run_a2c(B=64, L=80, T_il=7, T_rl=10, C=8, store=None, graph=None, read_actions=True, build_only=False, seed=2020, chain_il=True):
    """EnvDrop IL (teacher-forced rollout, T_il steps) + RL (sampled rollout up to T_rl steps -- the reference caps episodes at
    MAX_EPISODE_LEN = 35, configs/envdrop/envdrop_config.yaml:31 -- A2C with the critic, envdrop.py:186-264) per optimizer step
    (trainer.py:411-427); one RMSprop over encoder / decoder / critic, clip 40 on encoder and decoder only (:425-426).

    read_actions: the reference's loop shape for the sampled rollout (envdrop.py:196-206): after EVERY step the sampled action goes
    to the host (D2H into pinned memory, stream synchronize) where the simulator would take it -- here a host bookkeeping of the
    `ended` flags stands in for env.step -- before the next step is issued.  "poll" (graph form): the host spins on the pinned words
    (armed with -1 before the replay) instead of synchronising the stream: the wake-up of a synchronisation costs ~17 us per step.
    graph (default: args.graph): the iteration as graphs.SegmentedIterationGraph -- [graph: prologue, the IL rollout, the RL
    encoder, RL step 0 + draw + D2H of a_0] [host: wait, read a_0] [graph: RL step 1 ...] ... [graph: last step, critic, A2C loss,
    the backward of BOTH rollouts, clip + RMSprop]: T_rl + 1 graph launches per iteration instead of ~250 Python-driven calls; the
    dropout offsets and the draws' Philox offsets come from a runtime.DeviceClock."""
    graph = args.graph if graph is None else graph
    if store is None:
        cpu_tape = bench.make_tape(B, L, T_rl, C, seed)
        tape = bench.tape_to(cpu_tape, dev, store_dtype=dt)
    else:
        tape = bench.tape_to(bench.make_tape(B, L, T_rl, C, seed, n_rows=store.N), dev, store=store)
    enc = vln.EncoderLSTM(992, 256, 512, 0, 0.5, True, 1, compute_dtype=dt).to(dev).train()
    dec = vln.EnvDropDecoder(512, 0.5, 0.3, 64, 128, F, compute_dtype=dt).to(dev).train()
    cri = vln.Critic(512, 0.5).to(dev).train()
    opt = vln.optim.FusedRMSprop([list(enc.parameters()), list(dec.parameters()), list(cri.parameters())], lr=1e-4,
                                 clip_norm=[40.0, 40.0, 0.0])
    store = tape["store"]
    g = torch.Generator().manual_seed(7)
    rewards = [torch.randn(B, generator=g).sign().to(dev) for _ in range(T_rl)]
    lens_rl = torch.randint(4, T_rl + 1, (B,), generator=g)
    lens_rl[0] = T_rl
    masks = [(t < lens_rl).to(dev) for t in range(T_rl)]
    ended = (lens_rl < T_rl).to(dev)
    clock = vln.DeviceClock(dev).attach(enc, dec, cri) if graph else None
    a_host = torch.zeros(T_rl, B, dtype=torch.int64).pin_memory()
    a_np = a_host.numpy()         # the same pinned memory, for the polling form of the action read
    import ctypes as C_
    _d = C_.c_void_p()
    vln._lib.check(vln._lib.load().vln_host_device_pointer(a_host.data_ptr(), C_.byref(_d)), "vln_host_device_pointer")
    a_host_dev = int(_d.value)    # the device-visible address of the pinned action words (the step's draw stores there itself)
    in_step = not getattr(args, "separate_sampler", False) and not getattr(args, "per_step_sampler", False)
    poll = read_actions in ("poll", "handshake")
    host_ended = [0]              # what the stand-in for env.step keeps: episodes that chose STOP so far (read, never fed back)

    def gather_of(s):
        return (store, s["rows"], s["vidx"], s["crow"], s["cview"], s["chead"], s["celev"])

    st = {}                       # the sampled rollout's running state between two segments

    def il_rollout():
        ctx, h, c = enc(tape["tokens"], tape["lengths32"])
        ht = h
        dec.defer_logits = True   # teacher forcing: the logits are only needed by the loss (formed once per rollout)
        dec.chain_steps = chain_il   # ... and nothing reads a step's h_tilde but the next step: consecutive steps share launches
        ce = vln.losses.RolloutCE()
        for s in tape["steps"][:T_il]:
            logit, (h, c), ht = dec(s["angle"], None, None, ht, h, c, ctx, tape["seq_mask"], gather=gather_of(s))
            ce.add(logit, s["target"], s["cand_mask"])
        return ce.sum(scale=0.2 / B)

    def rl_begin():
        ctx, h, c = enc(tape["tokens"], tape["lengths32"])
        dec.defer_logits = False
        dec.chain_steps = False      # the sampled rollout reads every step's logits
        dec.chain_backward = not getattr(args, "no_chain_backward", False)     # ... but nothing except the next step consumes its h_tilde
        st.update(ctx=ctx, h=h, c=c, ht=h, hidden=[], logps=[], ents=[],
                  sampler=None if getattr(args, "per_step_sampler", False) else vln.losses.RolloutSampler(clock=clock))

    def rl_step(t):
        s = tape["steps"][t]
        if in_step and st["sampler"] is not None:
            # mask + softmax + draw + log-prob + entropy inside the step's logits launch; the action goes to the pinned words itself
            logit, (h, c), ht = dec(s["angle"], None, None, st["ht"], st["h"], st["c"], st["ctx"], tape["seq_mask"], gather=gather_of(s),
                                    sampler=(st["sampler"], s["cand_mask"], None, (a_host_dev + 8 * B * t) if read_actions else 0))
            st.update(h=h, c=c, ht=ht)
            st["hidden"].append(h)
            return
        logit, (h, c), ht = dec(s["angle"], None, None, st["ht"], st["h"], st["c"], st["ctx"], tape["seq_mask"], gather=gather_of(s))
        st.update(h=h, c=c, ht=ht)
        st["hidden"].append(h)
        if st["sampler"] is not None:
            a = st["sampler"].step(logit, s["cand_mask"])                       # envdrop.py:186-195 as one launch per step ...
        else:
            a, lp_a, en_a = vln.losses.sample_action(logit, s["cand_mask"])
            st["logps"].append(lp_a); st["ents"].append(en_a)
        if read_actions:
            a_host[t].copy_(a, non_blocking=True)                               # envdrop.py:198: cpu_a_t = a_t.cpu().numpy()

    def host_step(t):
        if poll and not torch.cuda.is_current_stream_capturing() and polling[0]:
            # the D2H copy of a_t is the segment's last node: the host spins on the pinned words (armed with -1 before the launch)
            # instead of paying a stream synchronisation's wake-up; sampled actions are >= 0
            row = a_np[t]
            while (row < 0).any():
                pass
            host_ended[0] = int((row == C - 1).sum())
        elif read_actions:
            torch.cuda.current_stream().synchronize()
            host_ended[0] = int((a_host[t] == C - 1).sum())                     # stand-in for env.step(cpu_a_t): the host reads the actions

    def rl_end():
        if st["sampler"] is not None:
            logps, ents = st["sampler"].stats()                                 # ... and ONE backward node for all steps
        else:
            logps, ents = st["logps"], st["ents"]
        hidden = st["hidden"]
        sl = tape["steps"][len(hidden) - 1]
        _, (last_h, _), _ = dec(sl["angle"], None, None, st["ht"], st["h"], st["c"], st["ctx"], tape["seq_mask"], gather=gather_of(sl))
        with torch.no_grad():
            last_v = cri(last_h).detach()
        # the critic is row-wise: V of all T steps in ONE call over (steps x batch) rows instead of T calls (the reference
        # loops `self.critic(hidden_states[t])`, envdrop.py:246 -- same function of the same rows)
        vals = list(cri(torch.cat(hidden, 0)).view(len(hidden), B).unbind(0))
        T = len(hidden)
        rl, _ = vln.losses.a2c_loss(logps, ents, vals, rewards[:T], masks[:T], last_v, ended, 0.9, "total")
        st.clear()
        return rl

    arena = vln.ops.RolloutArena()
    dec.step_graphs = True

    def in_arena(fn, begin=False):
        def run():
            vln.ops.set_arena(arena)
            if begin:
                arena.begin()
            try:
                return fn()
            finally:
                vln.ops.set_arena(None)
        return run

    def first():
        if clock is not None:
            clock.prologue(modules=(enc, dec))
        opt.zero_grad()               # no launch after a step(zero_grads=True): the update cleared the buffer while it read it
        st["il"] = il_rollout()
        rl_begin()
        rl_step(0)

    def last():
        il = st.pop("il")
        loss = il + rl_end()
        loss.backward()
        opt.step(zero_grads=True)
        return loss

    segs = [("graph", in_arena(first, begin=True)), ("host", lambda: host_step(0))]
    for t in range(1, T_rl):
        segs += [("graph", in_arena(lambda t=t: rl_step(t))), ("host", lambda t=t: host_step(t))]
    segs.append(("graph", in_arena(last)))

    def it():
        out = None
        for _, fn in segs:
            r = fn()
            out = r if r is not None else out
        return out

    polling = [False]

    def captured():
        if read_actions == "handshake":
            # ONE graph for the iteration: the host's turns are waits INSIDE it (graphs.HandshakeIterationGraph); the host polls the
            # pinned action words of step t, does its turn and releases step t + 1
            hg = vln.HandshakeIterationGraph(segs, clock).capture()
            polling[0] = True

            def run_h():
                a_np[:] = -1
                return hg.replay()
            return run_h
        sg = vln.SegmentedIterationGraph(segs, clock).capture()
        if not poll:
            return sg.replay
        polling[0] = True

        def run():
            a_np[:] = -1              # arm the pinned words (every replayed segment's copy overwrites its row)
            return sg.replay()
        return run

    if build_only:          # tests: (eager iteration, a function that captures and returns the replay, the state to compare)
        return it, captured, dict(opt=opt, enc=enc, dec=dec, cri=cri, a_host=a_host, clock=clock)
    if graph:
        for _ in range(3):
            it()
        run = captured()
    else:
        run = it
    ms = timed(run)
    roof = None
    if getattr(args, "roofline", False):
        # the dominant kernel of THIS workload against the HBM roofline: per-kernel hip-event timers ride on plain launches, so five
        # iterations are issued eagerly (no iteration graph, no per-step graphs while the timers are on)
        lib = vln._lib.load()
        nk = 0
        while lib.vln_prof_kernel_name(nk):
            nk += 1
        for k in range(nk):
            lib.vln_prof_enable(k, 1)
        bench.read_prof(lib, nk)
        torch.cuda.synchronize()
        n_it = 5
        for _ in range(n_it):
            it()
        torch.cuda.synchronize()
        rows = sorted(bench.read_prof(lib, nk), key=lambda r: -r["ms"])
        for k in range(nk):
            lib.vln_prof_enable(k, 0)
        if rows:
            top = rows[0]
            ach = top["bytes"] / (top["ms"] * 1e-3) / 1e9
            roof = dict(bound="hbm", kernel=top["kernel"], achieved=round(ach, 1), peak=bench.HBM_PEAK_GBS, unit="GB/s",
                        frac=round(ach / bench.HBM_PEAK_GBS, 4), traffic=None, avg_launch_us=round(top["ms"] * 1e3 / top["launches"], 2),
                        algo_bytes_per_launch=round(top["bytes"] / top["launches"]),
                        kernels=[dict(kernel=r["kernel"], launches_per_iteration=r["launches"] / n_it, us_per_iteration=round(r["ms"] * 1e3 / n_it, 1),
                                      GBps=round(r["bytes"] / (r["ms"] * 1e-3) / 1e9, 1)) for r in rows[:6]])
    return dict(workload=f"envdrop_il_T{T_il}_plus_a2c_T{T_rl}_B{B}_L{L}_rmsprop_arena", ms_per_iteration=round(ms, 3), roofline=roof,
                iteration=(("ONE hipGraph, the host's turns are waits inside it" if read_actions == "handshake" else f"{T_rl + 1} hipGraph segments") if graph else "per-step hipGraphs, Python-driven"),
                per_step_action_read=("host spins on the pinned action words" if (poll and graph) else bool(read_actions)), plan_hits=dec.plan_hits, arena_misses=arena.misses,
                iterations_per_s=round(1e3 / ms, 2), dtype=args.dtype)


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
    ap.add_argument("--no-grad-in-place", action="store_true", help="parameter gradients of the fused nodes through autograd's AccumulateGrad")
    a = ap.parse_args()
    configure(a.steps, a.warmup, a.dtype, a.arena, graph=not a.no_graph)
    args.python_step = a.python_step
    args.per_step_sampler = a.per_step_sampler
    args.separate_sampler, args.no_chain_backward, args.roofline = a.separate_sampler, a.no_chain_backward, a.roofline
    args.two_bn_mlp_calls = a.two_bn_mlp_calls
    for tv in a.tunable:
        tid, val = tv.split("=")
        vln._lib.check(vln._lib.load().vln_set_tunable(int(tid), int(val)), "vln_set_tunable")
    vln.functional.set_grad_in_place(not a.no_grad_in_place)
    vln.functional.set_rollout_wgrads(not a.per_step_wgrads and not a.no_grad_in_place)
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
