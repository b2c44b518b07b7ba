"""GPU: one training iteration captured as ONE hipGraph (graphs.IterationGraph + runtime.DeviceClock) must be the eager
iteration bit for bit -- losses, every gradient, every parameter and the optimizer state after several optimizer steps over
DIFFERENT episode batches, dropout ON (the replays must draw fresh masks: the offsets come from a device word the captured
tick launch bumps), with the step gathering its own features and with the rollout-wide gather as a captured branch."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def vln():
    import vln_amd
    vln_amd._lib.load()
    return vln_amd


def _run(vln, dtype, graph, branch, n_eager=2, n_more=4, near_wrap=False, segmented=False, calls=None, source="device", chain=True,
         prologue=True, ride=True, shape=(16, 24, 4, 6), ride_shadows=False):
    dev = torch.device(DEV)
    torch.manual_seed(77)
    store = vln.synthetic.build_store(dev, dtype, n_rows=300, seed=5)
    B_, L_, T_, C_ = shape
    tapes = [vln.synthetic.tape_to(vln.synthetic.make_tape(B_, L_, T_, C_, seed=500 + k, n_rows=store.N), dev, store=store) for k in range(5)]
    live = vln.LiveBatch(tapes, source=source)
    torch.manual_seed(78)
    ag = vln.trainers.EnvDropILIteration(dev, dtype, 1, arena=True)
    ag.use_live(live)
    ag.dec.chain_steps = chain
    ag.use_prologue = prologue
    ag.ride_shadows = ride_shadows                       # the decoder's weight shadows refreshed by the gather ride's passengers
    ag.dec.ride_wgrads = bool(ride and not segmented)    # as bench.py sets it for one GPU (bf16 mode only: the module checks)
    ag.clear_grads_in_step = True
    ag.enc.deterministic_embedding_grad = True           # float atomics would differ between two runs of the SAME path
    ag.rollout_gather = ag.gather_branch = branch == "branch"
    ag.ride_gather = branch == "ride"            # the gather as passengers of the encoder's recurrence launch
    ag.use_clock(store)
    if segmented:                  # the data-parallel form: three graph segments, the gradient exchange issued between them
        ag.segmented = True
        if calls is not None:      # stand-ins for the collectives: record WHEN the host segments run
            ag.opt.start_allreduce = lambda gi: calls.append(("start", gi))
            ag.opt.allreduce = lambda: calls.append(("finish",))
    out = []

    def record(loss):
        torch.cuda.synchronize()
        out.append((loss.detach().clone(), ag.opt.flat_p.clone(), ag.opt.sq.clone(), ag.opt.norms.clone()))

    for k in range(n_eager):
        record(ag.iteration(live.load(k)))
    if graph:
        ag.capture(live.live)
        assert ag.clock.host == n_eager * ag.clock.STRIDE           # the captured tick was not counted
    if near_wrap:
        # the recurrence's 24-bit launch sequence about to wrap (in a real run after ~260,000 iterations = minutes): two
        # iterations with tags at the top of the range, then the guard clears the exchange and restarts the sequence
        assert ag.clock._seqs
        for s in ag.clock._seqs:
            v = (1 << 24) - 5 * ag.clock.STRIDE
            s[0][s[1]] = v
            s[3] = v
    for k in range(n_eager, n_eager + n_more):
        live.load(k)
        record(ag.replay() if graph else ag.iteration(live.live))
    vln._lib.check(vln._lib.load().vln_persistent_check(), "vln_persistent_check")
    return out, int(ag.clock.word.item()), ag.clock.host


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("branch", ["steps", "branch", "ride"])
def test_iteration_graph_equals_eager(vln, dtype, branch):
    eager, word_e, host_e = _run(vln, dtype, False, branch)
    graph, word_g, host_g = _run(vln, dtype, True, branch)
    assert word_e == host_e == word_g == host_g == 6 * vln.DeviceClock.STRIDE
    losses = [float(o[0]) for o in eager]
    assert len(set(losses)) == len(losses)                           # different batches, fresh dropout masks every iteration
    for i, (a, b) in enumerate(zip(eager, graph)):
        assert torch.isfinite(a[0]).all()
        for x, y, what in zip(a, b, ("loss", "parameters", "RMSprop state", "gradient norms")):
            assert torch.equal(x, y), f"iteration {i}: {what} differ between the eager and the replayed iteration"


@pytest.mark.parametrize("graph", [True, False])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_chained_decoder_steps_equal_unchained_steps(vln, dtype, graph):
    """vln_envdrop_step.chain (round 4): a step leaves the tanh + dropout epilogue of its last product pending and the NEXT step's
    first launch finishes it; in the backward the act-embedding / h_tilde_prev stage of step t rides in the first launch of step
    t - 1's backward; the rollout-level logits / loss and the weight gradients flush what the last / first step left.  The same
    arithmetic in the same order: losses, parameters, optimizer state and gradient norms over six iterations with fresh batches
    and masks equal the unchained steps' bit for bit -- as eager launches with per-step hipGraphs and as one captured iteration."""
    ref, _, _ = _run(vln, dtype, graph, "ride", chain=False)
    got, _, _ = _run(vln, dtype, graph, "ride", chain=True)
    for i, (a, b) in enumerate(zip(ref, got)):
        for x, y, what in zip(a, b, ("loss", "parameters", "RMSprop state", "gradient norms")):
            assert torch.equal(x, y), f"iteration {i}: {what} differ between chained and unchained decoder steps"


@pytest.mark.parametrize("graph", [True, False])
def test_decoder_gradient_ride_equals_its_own_launches(vln, graph):
    """ops.GradRide / vln_wgrad_ride_post (round 4): the decoder's rollout-level weight and bias gradients (pack, packed contraction,
    column sums) travel as passenger workgroups of the encoder's BPTT launch instead of three launches in front of it.  Same blocks,
    same tiles, same arithmetic: losses, parameters, optimizer state and gradient norms over six iterations equal the stand-alone
    launches' bit for bit, eager and as one captured iteration -- and the rides were really carried (vln_wgrad_ride_stats)."""
    before = vln.ops.GradRide.stats()
    ref, _, _ = _run(vln, torch.bfloat16, graph, "ride", ride=False)
    mid = vln.ops.GradRide.stats()
    assert mid == before                                             # nothing was posted with the switch off
    got, _, _ = _run(vln, torch.bfloat16, graph, "ride", ride=True)
    after = vln.ops.GradRide.stats()
    assert after["carried"] - mid["carried"] >= (3 if graph else 6) and after["issued_alone"] == mid["issued_alone"]
    for i, (a, b) in enumerate(zip(ref, got)):
        for x, y, what in zip(a, b, ("loss", "parameters", "RMSprop state", "gradient norms")):
            assert torch.equal(x, y), f"iteration {i}: {what} differ between the gradient ride and its own launches"


@pytest.mark.parametrize("dtype,read", [("fp32", True), ("bf16", True), ("bf16", "poll"), ("bf16", "handshake"), ("fp32", "handshake")])
def test_il_plus_a2c_iteration_as_graph_segments_equals_eager(vln, dtype, read):
    """BASELINE config 3's per-rank iteration (trainer.py:411-427: IL rollout + sampled A2C rollout + critic, one RMSprop) as
    graphs.SegmentedIterationGraph -- one hipGraph per sampled step with the action read on the host between them, the backward
    of both rollouts + the update in the last segment (trainers.EnvDropA2CIteration) -- against the same pieces issued
    eagerly: sampled actions (the draws follow the device clock), loss, parameters and RMSprop state bit for bit over 5 iterations."""
    dev = torch.device(DEV)
    dt = torch.bfloat16 if dtype == "bf16" else torch.float32
    outs = []
    for use_graph in (False, True):
        torch.manual_seed(91)
        store = vln.synthetic.build_store(dev, dt, n_rows=300, seed=5)
        torch.manual_seed(92)
        # (read = "poll": the replayed segments' host steps spin on the pinned action words instead of synchronising the stream;
        #  read = "handshake": ONE graph, every host turn a vln_host_wait inside it -- graphs.HandshakeIterationGraph)
        tape = vln.synthetic.tape_to(vln.synthetic.make_tape(16, 24, 5, 6, 600, n_rows=store.N), dev, store=store)
        ag = vln.trainers.EnvDropA2CIteration(dev, dt, tape, T_il=3, graph=True, read_actions=read)
        it, capture = ag.iteration, ag.capture
        state = dict(opt=ag.opt, enc=ag.enc, dec=ag.dec, cri=ag.cri, a_host=ag.a_host, clock=ag.clock)
        state["enc"].deterministic_embedding_grad = True
        rec = []

        def snap(loss):
            torch.cuda.synchronize()
            rec.append((loss.detach().clone(), state["opt"].flat_p.clone(), state["opt"].sq.clone(), state["a_host"].clone()))

        for _ in range(2):
            snap(it())
        run = capture() if use_graph else it
        for _ in range(3):
            snap(run())
        outs.append(rec)
        vln._lib.check(vln._lib.load().vln_persistent_check(), "vln_persistent_check")
    acts = [o[3] for o in outs[0]]
    assert any(not torch.equal(acts[0], a) for a in acts[1:])            # fresh draws every iteration
    for i, (a, b) in enumerate(zip(*outs)):
        assert torch.isfinite(a[0]).all()
        for x, y, what in zip(a, b, ("loss", "parameters", "RMSprop state", "sampled actions")):
            assert torch.equal(x, y), f"iteration {i}: {what} differ between the eager and the segment-replayed IL + A2C iteration"


def test_decoder_gradient_ride_at_baseline_size(vln):
    """The same equality at BASELINE config 1's per-GPU size (B 64, L 80, T 7): 128 recurrence workgroups + 64 passengers in the BPTT
    launch (the passengers-only barrier over 64 workgroups on 8 XCDs), 448-row contractions -- as one captured iteration."""
    ref, _, _ = _run(vln, torch.bfloat16, True, "ride", n_eager=2, n_more=3, ride=False, shape=(64, 80, 7, 8))
    mid = vln.ops.GradRide.stats()
    got, _, _ = _run(vln, torch.bfloat16, True, "ride", n_eager=2, n_more=3, ride=True, shape=(64, 80, 7, 8))
    after = vln.ops.GradRide.stats()
    assert after["carried"] > mid["carried"] and after["issued_alone"] == mid["issued_alone"]
    for i, (a, b) in enumerate(zip(ref, got)):
        for x, y, what in zip(a, b, ("loss", "parameters", "RMSprop state", "gradient norms")):
            assert torch.equal(x, y), f"iteration {i}: {what} differ between the gradient ride and its own launches"


def test_rides_fall_back_to_their_own_launches_when_the_recurrence_fills_the_chip(vln):
    """B = 128 with a bidirectional 256-unit encoder: 256 recurrence workgroups = every CU of an MI355X, no room for passengers.  The
    rollout gather, the tail of the pulled batch (`launch_fetch_part`) and the decoder's gradient ride (`ride_issue_alone`) must then
    run as their own launches with the same results: pulled batches + rides against device-resident batches without the gradient ride."""
    ref, _, _ = _run(vln, torch.bfloat16, False, "ride", n_eager=2, n_more=2, source="device", ride=False, shape=(128, 24, 4, 6))
    before = vln.ops.GradRide.stats()
    got, _, _ = _run(vln, torch.bfloat16, False, "ride", n_eager=2, n_more=2, source="pull", ride=True, shape=(128, 24, 4, 6))
    after = vln.ops.GradRide.stats()
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    if cus <= 256:
        assert after["issued_alone"] - before["issued_alone"] == 4 and after["carried"] == before["carried"]
    for i, (a, b) in enumerate(zip(ref, got)):
        for x, y, what in zip(a, b, ("loss", "parameters", "RMSprop state", "gradient norms")):
            assert torch.equal(x, y), f"iteration {i}: {what} differ"


def test_gradient_ride_without_a_carrier_is_issued_by_the_flush(vln):
    """A posted ride that no backward recurrence picks up: vln_wgrad_ride_flush (the autograd engine's end-of-backward callback in
    EnvDropDecoder._deferred_wgrads) issues it as its own launches -- same results as WgradBatch / ColsumBatch run directly."""
    g = torch.Generator().manual_seed(3)
    dy, x = torch.randn(96, 256, generator=g).to(DEV), torch.randn(96, 384, generator=g).to(DEV)
    outs = []
    for ride in (False, True):
        dw, db = torch.zeros(256, 384, device=DEV), torch.zeros(256, device=DEV)
        before = vln.ops.GradRide.stats()
        with (vln.ops.GradRide.collect() if ride else __import__("contextlib").nullcontext()):
            wb = vln.ops.WgradBatch(True); wb.add(dy, x, dw, True); wb.run()
            cb = vln.ops.ColsumBatch(); cb.add(dy, db, None, True); cb.run()
        if ride:
            assert float(dw.abs().max()) == 0.0                       # posted, not launched
            vln.ops.GradRide.flush()
            assert vln.ops.GradRide.stats()["issued_alone"] == before["issued_alone"] + 1
        torch.cuda.synchronize()
        outs.append((dw, db))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert float(outs[0][0].abs().max()) > 0


@pytest.mark.parametrize("source", ["device", "pull"])
@pytest.mark.parametrize("graph", [True, False])
def test_prologue_launch_equals_separate_launches(vln, graph, source):
    """runtime.DeviceClock.prologue / vln_prologue: the batch pull, the device clock's tick and both modules' weight-shadow refresh
    as ONE launch at the top of the iteration (block ranges of one kernel) against the same work as separate launches -- the
    shadows follow every optimizer step, the offsets every tick, the batches the slot ring: bit-identical over six iterations."""
    ref, w0, h0 = _run(vln, torch.bfloat16, graph, "ride", source=source, prologue=False)
    got, w1, h1 = _run(vln, torch.bfloat16, graph, "ride", source=source, prologue=True)
    assert w0 == w1 == h0 == h1
    for i, (a, b) in enumerate(zip(ref, got)):
        for x, y, what in zip(a, b, ("loss", "parameters", "RMSprop state", "gradient norms")):
            assert torch.equal(x, y), f"iteration {i}: {what} differ between the prologue launch and separate launches"


@pytest.mark.parametrize("source", ["pull", "push"])
@pytest.mark.parametrize("graph", [True, False])
def test_batches_pulled_from_pinned_host_memory_equal_copied_batches(vln, graph, source):
    """vln.LiveBatch("pull") / staging.HostBatchFeed / vln_host_fetch: the batches wait in pinned host memory, `load(k)` is one
    host store into a ring of slot words and the iteration's FIRST launch pulls the batch through PCIe -- also as the first node of
    the captured iteration.  22 iterations over 5 different batches (the 16-slot ring wraps, the host runs ahead of the device):
    losses, parameters, optimizer state and gradient norms equal the device-resident batches' bit for bit.
    source "push" (round 5): the same batches SENT ahead -- an asynchronous H2D copy on a copy stream into a device ring slot, one
    select early (HostBatchFeed(prefetch=True).send_ahead), the first launch then moves the batch HBM -> HBM -- and, in that run, the
    decoder's weight shadows refreshed by the gather ride's passengers instead of the prologue launch (RolloutRide.carry_shadows)."""
    ref, _, _ = _run(vln, torch.bfloat16, graph, "ride", n_more=20)
    got, _, _ = _run(vln, torch.bfloat16, graph, "ride", n_more=20, source=source, ride_shadows=source == "push")
    assert len(ref) == len(got) == 22
    for i, (a, b) in enumerate(zip(ref, got)):
        for x, y, what in zip(a, b, ("loss", "parameters", "RMSprop state", "gradient norms")):
            assert torch.equal(x, y), f"iteration {i}: {what} differ between pulled and device-resident batches"


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_segmented_iteration_graph_equals_the_single_graph(vln, dtype):
    """ONE path for N = 1 and N > 1.  The data-parallel form of the iteration -- graph A (forward, loss,
    the decoder's backward), host (start the decoder slice's all-reduce), graph B (the encoder's backward), host (reduce the
    rest, wait), graph C (clip + update); graphs.SegmentedIterationGraph -- replays the same kernels in the same order as the
    single graph: losses, parameters, RMSprop state and gradient norms are equal bit for bit over four replays with fresh
    batches and masks, its eager form too, and the host segments run exactly once per iteration, in order, between the graphs."""
    single, word_s, host_s = _run(vln, dtype, True, "ride")
    calls = []
    seg, word_g, host_g = _run(vln, dtype, True, "ride", segmented=True, calls=calls)
    seg_eager, _, _ = _run(vln, dtype, False, "ride", segmented=True)
    assert word_s == host_s == word_g == host_g
    # 2 eager iterations + 1 capture pass + 4 replays, every one of them: start(decoder group) then finish
    assert calls == [("start", 1), ("finish",)] * 7
    for i, (a, b, c) in enumerate(zip(single, seg, seg_eager)):
        for x, y, z, what in zip(a, b, c, ("loss", "parameters", "RMSprop state", "gradient norms")):
            assert torch.equal(x, y), f"iteration {i}: {what} differ between the single graph and the three segments"
            assert torch.equal(x, z), f"iteration {i}: {what} differ between the single graph and the eager segments"


@pytest.mark.parametrize("graph", [True, False])
def test_launch_sequence_wrap_is_transparent(vln, graph):
    """DeviceClock's wrap guard: the recurrence's granule tags carry a 24-bit launch sequence that the device clock bumps by 64
    per iteration -- it wraps after 262,144 iterations.  Forced to the top of its range, the iterations before, across and after
    the wrap (exchange cleared, sequence restarted, stream-ordered between replays) give the same bits as a run far from it,
    and no wait times out."""
    far, word_f, host_f = _run(vln, torch.bfloat16, graph, "ride", n_more=6)
    near, word_n, host_n = _run(vln, torch.bfloat16, graph, "ride", n_more=6, near_wrap=True)
    assert word_f == word_n == host_f == host_n
    for i, (a, b) in enumerate(zip(far, near)):
        for x, y, what in zip(a, b, ("loss", "parameters", "RMSprop state", "gradient norms")):
            assert torch.equal(x, y), f"iteration {i}: {what} differ across the sequence wrap"


def test_clock_offsets_are_the_host_counter_offsets(vln):
    """A module driven by the clock uses exactly the Philox offsets a host counter standing at `clock.host` would: the
    encoder's context with the clock at k ticks equals the context of a clock-less encoder whose call counter is k * STRIDE."""
    dev = torch.device(DEV)
    torch.manual_seed(3)
    enc = vln.EncoderLSTM(50, 32, 64, 0, 0.5, True, 1).to(dev).train()
    tokens = torch.randint(4, 50, (8, 12), device=dev)
    lens = torch.tensor([12, 12, 11, 9, 7, 5, 3, 2])
    clock = vln.DeviceClock(dev).attach(enc)
    clock.tick(); clock.tick()
    a = enc(tokens, lens)[0]
    assert enc._calls == 2 * clock.STRIDE + 1
    del enc.clock
    enc._calls = 2 * clock.STRIDE
    b = enc(tokens, lens)[0]
    assert torch.equal(a, b)


def _small_agent(vln, kind, dev, seed):
    """A miniature Self-Monitor / Speaker-Follower training iteration (scripts/bench_agents.py in small): encoder, T decoder
    steps with their losses, backward, fused Adam -- with a DeviceClock attached to everything that owns dropout sites."""
    torch.manual_seed(seed)
    B, L, T, C, F = 16, 24, 3, 6, 192
    g = torch.Generator().manual_seed(seed + 1)
    if kind == "monitor":
        enc = vln.EncoderLSTM(60, 32, 64, 0, 0.5, False, 1).to(dev).train()
        dec = vln.MonitorDecoder(64, 0.5, L, (32, 128), F, F).to(dev).train()
        dec.merge_projections = True              # the BN-MLP's two calls per step as one two-batch call
        opts = [vln.optim.FusedAdam([list(enc.parameters()) + list(dec.parameters())], lr=1e-3)]
    else:
        enc = vln.EncoderLSTM(60, 32, 64, 0, 0.5, True, 2).to(dev).train()
        dec = vln.AttnDecoderLSTM(64, 0.5, F, F).to(dev).train()
        opts = [vln.optim.FusedAdam([list(enc.parameters())], lr=1e-3), vln.optim.FusedAdam([list(dec.parameters())], lr=1e-3)]
    enc.deterministic_embedding_grad = True
    clock = vln.DeviceClock(dev).attach(enc, dec)
    for o in opts:
        o.use_clock(clock)
    batches = []
    for k in range(4):
        tokens = torch.randint(4, 60, (B, L), generator=g)
        lens = torch.sort(torch.randint(4, L + 1, (B,), generator=g), descending=True).values; lens[0] = L
        for i, n in enumerate(lens.tolist()):
            tokens[i, n:] = 0
        steps = []
        for t in range(T):
            ncand = torch.randint(2, C + 1, (B,), generator=g)
            cmask = torch.arange(C)[None, :] >= ncand[:, None]
            steps.append(dict(cand=(torch.randn(B, C, F, generator=g).abs() * 0.5 * (~cmask)[..., None]).to(dev), cmask=cmask.to(dev),
                              img=(torch.randn(B, 36, F, generator=g).abs() * 0.5).to(dev),
                              target=(torch.rand(B, generator=g) * ncand.float()).long().to(dev),
                              start=(torch.rand(B, generator=g) * 15 + 4).to(dev), cur=(torch.rand(B, generator=g) * 10 + 0.2).to(dev),
                              ended=(torch.rand(B, generator=g) < 0.1 * t).to(dev)))
        batches.append(dict(tokens=tokens.to(dev), lens=lens.to(dev, torch.int32), steps=steps))
    live = {k: (v.clone() if torch.is_tensor(v) else [{kk: vv.clone() for kk, vv in s.items()} for s in v]) for k, v in batches[0].items()}

    def load(k):                       # the batch into the fixed buffers the (captured) iteration reads
        b = batches[k % len(batches)]
        live["tokens"].copy_(b["tokens"]); live["lens"].copy_(b["lens"])
        for ls, bs in zip(live["steps"], b["steps"]):
            for kk in ls:
                ls[kk].copy_(bs[kk])

    def it():
        clock.tick()
        for o in opts:
            o.zero_grad()
        ctx, h, c = enc(live["tokens"], live["lens"])
        seq_mask = live["tokens"] == 0
        a_prev = torch.zeros(B, F, device=dev)
        loss = 0.0
        for t, s in enumerate(live["steps"]):
            if kind == "monitor":
                (logit, prog), (h, c), _ = dec(None, a_prev, s["cand"], h, c, ctx, seq_mask, s["cmask"])
                lt, _ = vln.losses.monitor_mixed_loss(logit, s["target"], s["cmask"], prog, s["start"], s["cur"], s["ended"], t, 0.5)
            else:
                logit, (h, c), _ = dec(s["img"], a_prev, s["cand"], h, c, ctx, seq_mask)
                lt = vln.losses.masked_cross_entropy(logit, s["target"], s["cmask"], "mean")
            loss = loss + lt
            a_prev = s["cand"][torch.arange(B, device=dev), s["target"]].detach()
        loss.backward()
        for o in opts:
            o.step()
        return loss

    return it, load, opts, clock, dec


@pytest.mark.parametrize("rollout_wgrads", [False, True], ids=["per_step_wgrads", "rollout_wgrads"])
@pytest.mark.parametrize("kind", ["monitor", "follower"])
def test_other_agents_iteration_graph_equals_eager(vln, kind, rollout_wgrads):
    """The Self-Monitor and Speaker-Follower iterations (one C call per decoder step each way, BN-MLP, fused step loss, fused
    Adam whose step count lives in a device word) captured whole and replayed: equal to the eager iterations bit for bit --
    also with the parameter gradients formed once per rollout from autograd's end-of-backward callback
    (functional.RolloutWgrads: the callback's launches are captured like any other)."""
    dev = torch.device(DEV)
    runs = []
    F_ = vln.functional
    try:
        F_.set_grad_in_place(rollout_wgrads)
        F_.set_rollout_wgrads(rollout_wgrads)
        for graph in (False, True):
            F_.ROLLOUT_WGRADS.stats[:] = [0, 0]
            it, load, opts, clock, dec = _small_agent(vln, kind, dev, 31)
            out = []
            for k in range(2):
                load(k)
                out.append((it().detach().clone(), [o.flat_p.clone() for o in opts]))
            torch.cuda.synchronize()
            g = vln.IterationGraph(it, clock).capture() if graph else None
            for k in range(2, 6):
                load(k)
                loss = g.replay() if graph else it()
                torch.cuda.synchronize()
                out.append((loss.detach().clone(), [o.flat_p.clone() for o in opts]))
            runs.append(out)
            assert (F_.ROLLOUT_WGRADS.stats[0] > 0) == rollout_wgrads
    finally:
        F_.set_rollout_wgrads(False)
        F_.set_grad_in_place(False)
    for i, (a, b) in enumerate(zip(*runs)):
        assert torch.isfinite(a[0]).all()
        assert torch.equal(a[0], b[0]), f"iteration {i}: loss {float(a[0])} vs {float(b[0])}"
        for x, y in zip(a[1], b[1]):
            assert torch.equal(x, y), f"iteration {i}: parameters differ"


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_speaker_iteration_as_one_graph_equals_eager(vln, dtype):
    """Round 6: the speaker's teacher-forcing iteration (three sequence LSTMs, two attentions, vocabulary CE, clip, two Adam) on
    device-clock dropout offsets and launch sequences, captured whole (trainers.SpeakerIteration.capture) and replayed over
    changing batches: losses and parameters equal the eager iterations on the same clock bit for bit."""
    dev = torch.device(DEV)
    B, Lp, Lw, V, F, vocab = 8, 4, 12, 36, 96, 50

    def batch(k):
        g = torch.Generator().manual_seed(100 + k)
        lengths = torch.randint(2, Lp + 1, (B,), generator=g); lengths[0] = Lp
        wl = torch.randint(4, Lw + 1, (B,), generator=g); wl[0] = Lw
        insts = torch.zeros(B, Lw, dtype=torch.long)
        for b in range(B):
            n = int(wl[b])
            insts[b, 0] = 3; insts[b, 1:n - 1] = torch.randint(4, vocab, (n - 2,), generator=g); insts[b, n - 1] = 2
        return dict(can=(torch.randn(B, Lp, F, generator=g).abs() * 0.5).to(dev), img=(torch.randn(B, Lp, V, F, generator=g).abs() * 0.5).to(dev),
                    lengths=lengths, insts=insts.to(dev))

    runs = []
    for graph in (False, True):
        torch.manual_seed(23)
        it = vln.trainers.SpeakerIteration(dev, dtype, vocab=vocab, wemb=32, rnn=64, feature_size=F, angle_size=32, lr=1e-3, graph=True)
        it.dec.deterministic_embedding_grad = True
        out = []
        it.load(batch(0))
        if graph:
            it.capture(warmup=3)
        else:
            for _ in range(3):
                it.iteration()
        for k in range(1, 5):
            it.load(batch(k))
            loss = it.replay() if graph else it.iteration()
            torch.cuda.synchronize()
            out.append((loss.detach().clone(), it.opt_e.flat_p.clone(), it.opt_d.flat_p.clone()))
        runs.append(out)
    for i, (a, b) in enumerate(zip(*runs)):
        assert torch.isfinite(a[0]).all()
        assert torch.equal(a[0], b[0]), f"iteration {i}: loss {float(a[0])} vs {float(b[0])}"
        assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]), f"iteration {i}: parameters differ"
    assert not torch.equal(runs[0][0][0], runs[0][1][0])          # (the batches do change)


@pytest.mark.parametrize("graph", ["on", "off"])
def test_bench_falls_back_to_per_step_launches_after_a_timeout(graph):
    """the sticky timeout raises VlnError from the next library entry, which bench.py did not catch -- the
    fallback branch was dead.  `--inject-timeout K` raises the sticky word as a timed-out wait would: the run must catch it,
    switch to per-step launches, warm up again (re-recording the iteration graph) and still print its JSON line."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "2", "--no-secondary", "--no-roofline",
                          "--no-cpu-baseline", "--viewpoints", "400", "--iteration-graph", graph, "--inject-timeout", "2"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "timed out" in out.stderr and "using per-step launches" in out.stderr
    rep = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rep["ms_per_step"] > 0 and rep["steps"] == 3


def test_replay_reports_what_an_earlier_replay_raised_on_the_device(vln):
    """A replayed iteration has no host code between its launches.  What a launch of an EARLIER replay raised on the device -- a
    timed-out bounded wait, an out-of-range gather index: host-mapped sticky words -- must surface as VlnError at the next
    replay, once, and the graph keeps working afterwards."""
    dev = torch.device(DEV)
    lib = vln._lib.load()
    store = vln.synthetic.build_store(dev, torch.bfloat16, n_rows=300, seed=5)
    tapes = [vln.synthetic.tape_to(vln.synthetic.make_tape(16, 24, 4, 6, seed=900 + k, n_rows=store.N), dev, store=store) for k in range(3)]
    live = vln.LiveBatch(tapes)
    ag = vln.trainers.EnvDropILIteration(dev, torch.bfloat16, 1, arena=True)
    ag.clear_grads_in_step = True
    ag.ride_gather = True
    ag.use_clock(store)
    for k in range(2):
        ag.iteration(live.load(k))
    ag.capture(live.live)
    live.load(2); ag.replay()
    torch.cuda.synchronize()
    # (a) a REAL bad index inside a replay: the passengers' gather zeroes the row and raises the word
    live.live["steps"][1]["rows"][3] = store.N + 7
    ag.replay()
    torch.cuda.synchronize()
    with pytest.raises(vln.VlnError, match="out of range"):
        ag.replay()
    live.load(0)                                     # a good batch again: the graph is intact
    loss = ag.replay()
    torch.cuda.synchronize()
    assert torch.isfinite(loss).all()
    # (b) a timeout, injected through the test hook
    vln._lib.check(lib.vln_debug_raise_sticky(0), "vln_debug_raise_sticky")
    with pytest.raises(vln.VlnError, match="timed out"):
        ag.replay()
    lib.vln_set_persistent(1)                        # (the report switched this process to per-step launches)
    vln._lib.check(lib.vln_persistent_check(), "clean again")


def test_bench_survives_a_failed_graph_capture():
    """If stream capture of the iteration fails on some box or runtime, the bench line is still owed: eager launches (per-step
    graphs, the same kernels) run instead and the JSON says so."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "2", "--no-secondary", "--no-roofline",
                          "--no-cpu-baseline", "--viewpoints", "400", "--inject-capture-failure"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "could not be captured" in out.stderr
    rep = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rep["config"]["iteration_graph"] is False and rep["ms_per_step"] > 0


def test_host_batch_feed_refuses_a_fetch_without_a_select_and_two_selects_in_a_row(vln):
    """the device picks its slot by a count of the fetches that ran, the host writes the slot of its count of
    selects; nothing else ties the two.  The protocol is checked on the host: a fetch issued with no select before it (the eager
    warm-up of a capture, a retry after an exception) and a second select before the first one's fetch was issued RAISE instead of
    pulling stale or empty slots from then on; `resync()` realigns the counts and the pulls are right again."""
    dev = torch.device(DEV)
    live = torch.zeros(64, dtype=torch.uint8, device=dev)
    feed = vln.HostBatchFeed(live, ring=4)
    blobs = [feed.register(torch.full((64,), k + 1, dtype=torch.uint8)) for k in range(6)]
    with pytest.raises(vln._lib.VlnError, match="without a select"):
        feed.fetch()
    feed.select(blobs[0]); feed.fetch(); feed.launched()
    torch.cuda.synchronize()
    assert int(live[0]) == 1
    feed.select(blobs[1])
    with pytest.raises(vln._lib.VlnError, match="has not been issued"):
        feed.select(blobs[2])
    feed.resync()                                   # the select of blob 1 never ran: forgotten, device and host counts equal again
    for k in (3, 4, 5, 2, 1):                       # (more than the ring holds)
        feed.select(blobs[k]); feed.fetch(); feed.launched()
        torch.cuda.synchronize()
        assert int(live[0]) == k + 1 and int(live[63]) == k + 1
    with pytest.raises(vln._lib.VlnError, match="without a select"):
        feed.fetch()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_chained_steps_survive_other_work_on_the_shared_workspace(vln, dtype):
    """a chained decoder step leaves split-K slabs PENDING between two step calls.  They live in a workspace of
    the decoder's own now: products, weight-gradient batches or rides issued on the shared per-stream workspace between two steps
    (here: a product big enough to overwrite -- and to REALLOCATE -- `ops.workspace`) leave the rollout bit-identical."""
    dev = torch.device(DEV)
    B, L, V, C, H, IMG, ANG, AE, T = 16, 24, 36, 6, 64, 96, 32, 16, 4
    F = IMG + ANG

    def rollout(disturb):
        torch.manual_seed(5)
        dec = vln.EnvDropDecoder(H, 0.5, 0.3, AE, ANG, F, compute_dtype=dtype).to(dev).train()
        dec.defer_logits = dec.chain_steps = True
        g = torch.Generator().manual_seed(6)
        ctx = (torch.randn(B, L, H, generator=g) * 0.5).to(dev).requires_grad_(True)
        ht = torch.tanh(torch.randn(B, H, generator=g)).to(dev).requires_grad_(True); c = (torch.randn(B, H, generator=g) * 0.5).to(dev)
        mask = (torch.arange(L)[None, :] >= torch.randint(4, L + 1, (B, 1), generator=g)).to(dev)
        ce = vln.losses.RolloutCE()
        h = ht
        big = torch.randn(4096, 1024, device=dev); w = torch.randn(1024, 1024, device=dev)
        for t in range(T):
            img = (torch.randn(B, V, F, generator=g).abs() * 0.5).to(dev); cand = (torch.randn(B, C, F, generator=g).abs() * 0.5).to(dev)
            logit, (h, c), ht = dec(torch.sin(torch.randn(B, ANG, generator=g)).to(dev), img, cand, ht, h, c, ctx, mask)
            ce.add(logit, torch.randint(0, C, (B,), generator=g).to(dev), torch.zeros(B, C, dtype=torch.bool, device=dev))
            if disturb:                # scribbles over (and, the first time, reallocates) the shared workspace between two steps
                vln.ops.linear_fwd_slabs(big, w, ws_floats=(1 << 22) + (t + 1) * (1 << 20))
        loss = ce.sum(scale=0.1)
        if disturb:
            hook = ctx.register_hook(lambda g_: (vln.ops.linear_fwd_slabs(big, w), g_)[1])
        loss.backward()
        torch.cuda.synchronize()
        return [loss.detach().clone(), ctx.grad.clone()] + [p.grad.clone() for p in dec.parameters()]

    ref, got = rollout(False), rollout(True)
    for i, (a, b) in enumerate(zip(ref, got)):
        assert torch.equal(a, b), f"tensor {i} differs after other work used the shared workspace between chained steps"


# ---- the host inside ONE captured iteration (graphs.HandshakeIterationGraph; VERDICT r5 weak 4 / ADVICE r5) ---------------------------
def _handshake_toy(vln, spin_limit, fail_turn=None, never_answer=False):
    """A miniature iteration with two host turns: x <- x + 1 (graph) | host | x <- 2 x (graph) | host | x <- x + 3 (graph)."""
    dev = torch.device(DEV)
    clock = vln.DeviceClock(dev)
    x = torch.zeros(64, device=dev)
    turns = []

    def host(i):
        def run():
            if fail_turn == i:
                raise RuntimeError(f"the simulator failed in host turn {i}")
            turns.append(i)
        return run

    def first():
        clock.tick()
        x.add_(1.0)
    segs = [("graph", first), ("host", host(0)), ("graph", lambda: x.mul_(2.0)), ("host", host(1)), ("graph", lambda: x.add_(3.0))]
    hg = vln.HandshakeIterationGraph(segs, clock, spin_limit=spin_limit).capture()
    if never_answer:
        hg.host_fns = []                      # the host's part of replay() never runs: no flag is ever written
    return hg, x, turns


def test_handshake_graph_plays_the_host_turns_in_order(vln):
    hg, x, turns = _handshake_toy(vln, spin_limit=0)
    for k in range(3):
        x.zero_()
        hg.replay()
        torch.cuda.synchronize()
        assert turns == [0, 1] * (k + 1) and float(x[0]) == 5.0
    assert vln._lib.load().vln_persistent_check() == 0


def test_handshake_graph_host_running_ahead_of_the_device(vln):
    """Host turns that wait for nothing: the host finishes its part of replay k long before the device has reached that replay's waits and
    starts replay k + 1.  The per-turn acknowledgement words keep it from rewriting turn i's flag before the device has passed turn i of
    the previous replay (without them the device would see the NEXT iteration's value, never its own, and time out)."""
    hg, x, turns = _handshake_toy(vln, spin_limit=0)
    n = 20                                   # (the value stays below 2^24: exact in fp32)
    for _ in range(n):                       # no synchronisation between replays
        hg.replay()
    torch.cuda.synchronize()
    # x <- ((x + 1) * 2) + 3 per replay, from 0
    want = 0.0
    for _ in range(n):
        want = (want + 1.0) * 2.0 + 3.0
        want = float(torch.tensor(want, dtype=torch.float32))
    assert float(x[0]) == want and turns == [0, 1] * n
    assert vln._lib.load().vln_persistent_check() == 0


def test_handshake_graph_host_never_answers(vln):
    """A host that never writes its flag: the in-graph wait is BOUNDED (here 20 ms of wall clock: spin_limit < 0 = microseconds on the
    100 MHz constant clock; the default is 2 s), raises the sticky word and lets the queue drain; the NEXT replay reports the
    iteration as invalid instead of running on (VlnError from vln_persistent_check)."""
    import time
    lib = vln._lib.load()
    hg, x, _ = _handshake_toy(vln, spin_limit=-20000, never_answer=True)
    t0 = time.perf_counter()
    hg.replay()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert 0.03 < dt < 1.0, f"two 20 ms waits were expected to time out, the replay took {dt:.3f} s"
    with pytest.raises(vln.VlnError, match="wait\\(s\\) for the host timed out"):
        hg.replay()
    assert lib.vln_persistent_check() == 0                       # reported once
    # a poll-count bound (spin_limit > 0) ends the same way
    hg2, _, _ = _handshake_toy(vln, spin_limit=2000, never_answer=True)
    hg2.replay()
    torch.cuda.synchronize()
    assert lib.vln_persistent_check() != 0 and lib.vln_persistent_check() == 0


def test_handshake_graph_host_turn_raises(vln):
    """an exception in a host turn must not leave the device spinning in the remaining waits until their bound.
    replay() releases every flag not yet written with the POISON value: each wait ends at once (well under the 2 s default bound),
    the exception propagates, the sticky word marks the iteration invalid for the next library entry, and the graph stays usable."""
    import time
    lib = vln._lib.load()
    hg, x, turns = _handshake_toy(vln, spin_limit=0, fail_turn=0)
    t0 = time.perf_counter()
    with pytest.raises(RuntimeError, match="simulator failed"):
        hg.replay()
    torch.cuda.synchronize()
    assert time.perf_counter() - t0 < 0.5                        # drained at once, not after 2 x 2 s
    assert hg.poisoned == 1 and turns == []
    with pytest.raises(vln.VlnError, match="host"):
        hg.replay()                                              # the abandoned iteration is reported before anything new runs
    assert lib.vln_persistent_check() == 0
    # the same graph, a healthy host: works again
    hg2, x2, turns2 = _handshake_toy(vln, spin_limit=0)
    hg2.replay(); torch.cuda.synchronize()
    assert turns2 == [0, 1] and float(x2[0]) == 5.0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_host_in_the_loop_iteration_as_one_graph_equals_eager(vln, dtype):
    """trainers.EnvDropHostLoopIteration (the reference's loop shape, envdrop.py:151-220: per step the observation arrives from the
    host and the action returns to it): ONE hipGraph whose per-step waits also pull the step's index vectors out of pinned memory
    (vln_host_wait_fetch) and whose actions land in pinned words the host polls, against the eager form (pinned H2D copy + D2H +
    stream synchronise per step): loss, parameters and RMSprop state bit for bit over 5 iterations on rotating batches, every action
    seen by the fake environment."""
    dev = torch.device(DEV)
    outs = []
    for handshake in (False, True):
        torch.manual_seed(77)
        store = vln.synthetic.build_store(dev, dtype, n_rows=300, seed=5)
        tapes = [vln.synthetic.tape_to(vln.synthetic.make_tape(16, 24, 4, 6, seed=500 + k, n_rows=store.N), dev, store=store) for k in range(3)]
        ls = vln.LiveSteps(tapes, dev)
        torch.manual_seed(78)
        it = vln.trainers.EnvDropHostLoopIteration(dev, dtype, ls, store)
        it.enc.deterministic_embedding_grad = True
        it.clock = vln.DeviceClock(dev).attach(it.enc, it.dec)
        it.clock.attach(store)
        rec = []

        def snap(loss):
            torch.cuda.synchronize()
            rec.append((loss.detach().clone(), it.opt.flat_p.clone(), it.opt.sq.clone()))
        for k in range(2):
            snap(it.iteration(k))
        if handshake:
            it.capture(warmup=0)
        for k in range(2, 5):
            snap(it.replay(k) if handshake else it.iteration(k))
        assert it.mismatches == 0                                # the environment saw exactly the teacher's actions, in order
        vln._lib.check(vln._lib.load().vln_persistent_check(), "vln_persistent_check")
        outs.append(rec)
    for i, (a, b) in enumerate(zip(*outs)):
        assert torch.isfinite(a[0]).all()
        for x, y, what in zip(a, b, ("loss", "parameters", "RMSprop state")):
            assert torch.equal(x, y), f"iteration {i}: {what} differ between the eager and the one-graph host-in-the-loop iteration"


def test_live_batch_put_repacks_a_new_batch_into_a_pinned_slot(vln):
    """batches.LiveBatch.put: a trainer's new batch packed into blob slot k and pulled by the captured iteration's first launch gives the
    iteration the same inputs as a LiveBatch built from that batch (loss of the replay bit for bit)."""
    dev = torch.device(DEV)
    dtype = torch.bfloat16
    torch.manual_seed(77)
    store = vln.synthetic.build_store(dev, dtype, n_rows=300, seed=5)
    cpu = [vln.synthetic.make_tape(16, 24, 4, 6, seed=800 + k, n_rows=store.N) for k in range(3)]
    tapes = [vln.synthetic.tape_to(t, dev, store=store) for t in cpu]
    outs = []
    for via_put in (False, True):
        live = vln.LiveBatch(tapes if not via_put else [tapes[0], tapes[0], tapes[0]], source="pull")
        torch.manual_seed(78)
        ag = vln.trainers.EnvDropILIteration(dev, dtype, 1, arena=True)
        ag.use_live(live); ag.ride_gather = True; ag.clear_grads_in_step = True
        ag.enc.deterministic_embedding_grad = True
        ag.use_clock(store)
        for k in range(2):
            ag.iteration(live.load(0))
        ag.capture(live.live)
        rec = []
        for k in (1, 2, 1):
            if via_put:
                live.put(k, tapes[k])                         # (device tensors: put() brings them to the host itself)
            live.load(k)
            loss = ag.replay()
            torch.cuda.synchronize()
            rec.append(loss.detach().clone())
        outs.append(rec)
    for a, b in zip(*outs):
        assert torch.equal(a, b)
