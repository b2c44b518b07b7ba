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


def _run(vln, dtype, graph, branch, n_eager=2, n_more=4):
    import bench
    dev = torch.device(DEV)
    torch.manual_seed(77)
    store = bench.build_store(vln, dev, dtype, n_rows=300, seed=5)
    tapes = [bench.tape_to(bench.make_tape(16, 24, 4, 6, seed=500 + k, n_rows=store.N), dev, store=store) for k in range(5)]
    live = bench.LiveBatch(tapes)
    torch.manual_seed(78)
    ag = bench.GpuAgent(vln, dev, dtype, 1, arena=True)
    ag.clear_grads_in_step = True
    ag.enc.deterministic_embedding_grad = True           # float atomics would differ between two runs of the SAME path
    ag.rollout_gather = ag.gather_branch = branch
    ag.use_clock(store)
    out = []

    def record(loss):
        torch.cuda.synchronize()
        out.append((loss.detach().clone(), ag.opt.flat_p.clone(), ag.opt.sq.clone(), ag.opt.norms.clone()))

    for k in range(n_eager):
        record(ag.iteration(live.load(k)))
    if graph:
        ag.capture(live.live)
        assert ag.clock.host == n_eager * ag.clock.STRIDE           # the captured tick was not counted
    for k in range(n_eager, n_eager + n_more):
        live.load(k)
        record(ag.replay() if graph else ag.iteration(live.live))
    vln._lib.check(vln._lib.load().vln_persistent_check(), "vln_persistent_check")
    return out, int(ag.clock.word.item()), ag.clock.host


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("branch", [False, True])
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
