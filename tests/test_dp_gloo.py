"""CPU, world_size 2 over gloo: the data-parallel path (stride-sharded episodes + ONE flat-bucket all-reduce +
global-batch normalisation) must reproduce the single-process big-batch gradient.  The replica math is the CPU
oracle (the HIP modules need a GPU); GradBucket / stride_shard are the shipped code."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_world(target, extra=(), world=2, attempts=2):
    """Spawn `world` ranks of `target(rank, world, port, queue, *extra)` and collect one result per rank.  The rendezvous port is
    picked by binding port 0 and closing it again, so another process can take it in between (or the previous test's listener
    may still be shutting down): a world that fails to come up is started once more on a fresh port before the test fails."""
    last = None
    for attempt in range(attempts):
        port = _free_port()
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        procs = [ctx.Process(target=target, args=(r, world, port, q) + tuple(extra)) for r in range(world)]
        for p in procs:
            p.start()
        got = []
        try:
            got = [q.get(timeout=150) for _ in range(world)]
        except Exception as e:             # queue.Empty: a rank died or hung before reporting
            last = e
        for p in procs:
            p.join(30)
            if p.is_alive():
                p.terminate()
                p.join(10)
        if len(got) == world and all(p.exitcode == 0 for p in procs):
            return got
        last = last or RuntimeError(f"rank exit codes {[p.exitcode for p in procs]}")
    raise AssertionError(f"the {world}-rank gloo world failed {attempts} times: {last!r}")


def _problem():
    """A tiny EnvDrop IL batch on the oracle: returns (params, loss_fn(rows) -> summed CE over those rows)."""
    sys.path.insert(0, ROOT)
    from oracle import torch_port as O
    g = torch.Generator().manual_seed(11)
    B, L, V, C, H, IMG, ANG, AE = 6, 7, 4, 4, 16, 24, 8, 8
    F = IMG + ANG
    P = {"act_embed.0.weight": torch.randn(AE, ANG, generator=g) * 0.3, "act_embed.0.bias": torch.zeros(AE),
         "lstm.weight_ih": torch.randn(4 * H, AE + F, generator=g) * 0.1, "lstm.weight_hh": torch.randn(4 * H, H, generator=g) * 0.1,
         "lstm.bias_ih": torch.zeros(4 * H), "lstm.bias_hh": torch.zeros(4 * H),
         "text_attn.linear_in.weight": torch.randn(H, H, generator=g) * 0.2,
         "text_attn.linear_out.weight": torch.randn(H, 2 * H, generator=g) * 0.2,
         "visual_attn.linear_in.weight": torch.randn(F, H, generator=g) * 0.2,
         "cand_attn.weight": torch.randn(F, H, generator=g) * 0.2}
    P = {k: v.double() for k, v in P.items()}
    data = dict(a=torch.randn(B, ANG, generator=g).double(), img=torch.randn(B, V, F, generator=g).double().abs(),
                cand=torch.randn(B, C, F, generator=g).double().abs(), h=torch.randn(B, H, generator=g).double(),
                c=torch.randn(B, H, generator=g).double(), ctx=torch.randn(B, L, H, generator=g).double(),
                tgt=torch.tensor([0, 1, 2, 3, 1, -1]))

    def loss_rows(Pm, rows):
        r = torch.tensor(rows)
        lo, *_ = O.envdrop_step(Pm, data["a"][r], data["img"][r], data["cand"][r], data["h"][r], data["c"][r],
                                data["ctx"][r], None)
        return O.masked_cross_entropy(lo, data["tgt"][r], None, "sum")

    return P, loss_rows, B


def _worker(rank, world, port, q, overlap=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import vln_amd as vln
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        P, loss_rows, B = _problem()
        params = [torch.nn.Parameter(v.clone()) for v in P.values()]
        Pm = dict(zip(P.keys(), params))
        bucket = vln.dp.GradBucket(params)
        bucket.zero()
        rows = vln.dp.stride_shard(B, rank, world)
        loss = loss_rows(Pm, rows) * 0.2 / B                 # ML_WEIGHT / GLOBAL batch (envdrop.py:268)
        loss.backward()
        assert all(p.grad is v for p, v in zip(bucket.params, bucket.views)), "autograd must accumulate INTO the bucket views"
        if overlap:      # a finished slice goes out early (what the decoder's grads_ready_hook does), the rest at the end
            bucket.start_allreduce(params[4:])
            assert len(bucket.reducer.pending) == 1
        bucket.allreduce()
        assert not bucket.reducer.pending
        total = vln.dp.allreduce_scalar(torch.tensor([float(len(rows))]))
        # SELF-PACE bookkeeping: every replica ends up with every episode's (index, loss)
        gi, gl = vln.dp.gather_item_losses(torch.tensor(rows), torch.tensor([10.0 * r for r in rows], dtype=torch.float64))
        assert sorted(gi.tolist()) == list(range(B)) and torch.equal(gl, gi.double() * 10.0)
        q.put((rank, bucket.flat.clone(), float(total), rows))
    finally:
        dist.destroy_process_group()


def _worker_fused(rank, world, port, q):
    """The same exchange through the object bench.py uses: optim.FusedRMSprop's flat gradient buffer, the decoder-style early
    slice (`start_allreduce(group)`) and the closing `allreduce()`.  (The update itself is a HIP kernel: not run here.)"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import vln_amd as vln
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        P, loss_rows, B = _problem()
        params = [torch.nn.Parameter(v.clone().float()) for v in P.values()]
        opt = vln.optim.FusedRMSprop([params[:4], params[4:]], lr=1e-4, clip_norm=[40.0, 0.0])
        assert opt.clip_norm == [40.0, 0.0]
        Pm = {k: p for k, p in zip(P.keys(), params)}
        opt.zero_grad()
        rows = vln.dp.stride_shard(B, rank, world)
        data_loss = loss_rows({k: v.double() for k, v in Pm.items()}, rows) * 0.2 / B
        data_loss.backward()
        assert all(p.grad is v for p, v in zip(opt.params, opt.views)), "autograd must accumulate INTO the flat gradient views"
        opt.start_allreduce(1)                               # the second clip group's slice goes out early, asynchronously
        assert len(opt._reducer.pending) == 1
        opt.allreduce()                                      # the rest + wait
        assert not opt._reducer.pending
        try:
            opt.step()
            stepped = True
        except vln.VlnError:
            stepped = False                                  # CPU parameters: the update kernel refuses, no CPU fallback
        q.put((rank, torch.cat([p.grad.reshape(-1) for p in params]).clone(), stepped))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(420)
def test_two_rank_fused_optimizer_bucket_equals_big_batch():
    got = _run_world(_worker_fused)
    P, loss_rows, B = _problem()
    params = [v.clone().requires_grad_(True) for v in P.values()]
    (loss_rows(dict(zip(P.keys(), params)), list(range(B))) * 0.2 / B).backward()
    ref = torch.cat([p.grad.reshape(-1) for p in params])
    for rank, flat, stepped in got:
        assert not stepped
        assert torch.allclose(flat.double(), ref, rtol=1e-5, atol=1e-7), f"rank {rank}: DP gradient != big-batch gradient"
    assert torch.equal(got[0][1], got[1][1])


def _worker_segments(rank, world, port, q):
    """The SEGMENTED iteration of the data-parallel path (graphs.SegmentedIterationGraph, dp.BackwardCut) in its eager form:
    [forward + the 'decoder' half of the backward] [start_allreduce(decoder group)] [the 'encoder' half] [allreduce] [update]."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import vln_amd as vln
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(3)
        enc = torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.Tanh())
        dec = torch.nn.Linear(8, 3)
        X, Y = torch.randn(10, 6), torch.randn(10, 3)
        opt = vln.optim.FusedRMSprop([list(enc.parameters()), list(dec.parameters())], lr=1e-3, clip_norm=[40.0, 40.0])
        rows = vln.dp.stride_shard(10, rank, world)
        cut = vln.dp.BackwardCut()
        log = []

        def part_a():
            opt.zero_grad()
            (h,) = cut.at(enc(X[rows]))
            loss = ((dec(h) - Y[rows]) ** 2).sum() / 10
            loss.backward()
            # the decoder's gradients are final, the encoder's untouched: exactly what the early slice relies on
            log.append(("A", float(opt.flat_g[:opt._begins[1]].abs().sum()) == 0.0, float(opt.flat_g[opt._begins[1]:].abs().sum()) > 0.0))
            return loss

        def host1():
            opt.start_allreduce(1); log.append(("start", len(opt._reducer.pending)))

        def part_b():
            cut.resume(); log.append(("B", float(opt.flat_g[:opt._begins[1]].abs().sum()) > 0.0))

        def host2():
            opt.allreduce(); log.append(("finish", len(opt._reducer.pending)))

        def part_c():
            log.append(("C",))

        seg = vln.SegmentedIterationGraph([("graph", part_a), ("host", host1), ("graph", part_b), ("host", host2), ("graph", part_c)], None)
        seg.run_eager()
        q.put((rank, opt.flat_g.clone(), log))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(420)
def test_two_rank_segmented_iteration_order_and_gradients():
    """the segment order of the one-path data-parallel iteration over a real two-rank world -- the early
    slice leaves between the two halves of the backward with the decoder's gradients final and the encoder's still zero, the
    closing all-reduce covers the rest, and the reduced bucket equals the big-batch gradient on both ranks."""
    got = _run_world(_worker_segments)
    torch.manual_seed(3)
    enc = torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.Tanh())
    dec = torch.nn.Linear(8, 3)
    X, Y = torch.randn(10, 6), torch.randn(10, 3)
    (((dec(enc(X)) - Y) ** 2).sum() / 10).backward()
    ref = torch.cat([p.grad.reshape(-1) for p in list(enc.parameters()) + list(dec.parameters())])
    for rank, flat, log in got:
        assert log == [("A", True, True), ("start", 1), ("B", True), ("finish", 0), ("C",)], log
        assert torch.allclose(flat[:ref.numel()], ref, rtol=1e-5, atol=1e-7), f"rank {rank}: segmented DP gradient != big-batch gradient"
    assert torch.equal(got[0][1], got[1][1])


def _worker_abandon(rank, world, port, q):
    """Rank 1's iteration "raises" at three different points (before its early slice went out, after it, after the whole
    exchange) while rank 0 runs normally: `abandon_iteration` must issue exactly the collectives rank 0 issues -- the ranks
    neither deadlock nor fall out of step, and the iteration after it exchanges correctly."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import vln_amd as vln
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(1)
        params = [torch.nn.Parameter(torch.randn(n, generator=g)) for n in (7, 12, 5, 9)]
        opt = vln.optim.FusedRMSprop([params[:2], params[2:]], lr=1e-4, clip_norm=[40.0, 0.0])
        sums = []
        for it, fail_at in enumerate(("before_early", "after_early", "after_exchange", None)):
            opt.zero_grad()
            opt.flat_g.fill_(float(rank + 1 + it))
            failed = False
            try:
                if rank == 1 and fail_at == "before_early":
                    raise vln.VlnError("timed out (injected)")
                opt.start_allreduce(1)
                if rank == 1 and fail_at == "after_early":
                    raise vln.VlnError("timed out (injected)")
                opt.allreduce()
                if rank == 1 and fail_at == "after_exchange":
                    raise vln.VlnError("timed out (injected)")
            except vln.VlnError:
                failed = True
                opt.abandon_iteration(early_groups=(1,))
            if not failed:                       # the update would follow: it closes the iteration's book-keeping
                opt._started, opt._exchanged = [], False
            assert not opt._reducer.pending and not opt._started and not opt._exchanged
            sums.append(opt.flat_g.clone())
        q.put((rank, torch.stack(sums)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(420)
def test_two_rank_abandoned_iteration_keeps_the_collectives_matched():
    got = dict(_run_world(_worker_abandon))
    for it in range(4):
        want = float((1 + it) + (2 + it))
        for rank in (0, 1):
            flat = got[rank][it]
            live = flat != 0                      # the pads between clip groups stay zero
            assert torch.allclose(flat[live], torch.full_like(flat[live], want)), (rank, it)


@pytest.mark.timeout(480)
def test_bench_gpus_flag_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it must start two ranks itself (round-1 verdict: the flag was a
    no-op) -- checked without a GPU through --rendezvous-only over gloo: rank 0 reports n_gpus 2."""
    import json
    import subprocess
    for attempt in range(2):            # the launcher picks its rendezvous port by bind-and-close: retry once if it was taken meanwhile
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--rendezvous-only"],
                             capture_output=True, text=True, timeout=200,
                             env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
        if out.returncode == 0:
            break
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    rep = json.loads(line)
    assert rep["n_gpus"] == 2 and rep["config"]["world_size"] == 2 and rep["config"]["backend"] == "gloo"
    # a launcher that started the wrong number of ranks is refused
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--rendezvous-only"], capture_output=True,
                         text=True, timeout=60, env=dict(os.environ, WORLD_SIZE="1", RANK="0"))
    assert bad.returncode != 0 and "WORLD_SIZE=1" in bad.stderr


def test_stride_shard_keeps_sorted_batches_sorted():
    sys.path.insert(0, ROOT)
    import vln_amd as vln
    lens = [80, 71, 66, 52, 40, 33, 21, 9]
    for w in (2, 4, 8):
        shards = [vln.dp.stride_shard(len(lens), r, w) for r in range(w)]
        assert sorted(sum(shards, [])) == list(range(len(lens)))
        for s in shards:
            ls = [lens[i] for i in s]
            assert ls == sorted(ls, reverse=True)


@pytest.mark.timeout(420)
@pytest.mark.parametrize("overlap", [False, True], ids=["one_allreduce", "early_slice_async"])
def test_two_rank_bucket_allreduce_equals_big_batch(overlap):
    got = _run_world(_worker, (overlap,))
    # single-process reference on the full batch
    P, loss_rows, B = _problem()
    params = [v.clone().requires_grad_(True) for v in P.values()]
    Pm = dict(zip(P.keys(), params))
    (loss_rows(Pm, list(range(B))) * 0.2 / B).backward()
    ref = torch.cat([p.grad.reshape(-1) for p in params])
    for rank, flat, total, rows in got:
        assert total == B
        assert torch.allclose(flat, ref, rtol=1e-10, atol=1e-12), f"rank {rank}: DP gradient != big-batch gradient"
    assert torch.equal(got[0][1], got[1][1])                 # replicas agree bit-for-bit after the all-reduce


def _mixed_problem():
    """BASELINE config 3 in miniature: EnvDrop IL + A2C mixed loss (envdrop.py:173-195,235-268) on the oracle.  Returns
    (params, parts(rows) -> (summed CE, summed A2C terms, running (step, episode) pairs) over those episodes)."""
    from oracle import torch_port as O
    P, _, B = _problem()
    g = torch.Generator().manual_seed(12)
    H = 16
    P = dict(P, **{"critic.w": torch.randn(H, generator=g).double() * 0.3})
    g2 = torch.Generator().manual_seed(13)
    ANG, V, C, L, IMG = 8, 4, 4, 7, 24
    F = IMG + ANG
    T = 3
    data = dict(a=torch.randn(B, ANG, generator=g2).double(), img=torch.randn(T, B, V, F, generator=g2).double().abs(),
                cand=torch.randn(T, B, C, F, generator=g2).double().abs(), h=torch.randn(B, H, generator=g2).double(),
                c=torch.randn(B, H, generator=g2).double(), ctx=torch.randn(B, L, H, generator=g2).double(),
                tgt=torch.tensor([[0, 1, 2, 3, 1, -1], [1, 1, 0, 2, 3, 0], [2, -1, 1, 0, 0, 3]]),
                act=torch.tensor([[1, 0, 3, 2, 1, 0], [0, 2, 2, 1, 3, 3], [3, 1, 0, 0, 2, 1]]),
                rew=torch.randn(T, B, generator=g2).sign().double(),
                lens=torch.tensor([3, 2, 3, 1, 2, 3]))

    def parts(Pm, rows):
        r = torch.tensor(rows)
        Pd = {k: v for k, v in Pm.items() if k != "critic.w"}
        ht, c = data["h"][r], data["c"][r]
        ce, lps, ens, vals = 0.0, [], [], []
        for t in range(T):
            lo, (h1, c), ht, _ = O.envdrop_step(Pd, data["a"][r], data["img"][t][r], data["cand"][t][r], ht, c, data["ctx"][r], None)
            ce = ce + O.masked_cross_entropy(lo, data["tgt"][t][r], None, "sum")
            dist_ = torch.distributions.Categorical(logits=lo)
            lps.append(dist_.log_prob(data["act"][t][r])); ens.append(dist_.entropy())
            vals.append(h1 @ Pm["critic.w"])
        masks = [(t < data["lens"][r]) for t in range(T)]
        ended = data["lens"][r] < T
        a2c, total = O.a2c_loss(lps, ens, vals, [data["rew"][t][r] for t in range(T)], masks, vals[-1].detach(), ended, 0.9, "none")
        return ce, a2c, total

    return P, parts, B


def _worker_mixed(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import vln_amd as vln
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        P, parts, B = _mixed_problem()
        params = [torch.nn.Parameter(v.clone()) for v in P.values()]
        Pm = dict(zip(P.keys(), params))
        bucket = vln.dp.GradBucket(params)
        bucket.zero()
        rows = vln.dp.stride_shard(B, rank, world)
        ce, a2c, total = parts(Pm, rows)
        # the 'total' normaliser of the RL loss counts running (step, episode) pairs of the WHOLE batch (envdrop.py:262-264):
        # every rank divides its own sum by the global count, the IL term by the global batch (envdrop.py:268)
        total_g = vln.dp.allreduce_scalar(torch.tensor([float(total)], dtype=torch.float64))
        loss = ce * 0.2 / B + a2c / total_g
        loss.backward()
        bucket.allreduce()
        q.put((rank, bucket.flat.clone(), float(total_g), rows))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(420)
def test_two_rank_il_plus_a2c_equals_big_batch():
    """BASELINE config 3 (EnvDrop IL + RL mixed loss, data-parallel): per-rank sums normalised by the GLOBAL batch and the
    GLOBAL running-pair count + one flat all-reduce == the single-process big-batch gradient (critic included)."""
    got = _run_world(_worker_mixed)
    P, parts, B = _mixed_problem()
    params = [v.clone().requires_grad_(True) for v in P.values()]
    Pm = dict(zip(P.keys(), params))
    ce, a2c, total = parts(Pm, list(range(B)))
    (ce * 0.2 / B + a2c / total).backward()
    ref = torch.cat([p.grad.reshape(-1) for p in params])
    for rank, flat, total_g, rows in got:
        assert total_g == total
        assert torch.allclose(flat, ref, rtol=1e-10, atol=1e-12), f"rank {rank}: DP gradient != big-batch gradient"
    assert torch.equal(got[0][1], got[1][1])


def _worker_self_pace(rank, world, port, q, envdrop_form):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import vln_amd as vln
    from oracle import torch_port as O
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        P, parts, B = _mixed_problem()
        params = [torch.nn.Parameter(v.clone()) for v in P.values()]
        Pm = dict(zip(P.keys(), params))
        bucket = vln.dp.GradBucket(params)
        bucket.zero()
        rows = vln.dp.stride_shard(B, rank, world)
        weight = torch.linspace(0.1, 1.0, 40, dtype=torch.float64)            # the curriculum's per-item weights (curriculum.py:428-448)
        item_idx = torch.tensor([3, 17, 8, 29, 11, 35])                       # dataset indices of the batch's episodes (cur_batch_index)
        per_sample = _per_sample_losses(Pm, rows, envdrop_form)
        w_rows = weight[item_idx[torch.tensor(rows)]]
        loss = vln.dp.self_pace_batch_loss(w_rows, per_sample)                # torch.dot(self.weight[cur_batch_idx], cur_loss), curriculum.py:296
        if not envdrop_form:                                                  # curriculum.py:301: / the batch's weight sum -- GLOBAL under DP
            loss = loss / vln.dp.allreduce_scalar(w_rows.sum().reshape(1))
        loss.backward()
        bucket.allreduce()
        gi, gl = vln.dp.gather_item_losses(item_idx[torch.tensor(rows)], per_sample.detach())   # loss_for_item[cur_batch_idx] = ...
        q.put((rank, bucket.flat.clone(), gi.clone(), gl.clone()))
    finally:
        dist.destroy_process_group()


def _per_sample_losses(Pm, rows, envdrop_form):
    """Per-episode losses of the SELF-PACE curriculum: EnvDrop form = (IL + A2C per episode) * ML_WEIGHT / B-style scaling folded
    in by the agent (reduction="none" criteria, envdrop.py:70,178-179); other agents = the per-episode CE."""
    from oracle import torch_port as O
    P, parts, B = _mixed_problem()
    del P
    # re-run the step chain of _mixed_problem with per-episode reductions
    import types
    r = torch.tensor(rows)
    g2 = torch.Generator().manual_seed(13)
    H, ANG, V, C, L, IMG, T = 16, 8, 4, 4, 7, 24, 3
    F = IMG + ANG
    data = dict(a=torch.randn(B, ANG, generator=g2).double(), img=torch.randn(T, B, V, F, generator=g2).double().abs(),
                cand=torch.randn(T, B, C, F, generator=g2).double().abs(), h=torch.randn(B, H, generator=g2).double(),
                c=torch.randn(B, H, generator=g2).double(), ctx=torch.randn(B, L, H, generator=g2).double(),
                tgt=torch.tensor([[0, 1, 2, 3, 1, -1], [1, 1, 0, 2, 3, 0], [2, -1, 1, 0, 0, 3]]))
    Pd = {k: v for k, v in Pm.items() if k != "critic.w"}
    ht, c = data["h"][r], data["c"][r]
    loss = 0.0
    for t in range(T):
        lo, (h1, c), ht, _ = O.envdrop_step(Pd, data["a"][r], data["img"][t][r], data["cand"][t][r], ht, c, data["ctx"][r], None)
        loss = loss + O.masked_cross_entropy(lo, data["tgt"][t][r], None, "none")
        if envdrop_form:
            loss = loss + 0.05 * (h1 @ Pm["critic.w"]) ** 2             # a per-episode RL-style term through the critic
    return loss


@pytest.mark.timeout(420)
@pytest.mark.parametrize("envdrop_form", [True, False], ids=["dot", "dot_over_weight_sum"])
def test_two_rank_self_pace_weighted_loss_equals_big_batch(envdrop_form):
    """BASELINE config 4's loss in miniature (SELF-PACE, curriculum.py:286-314): per-episode losses weighted by the curriculum's
    item weights -- `torch.dot(weight[idx], cur_loss)` (EnvDrop) or the same over the batch's weight sum (other agents) --
    sharded over two ranks == the single-process gradient, and every replica ends up with every episode's (index, loss) for
    the weight update."""
    got = _run_world(_worker_self_pace, (envdrop_form,))
    P, parts, B = _mixed_problem()
    params = [v.clone().requires_grad_(True) for v in P.values()]
    Pm = dict(zip(P.keys(), params))
    weight = torch.linspace(0.1, 1.0, 40, dtype=torch.float64)
    item_idx = torch.tensor([3, 17, 8, 29, 11, 35])
    per_sample = _per_sample_losses(Pm, list(range(B)), envdrop_form)
    loss = torch.dot(weight[item_idx], per_sample)
    if not envdrop_form:
        loss = loss / weight[item_idx].sum()
    loss.backward()
    ref = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params])   # the critic is unused in the CE-only form
    for rank, flat, gi, gl in got:
        assert torch.allclose(flat, ref, rtol=1e-10, atol=1e-12), f"rank {rank}: DP gradient != big-batch gradient"
        order = torch.argsort(gi)
        assert torch.equal(gi[order], torch.sort(item_idx).values)
        assert torch.allclose(gl[order], per_sample.detach()[torch.argsort(item_idx)], rtol=1e-12, atol=0)
    assert torch.equal(got[0][1], got[1][1])


def test_self_pace_helpers_without_a_group():
    sys.path.insert(0, ROOT)
    import vln_amd as vln
    idx, loss = torch.tensor([3, 1]), torch.tensor([0.5, 2.0], requires_grad=True)
    gi, gl = vln.dp.gather_item_losses(idx, loss)
    assert gi is idx and gl is loss                                     # identity on one process
    w = torch.tensor([0.2, 1.0, 0.7, 0.9])
    bl = vln.dp.self_pace_batch_loss(w[idx], loss)                      # curriculum.py:296
    bl.backward()
    assert torch.allclose(bl, torch.tensor(0.9 * 0.5 + 1.0 * 2.0)) and torch.equal(loss.grad, w[idx])


def test_bucket_span_of_requires_adjacent_parameters():
    sys.path.insert(0, ROOT)
    import vln_amd as vln
    ps = [torch.nn.Parameter(torch.zeros(n)) for n in (3, 5, 7)]
    b = vln.dp.GradBucket(ps)
    assert b.span_of(ps[1:]) == (3, 15) and b.span_of([ps[0]]) == (0, 3)
    with pytest.raises(ValueError):
        b.span_of([ps[0], ps[2]])
    b.start_allreduce(ps[1:]); b.allreduce()          # no process group: both are no-ops
    assert not b.reducer.pending
