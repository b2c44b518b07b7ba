"""GPU: operator-level parity of the HIP kernels (through the C ABI) against
plain fp64 torch math on the same seeded inputs.  fp32 tolerance 1e-4
(north_star), bf16-streamed operands 1e-2."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vln():
    import vln_amd
    vln_amd._lib.load()
    return vln_amd


def dev():
    return torch.device("cuda:0")


from parity import check, rel_err


# (the last three: TALL products, M = B * L rows -- the encoder's input projection, the projected context K = ctx W_in; ragged M and N)
@pytest.mark.parametrize("M,N,K", [(64, 2048, 2752), (64, 2176, 512), (64, 512, 1024), (4, 48, 40), (7, 1, 64),
                                   (130, 100, 36), (64, 512, 2176), (256, 64, 128), (5120, 2048, 256), (5000, 512, 512), (1100, 96, 256)])
@pytest.mark.parametrize("wdt", [torch.float32, torch.bfloat16])
def test_linear_fwd(vln, M, N, K, wdt):
    g = torch.Generator().manual_seed(M * 131 + N * 7 + K)
    x = torch.randn(M, K, generator=g); w = torch.randn(N, K, generator=g) / K ** 0.5; b = torch.randn(N, generator=g)
    wq = w.to(wdt)
    ref = torch.tanh(x.double() @ wq.double().t() + b.double())
    # bf16 path: only the streamed weight is bf16; x is split hi+lo so it is exact to ~2^-17
    y = vln.ops.linear_fwd(x.to(dev()), wq.to(dev()), b.to(dev()), vln.ops.ACT_TANH)
    tol = 1e-4 if wdt == torch.float32 else 2e-4
    check(y, ref, tol, "y")


@pytest.mark.parametrize("M,N,K", [(64, 2048, 2752), (64, 2176, 512), (1152, 1024, 2176), (64, 512, 1024), (4, 48, 40), (7, 1, 64),
                                   (130, 100, 36), (5120, 512, 512), (5000, 2048, 256), (1100, 96, 512)])
def test_linear_fwd_split_fp32_weights(vln, M, N, K):
    """VLN_F32S (round 4): fp32 weights multiplied on the bf16 matrix pipe with BOTH operands split hi + lo (three products, the
    dropped lo * lo term is 2^-16 relative) -- the arithmetic of the bf16 mode's fp32-streamed matrices.  fp32-grade: 1e-4 of the
    fp64 product like the exact fp32 MFMA path, at the aligned decoder / BN-MLP shapes and at unaligned ones (bounds-checked form)."""
    g = torch.Generator().manual_seed(M * 131 + N * 7 + K + 1)
    x = torch.randn(M, K, generator=g); w = torch.randn(N, K, generator=g) / K ** 0.5; b = torch.randn(N, generator=g)
    ref = torch.tanh(x.double() @ w.double().t() + b.double())
    y = vln.ops.linear_fwd(x.to(dev()), w.to(dev()), b.to(dev()), vln.ops.ACT_TANH, split=True)
    check(y, ref, 1e-4, "y (split fp32 weights)")
    y_bf = vln.ops.linear_fwd(x.to(dev()), w.bfloat16().to(dev()), b.to(dev()), vln.ops.ACT_TANH)
    assert rel_err(y, ref) < 0.1 * max(rel_err(y_bf, ref), 1e-5) or rel_err(y, ref) < 2e-5     # an order below the bf16-streamed form
    # VLN_F32X: three bf16 pieces per operand, six products -- as close to the fp64 product as the exact fp32 MFMA is (its own
    # accumulation rounding is what remains), an order below the three-product form
    y6 = vln.ops.linear_fwd(x.to(dev()), w.to(dev()), b.to(dev()), vln.ops.ACT_TANH, split="x6")
    y32 = vln.ops.linear_fwd(x.to(dev()), w.to(dev()), b.to(dev()), vln.ops.ACT_TANH)
    check(y6, ref, 1e-5, "y (six-product fp32-grade form)")
    check(y32, ref, 1e-5, "y (exact fp32 MFMA)")
    assert rel_err(y6, ref) < 4 * max(rel_err(y32, ref), 5e-7), (rel_err(y6, ref), rel_err(y32, ref))


@pytest.mark.parametrize("M,N,K", [(1152, 1024, 2176), (1152, 2176, 1024), (5120, 256, 2048), (1100, 1000, 512), (300, 192, 128), (8064, 64, 64)])
def test_tall_products_row_block_tiling_is_bit_identical(vln, M, N, K):
    """gemm_rows.h (round 5): tall activations are tiled in 16-row blocks so that the workgroups fill whole rounds of the CUs
    (the BN-MLP's [1152, 2176] x [1024, 2176]^T: 256 workgroups of 80 / 64 rows instead of 288 of 64).  Same operand paths and the
    same MFMA sequence per output element as gemm_nt's 64-row tiles: bit-identical results in all four weight forms, with bias,
    activation and the accumulate flag, at aligned and ragged M / N."""
    lib = vln._lib.load()
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).to(dev()); w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev()); b = torch.randn(N, generator=g).to(dev())
    y0 = torch.randn(M, N, generator=g).to(dev())
    for wt, split in ((w, False), (w.bfloat16(), False), (w, True), (w, "x6")):
        outs = []
        for keep64 in (1, 0):
            vln._lib.check(lib.vln_set_tunable(12, keep64), "vln_set_tunable")
            try:
                y = vln.ops.linear_fwd(x, wt, b, vln.ops.ACT_TANH, split=split)
                ya = vln.ops.linear_fwd(x, wt, b, vln.ops.ACT_ACCUM, out=y0.clone(), split=split)
            finally:
                vln._lib.check(lib.vln_set_tunable(12, 0), "vln_set_tunable")
            outs.append((y, ya))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), (M, N, K, wt.dtype, split)
    ref = torch.tanh(x.double() @ w.double().t() + b.double())
    check(vln.ops.linear_fwd(x, w, b, vln.ops.ACT_TANH), ref, 1e-4, "y (row-block tiling)")


@pytest.mark.parametrize("M,N,K,wdt", [(64, 2048, 2752, torch.bfloat16), (128, 2048, 3072, torch.float32), (8, 512, 512, torch.bfloat16)])
def test_linear_fwd_slabs_sum_to_the_product(vln, M, N, K, wdt):
    """ops.linear_fwd_slabs / vln_linear_fwd_slabs: the product left as its split-K partial slabs (what the one-call decoder steps hand
    to the LSTM's pointwise launch): their sum in slab order IS what linear_fwd returns (the reduce launch adds them in that order)."""
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).to(dev()); w = (torch.randn(N, K, generator=g) / K ** 0.5).to(wdt).to(dev())
    slabs = vln.ops.linear_fwd_slabs(x, w)
    assert slabs.dim() == 3 and tuple(slabs.shape[1:]) == (M, N) and slabs.shape[0] >= 1
    acc = slabs[0].clone()
    for s_ in range(1, slabs.shape[0]):
        acc = acc + slabs[s_]
    y = vln.ops.linear_fwd(x, w)
    check(acc, y.double(), 1e-6, "sum of the slabs vs linear_fwd")
    check(acc, x.double() @ w.double().t(), 1e-4 if wdt == torch.float32 else 1e-2, "sum of the slabs vs the fp64 product")


def test_linear_fwd_strided_x(vln):
    g = torch.Generator().manual_seed(5)
    big = torch.randn(64, 300, generator=g).to(dev())
    x = big[:, 44:44 + 128]
    w = torch.randn(96, 128, generator=g).to(dev())
    y = vln.ops.linear_fwd(x, w)
    check(y, x.double() @ w.double().t(), 1e-4, "y")


@pytest.mark.parametrize("Mt,N,K", [(384, 2048, 2752), (64, 512, 512), (100, 48, 40), (5120, 96, 72), (3, 1, 5), (448, 2176, 512),
                                    (5120, 1024, 256), (77, 132, 260)])
def test_linear_wgrad(vln, Mt, N, K):
    g = torch.Generator().manual_seed(Mt + N + K)
    dy = torch.randn(Mt, N, generator=g); x = torch.randn(Mt, K, generator=g)
    ref = dy.double().t() @ x.double()
    out = vln.ops.linear_wgrad(dy.to(dev()), x.to(dev()))
    check(out, ref, 1e-4, "out")
    out2 = vln.ops.linear_wgrad(dy.to(dev()), x.to(dev()), out=out, accumulate=True)
    check(out2, 2 * ref, 1e-4, "out2")
    cs = vln.ops.colsum(dy.to(dev()))
    check(cs, dy.double().sum(0), 1e-4, "cs")
    # split-bf16 form (bf16 compute mode): hi + lo planes of both operands, 2^-16 relative per product
    o3 = vln.ops.linear_wgrad(dy.to(dev()), x.to(dev()), split_bf16=True)
    check(o3, ref, 5e-5, "o3")
    o3b = vln.ops.linear_wgrad(dy.to(dev()), x.to(dev()), out=o3, accumulate=True, split_bf16=True)
    check(o3b, 2 * ref, 5e-5, "o3b")


def test_select_rows_multi_equals_torch_indexing(vln):
    """ops.select_rows_multi (round 6): [src_t[arange(B), index_t]] for all steps in one launch == torch's advanced indexing, negative
    indices counting from the end; an index out of range gives a zero row and raises the library's sticky gather word."""
    g = torch.Generator().manual_seed(79)
    B, F = 37, 2176
    srcs = [torch.randn(B, C_, F, generator=g).to(dev()) for C_ in (3, 8, 16, 1)]
    idx = [torch.randint(-C_, C_, (B,), generator=g).to(dev()) for C_ in (3, 8, 16, 1)]
    outs = vln.ops.select_rows_multi(srcs, idx)
    ar = torch.arange(B, device=dev())
    for t, (x, i, o) in enumerate(zip(srcs, idx, outs)):
        assert torch.equal(o, x[ar, i]), f"step {t}"
    lib = vln._lib.load()
    vln._lib.check(lib.vln_persistent_check(), "vln_persistent_check")
    bad = idx[1].clone(); bad[5] = 8
    o = vln.ops.select_rows_multi([srcs[1]], [bad])[0]
    torch.cuda.synchronize()
    assert float(o[5].abs().max()) == 0.0 and torch.equal(o[4], srcs[1][4, bad[4]])
    with pytest.raises(vln._lib.VlnError):
        vln._lib.check(lib.vln_persistent_check(), "vln_persistent_check")
    lib.vln_persistent_check()                              # (reading the sticky words clears them)


def test_rollout_monitor_loss_equals_the_sum_of_the_step_losses(vln):
    """losses.RolloutMonitorLoss (round 6): the Self-Monitor agent's loss of a whole rollout in ONE launch each way == the sum of
    `monitor_mixed_loss` over the steps (itself held to the reference's sequence by test_fused_monitor_step_loss_*): value, d logits,
    d progress, the logged progress MSEs; the steps differ in candidate count, ignored rows and ended episodes."""
    g = torch.Generator().manual_seed(78)
    B, T, lam = 128, 7, 0.5
    rl = vln.losses.RolloutMonitorLoss(lam, ignore_index=-100)
    ref, xs, ps, rxs, rps, mses = 0.0, [], [], [], [], []
    for t in range(T):
        C_ = 5 + 2 * t                                          # 5 .. 17: both row forms of the kernel
        ncand = torch.randint(2, C_ + 1, (B,), generator=g)
        mask = (torch.arange(C_)[None, :] >= ncand[:, None]).to(dev())
        tgt = (torch.rand(B, generator=g) * ncand.float()).long()
        tgt[torch.rand(B, generator=g) < 0.08 * t] = -100
        tgt = tgt.to(dev())
        start = (torch.rand(B, generator=g) * 15 + 4).to(dev()); cur = (torch.rand(B, generator=g) * 10 + 0.2).to(dev())
        ended = (torch.rand(B, generator=g) < 0.1 * t).to(dev())
        lg, pr = torch.randn(B, C_, generator=g), torch.rand(B, 1, generator=g)
        x, p = lg.to(dev()).requires_grad_(True), pr.to(dev()).requires_grad_(True)
        rx, rp = lg.to(dev()).requires_grad_(True), pr.to(dev()).requires_grad_(True)
        l_t, mse_t = vln.losses.monitor_mixed_loss(rx, tgt, mask, rp, start, cur, ended, t, lam, ignore_index=-100)
        ref = ref + l_t
        mses.append(mse_t)
        rl.add(x, tgt, mask, p, start, cur, ended)
        xs.append(x); ps.append(p); rxs.append(rx); rps.append(rp)
    loss = rl.sum()
    ref.backward()
    loss.backward()
    check(loss, ref.detach(), 1e-6, "loss")
    check(rl.progress_mse[1:], torch.stack(mses)[1:], 1e-6, "progress MSE per step")
    for t in range(T):
        check(xs[t].grad, rxs[t].grad, 1e-6, f"d logits step {t}")
        check(ps[t].grad, rps[t].grad, 1e-6, f"d progress step {t}", floor=1e-12)


def test_rollout_ce_mean_per_step_equals_the_per_step_criterion(vln):
    """losses.RolloutCE.mean_per_step (round 6): sum_t CrossEntropyLoss(ignore_index)(masked logits_t, target_t) with the default mean
    reduction on every step's batch (follower.py:62,123-139) in ONE launch each way == torch's criterion step by step, value and
    d logits; steps differ in their candidate count and in their number of ignored rows."""
    g = torch.Generator().manual_seed(77)
    B, T = 64, 7
    ce = vln.losses.RolloutCE(ignore_index=-100)
    ref, lgs, rlgs = 0.0, [], []
    for t in range(T):
        C_ = 4 + 3 * t                                         # 4 .. 22: both row forms of the kernel
        ncand = torch.randint(2, C_ + 1, (B,), generator=g)
        mask = torch.arange(C_)[None, :] >= ncand[:, None]
        tgt = (torch.rand(B, generator=g) * ncand.float()).long()
        tgt[torch.rand(B, generator=g) < 0.1 * t] = -100       # ended episodes
        lg = torch.randn(B, C_, generator=g)
        r = lg.clone().double().requires_grad_(True)
        ref = ref + torch.nn.functional.cross_entropy(r.masked_fill(mask, float("-inf")), tgt, ignore_index=-100)
        x = lg.to(dev()).requires_grad_(True)
        ce.add(x, tgt.to(dev()), mask.to(dev()))
        lgs.append(x); rlgs.append(r)
    loss = ce.mean_per_step(0.5)
    (ref * 0.5).backward()
    loss.backward()
    check(loss, ref.detach() * 0.5, 1e-6, "loss")
    for t in range(T):
        check(lgs[t].grad, rlgs[t].grad, 1e-5, f"d logits step {t}")


@pytest.mark.parametrize("split", [False, True, "bf16"])
def test_wgrad_grouped(vln, split):
    """All weight gradients of a module from one call (one launch in the bf16 forms): the decoder's seven
    products over the same (steps x batch) rows, strided operands, accumulate and overwrite mixed.  False: exact fp32 MFMA;
    True: split-bf16 operands (three MFMAs, 5e-5); "bf16": plain bf16 operands, fp32 accumulation (one MFMA, 2^-9 per operand)."""
    before = vln.ops.get_wgrad_precision()
    vln.ops.set_wgrad_precision("bf16" if split == "bf16" else "split")
    try:
        _wgrad_grouped(vln, split)
    finally:
        vln.ops.set_wgrad_precision(before)


def _wgrad_grouped(vln, split):
    tol = {False: 1e-5, True: 5e-5, "bf16": 6e-3}[split]
    split = bool(split)
    g = torch.Generator().manual_seed(11)
    Mt = 448
    shapes = [(2048, 2240), (2176, 512), (2048, 512), (512, 1024), (512, 512), (64, 128), (132, 260)]
    wb = vln.ops.WgradBatch(split)
    refs, outs = [], []
    for i, (N, K) in enumerate(shapes):
        dy = torch.randn(Mt, N + 4, generator=g).to(dev())[:, :N]
        x = torch.randn(Mt, K + 8, generator=g).to(dev())[:, 4:4 + K]
        acc = bool(i % 2)
        out = torch.randn(N, K, generator=g).to(dev()) if acc else torch.empty(N, K, device=dev())
        ref = dy.double().t() @ x.double() + (out.double() if acc else 0)
        wb.add(dy, x, out, acc)
        refs.append(ref); outs.append(out)
    wb.run()
    for o, r in zip(outs, refs):
        check(o, r, tol, "o")


@pytest.mark.parametrize("R,D", [(128, 2176), (1024, 1024), (1792, 1024), (5, 8), (1000, 128), (601, 2176)])   # >= 512 rows: the row-chunked form
@pytest.mark.parametrize("relu", [False, True])
def test_batch_norm_kernel(vln, R, D, relu):
    """vln_bn_fwd / vln_bn_bwd vs torch.nn.functional.batch_norm (+ relu) in fp64: training mode (batch statistics,
    running statistics and num_batches_tracked updated in place) and eval mode, outputs and all three gradients."""
    from vln_amd import functional as Fh
    g = torch.Generator().manual_seed(R + D)
    x = (torch.randn(R, D, generator=g) * 0.7 + 0.3)
    w, b = torch.rand(D, generator=g) + 0.5, torch.randn(D, generator=g)
    rm0, rv0 = torch.randn(D, generator=g) * 0.1, torch.rand(D, generator=g) + 0.5
    r = torch.randn(R, D, generator=g)
    for training in (True, False):
        x64 = x.double().requires_grad_(True); w64 = w.double().requires_grad_(True); b64 = b.double().requires_grad_(True)
        rm, rv = rm0.double().clone(), rv0.double().clone()
        ref = torch.nn.functional.batch_norm(x64, rm, rv, w64, b64, training, 0.1, 1e-5)
        if relu:
            ref = torch.relu(ref)
        (ref * r.double()).sum().backward()
        d = dev()
        xd = x.to(d).requires_grad_(True); wd = w.to(d).requires_grad_(True); bd = b.to(d).requires_grad_(True)
        rmd, rvd = rm0.to(d).clone(), rv0.to(d).clone()
        nbt = torch.zeros((), dtype=torch.int64, device=d)
        y = Fh.batch_norm(xd, wd, bd, rmd, rvd, nbt, training, 0.1, 1e-5, relu)
        (y * r.to(d)).sum().backward()
        check(y, ref.detach(), 1e-5, "y")
        check(xd.grad, x64.grad, 1e-4, "xd.grad"); check(wd.grad, w64.grad, 1e-4, "wd.grad"); check(bd.grad, b64.grad, 1e-4, "bd.grad")
        check(rmd, rm, 1e-5, "rmd"); check(rvd, rv, 1e-5, "rvd")
        assert int(nbt) == (1 if training else 0)


@pytest.mark.parametrize("B,C", [(64, 8), (300, 15), (5, 1)])
def test_sample_action_kernel(vln, B, C):
    """losses.sample_action vs torch.distributions.Categorical on the masked softmax: log-prob and entropy of a given
    action, their gradients w.r.t. the logits, and the empirical distribution of the kernel's own draws."""
    g = torch.Generator().manual_seed(B + C)
    logits = torch.randn(B, C, generator=g) * 2
    ncand = torch.randint(1, C + 1, (B,), generator=g)
    mask = torch.arange(C)[None, :] >= ncand[:, None]
    act = (torch.rand(B, generator=g) * ncand.float()).long()
    l64 = logits.clone().requires_grad_(True)          # fp32 reference: clamp_probs' eps is the dtype's (1.19e-7 here)
    dist = torch.distributions.Categorical(torch.softmax(l64.masked_fill(mask, -float("inf")), 1))
    lp_ref, en_ref = dist.log_prob(act), dist.entropy()
    w1, w2 = torch.randn(B, generator=g), torch.randn(B, generator=g)
    ((lp_ref * w1).sum() + (en_ref * w2).sum()).backward()
    d = dev()
    ld = logits.to(d).requires_grad_(True)
    a, lp, en = vln.losses.sample_action(ld, mask.to(d), act.to(d))
    assert torch.equal(a.cpu(), act)
    assert (lp.cpu() - lp_ref.detach()).abs().max() < 1e-5 and (en.cpu() - en_ref.detach()).abs().max() < 1e-5
    ((lp * w1.to(d)).sum() + (en * w2.to(d)).sum()).backward()
    assert (ld.grad.cpu() - l64.grad).abs().max() < 1e-4 * max(1.0, l64.grad.abs().max().item())
    # draws: never a masked slot, frequencies follow the probabilities
    row = torch.tensor([[1.0, 0.0, -1.0, 2.0] + [0.0] * 4]).repeat(4096, 1).to(d)
    rmask = torch.tensor([[False, False, False, False] + [True] * 4]).repeat(4096, 1).to(d)
    draws, lpd, _ = vln.losses.sample_action(row, rmask)
    assert int(draws.max()) <= 3
    freq = torch.bincount(draws.cpu(), minlength=4).float() / 4096
    p = torch.softmax(torch.tensor([1.0, 0.0, -1.0, 2.0]), 0)
    assert (freq[:4] - p).abs().max().item() < 0.03
    check(lpd, torch.log(p)[draws.cpu()], 1e-5, "lpd")


@pytest.mark.parametrize("T,B,with_ent", [(7, 64, True), (35, 64, True), (5, 3, False), (1, 130, True)])
def test_a2c_loss_kernel_matches_the_restated_sweep(vln, T, B, with_ent):
    """vln_a2c_loss_fwd/bwd vs oracle/torch_port.py::a2c_loss (the restatement of envdrop.py:235-264, pinned by the
    reference's own rollout tape): loss, `total`, and the gradients w.r.t. log-probs, values and entropies."""
    from oracle import torch_port as O
    g = torch.Generator().manual_seed(T * 100 + B)
    lens = torch.randint(1, T + 1, (B,), generator=g)
    masks = [(t < lens) for t in range(T)]
    ended = torch.rand(B, generator=g) < 0.7
    rewards = [torch.randn(B, generator=g).sign() * m for m in masks]
    mk = lambda: [torch.randn(B, generator=g, dtype=torch.float64).requires_grad_(True) for _ in range(T)]
    lp, en, vl = mk(), mk(), mk()
    last_v = torch.randn(B, generator=g, dtype=torch.float64)
    for norm, per in (("total", False), ("batch", True), ("none", False)):
        ref_en = en if with_ent else [torch.zeros(B, dtype=torch.float64) for _ in range(T)]
        ref, total = O.a2c_loss(lp, ref_en, vl, [r.double() for r in rewards], masks, last_v, ended, 0.9, norm, per)
        w = torch.randn(B, generator=g, dtype=torch.float64) if per else None
        for x in lp + en + vl:
            x.grad = None
        ((ref * w).sum() if per else ref).backward()
        d = dev()
        dlp = [x.detach().float().to(d).requires_grad_(True) for x in lp]
        den = [x.detach().float().to(d).requires_grad_(True) for x in en] if with_ent else None
        dvl = [x.detach().float().to(d).requires_grad_(True) for x in vl]
        out, tot = vln.losses.a2c_loss(dlp, den, dvl, [r.to(d) for r in rewards], [m.to(d) for m in masks], last_v.float().to(d),
                                       ended.to(d), 0.9, norm, per)
        assert abs(float(tot) - total) < 1e-6
        check(out, ref.detach(), 1e-5, "out")
        ((out * w.float().to(d)).sum() if per else out).backward()
        for a, b in zip(dlp + dvl + (den or []), lp + vl + (en if with_ent else [])):
            if b.grad is None:
                assert a.grad is None or a.grad.abs().max().item() == 0.0
            else:
                check(a.grad, b.grad, 1e-5, "a.grad")


@pytest.mark.parametrize("rows", [448, 5120, 7])
def test_colsum_grouped(vln, rows):
    """All bias gradients of a module in one launch (two when the rows are split): strided inputs, a shared sum written
    to two destinations, accumulate and overwrite mixed, an odd shape through the single-matrix kernel."""
    g = torch.Generator().manual_seed(rows)
    big = torch.randn(rows, 2048 + 64 + 8, generator=g).to(dev())
    a1, a2 = big[:, :2048], big[:, 2048:2112]
    a3 = torch.randn(rows, 50, generator=g).to(dev())
    o1 = torch.randn(2048, generator=g).to(dev()); o1b = o1.clone()
    o2 = torch.empty(64, device=dev()); o3 = torch.empty(50, device=dev())
    r1 = a1.double().sum(0) + o1.double()
    cb = vln.ops.ColsumBatch()
    cb.add(a1, o1, o1b, True)
    cb.add(a2, o2, None, False)
    cb.add(a3, o3, None, False)
    cb.run()
    assert rel_err(o1, r1) < 1e-5 and torch.equal(o1, o1b)
    check(o2, a2.double().sum(0), 1e-5, "o2"); check(o3, a3.double().sum(0), 1e-5, "o3")


@pytest.mark.parametrize("N,K", [(2048, 2752), (48, 40), (1, 7)])
def test_transpose_and_cast(vln, N, K):
    w = torch.randn(N, K).to(dev())
    assert torch.equal(vln.ops.transpose_cast(w), w.t().contiguous())
    assert torch.equal(vln.ops.transpose_cast(w, torch.bfloat16), w.t().contiguous().bfloat16())
    assert torch.equal(vln.ops.cast_copy(w), w.bfloat16())


@pytest.mark.parametrize("odt", [torch.bfloat16, torch.float32])
def test_shadow_refresh_grouped(vln, odt):
    """vln_shadow_refresh: all shadows of a module in one launch -- plain / transposed / both, a summed pair (b_ih + b_hh),
    shapes that take the 16-byte path (everything a multiple of 4), ragged ones that take the scalar path, partial 64x64
    tiles and a strided source (a row block of a wider matrix); every element equals torch's cast of the fp32 value."""
    d = "cuda:0"
    g = torch.Generator().manual_seed(3)
    sb = vln.ops.ShadowBatch()
    exp = []
    for N, K, kind in ((2048, 2752, "both"), (2176, 512, "t"), (64, 128, "n"), (100, 36, "both"), (130, 67, "both"), (1, 2048, "n"),
                       (4, 4, "both")):
        src = torch.randn(N, K, generator=g).to(d)
        dst = torch.empty(N, K, dtype=odt, device=d) if kind in ("n", "both") else None
        dst_t = torch.empty(K, N, dtype=odt, device=d) if kind in ("t", "both") else None
        sb.add(src, dst, dst_t)
        exp.append((src, dst, dst_t))
    a, b = torch.randn(1, 2048, generator=g).to(d), torch.randn(1, 2048, generator=g).to(d)
    dsum = torch.empty(1, 2048, dtype=odt, device=d)
    sb.add(a, dsum, None, src2=b)
    wide = torch.randn(96, 512, generator=g).to(d)
    blk = wide[:, 128:384]                                   # strided source rows
    dblk, dblk_t = torch.empty(96, 256, dtype=odt, device=d), torch.empty(256, 96, dtype=odt, device=d)
    sb.add(blk, dblk, dblk_t)
    h = sb.run()
    torch.cuda.synchronize()
    for src, dst, dst_t in exp + [(blk, dblk, dblk_t)]:
        if dst is not None:
            assert torch.equal(dst, src.to(odt))
        if dst_t is not None:
            assert torch.equal(dst_t, src.t().contiguous().to(odt))
    assert torch.equal(dsum, (a + b).to(odt))
    exp[0][0].mul_(2.0)                                      # replay refreshes from the same addresses
    vln.ops.ShadowBatch.replay(h)
    assert torch.equal(exp[0][1], exp[0][0].to(odt)) and torch.equal(exp[0][2], exp[0][0].t().contiguous().to(odt))


@pytest.mark.parametrize("B,S,D", [(64, 36, 2176), (64, 80, 512), (64, 8, 2176), (4, 9, 48), (3, 5, 50)])
@pytest.mark.parametrize("cdt", [torch.float32, torch.bfloat16])
def test_attention_fwd_bwd(vln, B, S, D, cdt):
    if cdt == torch.bfloat16 and D % 8:
        pytest.skip("bf16 path wants D % 8 == 0 for the vector loads; scalar path covered by fp32")
    g = torch.Generator().manual_seed(B * S + D)
    ctx = (torch.randn(B, S, D, generator=g) * 0.5).to(cdt)
    vec = torch.randn(B, D, generator=g) / D ** 0.5
    mask = torch.zeros(B, S, dtype=torch.bool)
    for b in range(B):
        mask[b, max(1, S - (b % S)):] = True
    c64 = ctx.double().requires_grad_(True); v64 = vec.double().requires_grad_(True)
    logits = torch.einsum("bsd,bd->bs", c64, v64)
    attn = torch.softmax(logits.masked_fill(mask, -float("inf")), 1)
    wc = torch.einsum("bs,bsd->bd", attn, c64)
    r = torch.randn(B, D, generator=g).double(); ra = torch.randn(B, S, generator=g).double()
    ((wc * r).sum() + (attn * ra).sum()).backward()

    cd, vd = ctx.to(dev()), vec.to(dev())
    dots = vln.ops.attn_dot(cd, vd)
    tol = 1e-4 if cdt == torch.float32 else 1e-2
    check(dots, logits.detach(), tol, "dots")
    out, at = vln.ops.attn_softmax_wsum(cd, dots, mask.to(dev()))
    check(at, attn.detach(), tol, "at"); check(out, wc.detach(), tol, "out")
    assert at[mask.to(dev())].abs().max().item() == 0.0
    # backward: dalpha = ctx . dwc ; dvec, dctx
    dwc = r.float().to(dev())
    dalpha = vln.ops.attn_dot(cd, dwc)
    dctx = torch.zeros(B, S, D, device=dev())
    dvec, dl = vln.ops.attn_bwd(cd, at, dalpha, ra.float().to(dev()), dwc, vd, dctx, want_dl=True)
    check(dvec, v64.grad, tol, "dvec")
    check(dctx, c64.grad, tol, "dctx")
    # accumulate semantics
    vln.ops.attn_bwd(cd, at, dalpha, ra.float().to(dev()), dwc, vd, dctx)
    check(dctx, 2 * c64.grad, tol, "dctx")
    # plain weighted sum
    w = torch.randn(B, S, generator=g)
    check(vln.ops.rows_wsum(cd, w.to(dev())), torch.einsum("bs,bsd->bd", w.double(), ctx.double()), tol, "vln.ops.rows_wsum(cd, w.to(dev()))")


@pytest.mark.parametrize("B,S,D", [(64, 36, 2176), (64, 80, 512), (64, 8, 2176), (16, 12, 1024), (5, 33, 520), (4, 9, 48),
                                   (3, 5, 50), (2, 100, 512), (7, 1, 64)])
@pytest.mark.parametrize("cdt", [torch.float32, torch.bfloat16])
def test_attention_rows_one_launch(vln, B, S, D, cdt):
    """vln_attn_fwd_rows / vln_attn_bwd_rows (register-resident context block, one launch; shapes outside the
    configurations fall back to two launches) and the once-per-rollout context gradient vln_attn_dctx_deferred."""
    if cdt == torch.bfloat16 and D % 8:
        pytest.skip("bf16 path wants D % 8 == 0 for the vector loads; scalar path covered by fp32")
    g = torch.Generator().manual_seed(B * S + D + 1)
    ctx = (torch.randn(B, S, D, generator=g) * 0.5).to(cdt)
    mask = torch.zeros(B, S, dtype=torch.bool)
    for b in range(B):
        mask[b, max(1, S - (b % S)):] = True
    T = 3
    c64 = ctx.double().requires_grad_(True)
    tol = 1e-4 if cdt == torch.float32 else 1e-2
    cd = ctx.to(dev())
    terms, keep, total = [], [], 0.0
    big = torch.zeros(B, 2 * D + 8, device=dev())           # strided query / gradient rows
    for t in range(T):
        vec = torch.randn(B, D, generator=g) / D ** 0.5
        v64 = vec.double().requires_grad_(True)
        logits = torch.einsum("bsd,bd->bs", c64, v64)
        attn = torch.softmax(logits.masked_fill(mask, -float("inf")), 1)
        wc = torch.einsum("bs,bsd->bd", attn, c64)
        r = torch.randn(B, D, generator=g).double(); ra = torch.randn(B, S, generator=g).double()
        loss = (wc * r).sum() + (attn * ra).sum()
        gv, = torch.autograd.grad(loss, v64, retain_graph=True)
        total = total + loss
        vd = big[:, 4:4 + D]; vd.copy_(vec.to(dev()))
        out, at = vln.ops.attn_fwd_rows(cd, vd, mask.to(dev()))
        check(at, attn.detach(), tol, "at"); check(out, wc.detach(), tol, "out")
        assert (at * mask.to(dev())).abs().max().item() == 0.0
        dwc = torch.zeros(B, 2 * D, device=dev())[:, :D]; dwc.copy_(r.float().to(dev()))
        dvec, dl = vln.ops.attn_bwd_rows(cd, at, dwc, ra.float().to(dev()), want_dl=True)
        check(dvec, gv, tol, "dvec")
        # same numbers from the two-launch kernels
        dalpha = vln.ops.attn_dot(cd, dwc)
        dvec2, dl2 = vln.ops.attn_bwd(cd, at, dalpha, ra.float().to(dev()), want_dl=True)
        check(dvec, dvec2, 1e-5, "dvec"); check(dl, dl2, 1e-5, "dl")
        q = vd.clone()
        keep += [at, dl, dwc, q]
        terms.append((at.data_ptr(), dl.data_ptr(), dwc.data_ptr(), q.data_ptr()))
    if D % 4:
        return                                              # the deferred kernel needs 16-byte rows
    total.backward()
    dctx = torch.empty(B, S, D, device=dev())
    vln.ops.attn_dctx_deferred([x[0] for x in terms], [x[1] for x in terms], [x[2] for x in terms], 2 * D,
                               [x[3] for x in terms], D, dctx)
    check(dctx, c64.grad, tol, "dctx")
    vln.ops.attn_dctx_deferred([x[0] for x in terms], [x[1] for x in terms], [x[2] for x in terms], 2 * D,
                               [x[3] for x in terms], D, dctx, accumulate=True)
    check(dctx, 2 * c64.grad, tol, "dctx")


@pytest.mark.parametrize("B,S,D", [(64, 36, 2176), (64, 80, 512), (48, 17, 2176), (9, 80, 512), (64, 33, 544)])
@pytest.mark.parametrize("cdt", [torch.float32, torch.bfloat16])
def test_attention_rows_four_workgroups_per_row(vln, B, S, D, cdt):
    """csrc/attention_split.h: a row's [S, D] block on FOUR workgroups (columns split, the partial row dots exchanged once as
    data-tagged granules) against fp64 and against the one-workgroup kernel, forward and backward -- twelve launches in a row
    on ONE exchange buffer (tags = a per-row launch count that every launch bumps), odd S, B not a multiple of 8, masked rows."""
    g = torch.Generator().manual_seed(B + S + D)
    ctx = (torch.randn(B, S, D, generator=g) * 0.5).to(cdt)
    mask = torch.zeros(B, S, dtype=torch.bool)
    for b in range(B):
        mask[b, max(1, S - (b % S)):] = True
    tol = 1e-4 if cdt == torch.float32 else 1e-2
    cd, md = ctx.to(dev()), mask.to(dev())
    sync = vln.ops.attn_sync_buffer(B, dev())
    c64 = ctx.double()
    for t in range(6):
        vec = torch.randn(B, D, generator=g) / D ** 0.5
        r = torch.randn(B, D, generator=g); ra = torch.randn(B, S, generator=g)
        v64 = vec.double().requires_grad_(True)
        attn = torch.softmax(torch.einsum("bsd,bd->bs", c64, v64).masked_fill(mask, -float("inf")), 1)
        wc = torch.einsum("bs,bsd->bd", attn, c64)
        gv, = torch.autograd.grad((wc * r.double()).sum() + (attn * ra.double()).sum(), v64)
        out, at = vln.ops.attn_fwd_rows(cd, vec.to(dev()), md, sync=sync)
        out1, at1 = vln.ops.attn_fwd_rows(cd, vec.to(dev()), md)
        check(at, attn.detach(), tol, "attn (split)"); check(out, wc.detach(), tol, "out (split)")
        check(at, at1, 1e-5, "attn split vs one workgroup"); check(out, out1, 1e-5, "out split vs one workgroup")
        assert (at * md).abs().max().item() == 0.0
        dvec, dl = vln.ops.attn_bwd_rows(cd, at, r.to(dev()), ra.to(dev()), want_dl=True, sync=sync)
        dvec1, dl1 = vln.ops.attn_bwd_rows(cd, at, r.to(dev()), ra.to(dev()), want_dl=True)
        check(dvec, gv, tol, "dvec (split)")
        check(dvec, dvec1, 1e-5, "dvec split vs one workgroup"); check(dl, dl1, 1e-5, "dl split vs one workgroup")
    torch.cuda.synchronize()
    vln._lib.check(vln._lib.load().vln_persistent_check(), "vln_persistent_check")      # no bounded wait timed out
    assert int(sync[0].item()) == 12                                                   # row 0's launch count


def test_attention_dctx_deferred_many_steps(vln):
    """More steps than one launch stages in LDS (chunks of <= 14 at D = 512): RL rollouts run up to 35 steps."""
    B, S, D, T = 8, 20, 512, 35
    g = torch.Generator().manual_seed(9)
    al = torch.rand(T, B, S, generator=g).to(dev()); dl = torch.randn(T, B, S, generator=g).to(dev())
    gw = torch.randn(T, B, 2 * D, generator=g).to(dev()); q = torch.randn(T, B, D, generator=g).to(dev())
    ref = torch.einsum("tbs,tbd->bsd", al.double(), gw[:, :, :D].double()) + torch.einsum("tbs,tbd->bsd", dl.double(), q.double())
    out = torch.empty(B, S, D, device=dev())
    vln.ops.attn_dctx_deferred([al[t].data_ptr() for t in range(T)], [dl[t].data_ptr() for t in range(T)],
                               [gw[t].data_ptr() for t in range(T)], 2 * D, [q[t].data_ptr() for t in range(T)], D, out)
    check(out, ref, 1e-5, "out")


def test_lstm_pointwise(vln):
    B, H = 64, 512
    g = torch.Generator().manual_seed(3)
    slabs = torch.randn(3, B, 4 * H, generator=g)
    bi, bh, c0 = torch.randn(4 * H, generator=g), torch.randn(4 * H, generator=g), torch.randn(B, H, generator=g)
    gates = slabs.sum(0).double().requires_grad_(True)
    c064 = c0.double().requires_grad_(True)
    G = gates + bi.double() + bh.double()
    i, f, gg, o = torch.sigmoid(G[:, :H]), torch.sigmoid(G[:, H:2 * H]), torch.tanh(G[:, 2 * H:3 * H]), torch.sigmoid(G[:, 3 * H:])
    c1 = f * c064 + i * gg; h1 = o * torch.tanh(c1)
    p, seed, off = 0.5, 1234, 77
    mask = vln.ops.dropout_mask(B * H, seed, off, p, dev()).view(B, H)
    keep = (mask > 0).float().mean().item()
    assert abs(keep - 0.5) < 0.02 and set(mask.unique().tolist()) <= {0.0, 2.0}
    r1, r2, r3 = (torch.randn(B, H, generator=g).double() for _ in range(3))
    ((h1 * r1).sum() + (c1 * r2).sum() + (h1 * mask.cpu().double() * r3).sum()).backward()
    d = dev()
    oh1, oc1, act, tc, hd = vln.ops.lstm_pointwise_fwd(slabs.to(d), bi.to(d), bh.to(d), c0.to(d), seed, off, p, True)
    check(oh1, h1.detach(), 1e-5, "oh1"); check(oc1, c1.detach(), 1e-5, "oc1")
    check(hd, (h1.detach() * mask.cpu().double()), 1e-5, "hd")
    dg, dc0 = vln.ops.lstm_pointwise_bwd(r1.float().to(d), r3.float().to(d), r2.float().to(d), act, tc, c0.to(d), seed, off, p)
    check(dg, gates.grad, 1e-4, "dg"); check(dc0, c064.grad, 1e-4, "dc0")


def test_feature_dropout(vln):
    B, V, IMG, ANG = 64, 36, 2048, 128
    x = torch.rand(B, V, IMG + ANG).to(dev()) + 0.1
    x0 = x.clone()
    cp = torch.empty(B, V, IMG + ANG, dtype=torch.bfloat16, device=dev())
    vln.ops.feat_dropout_inplace(x, IMG, ANG, 99, 5, 0.3, cp)
    assert torch.equal(x[..., IMG:], x0[..., IMG:])                 # angle tail untouched
    img = x[..., :IMG]
    kept = img != 0
    assert abs(kept.float().mean().item() - 0.7) < 0.005
    assert torch.allclose(img[kept], x0[..., :IMG][kept] / 0.7, rtol=1e-6)
    assert torch.equal(cp, x.bfloat16())
    m = vln.ops.dropout_mask(B * V * IMG, 99, 5, 0.3, dev()).view(B, V, IMG)
    assert torch.equal(m > 0, kept)                                 # exported mask == applied mask


def test_split_attention_timeout_has_its_own_sticky_word_and_switches_the_split_kernels_off(vln):
    """a timed-out exchange of the four-workgroups-per-row attention used to raise the persistent LSTM's sticky
    word -- the report blamed the recurrence and the fallback (vln_set_persistent(0)) re-issued the same split kernels.  It has
    its own word now: the check names the attention, clears `vln_get_split_attention()`, the next attention of the same shape
    runs on one workgroup per row (bit-identical to what the tunable selects) and the LSTM switch is untouched."""
    lib = vln._lib.load()
    vln._lib.check(lib.vln_persistent_check(), "clean start")
    B, S, D = 64, 36, 2176
    g = torch.Generator().manual_seed(5)
    ctx = (torch.randn(B, S, D, generator=g) * 0.3).to("cuda:0")
    q = (torch.randn(B, D, generator=g) * 0.1).to("cuda:0")
    sync = torch.zeros(int(lib.vln_attn_sync_bytes(B)) // 4 + 1, dtype=torch.int32, device="cuda:0")
    out_split, attn_split = vln.ops.attn_fwd_rows(ctx, q, None, sync=sync)
    assert lib.vln_get_split_attention() == 1
    try:
        vln._lib.check(lib.vln_debug_raise_sticky(2), "vln_debug_raise_sticky")
        rc = lib.vln_persistent_check()
        assert rc != 0
        msg = lib.vln_last_error_string().decode()
        assert "four-workgroup attention" in msg and "LSTM" not in msg, msg
        assert lib.vln_get_split_attention() == 0
        assert lib.vln_persistent_check() == 0                     # reported once
        out_one, attn_one = vln.ops.attn_fwd_rows(ctx, q, None, sync=sync)          # one workgroup per row now
        torch.cuda.synchronize()
        check(out_one, out_split, 1e-6, "weighted sum, one workgroup per row vs four")
        check(attn_one, attn_split, 1e-6, "attention weights")
    finally:
        lib.vln_set_split_attention(1)
    out2, _ = vln.ops.attn_fwd_rows(ctx, q, None, sync=sync)
    assert torch.equal(out2, out_split)


@pytest.mark.parametrize("B,S,H,n", [(64, 80, 512, 8), (9, 13, 64, 1), (17, 96, 96, 5), (64, 33, 32, 19), (3, 1, 512, 2)])
@pytest.mark.parametrize("cdt", [torch.float32, torch.bfloat16])
def test_text_attention_on_projected_context_with_the_cell_in_the_launch(vln, B, S, H, n, cdt):
    """csrc/attention_textk.h (vln_attn_textk_fwd / _bwd): the LSTM cell's pointwise stage from split-K gate slabs + the text
    attention scored on K = ctx W_in, one launch each way, against fp64 torch math of policy.py:237-241 / units.py:106-117 with
    the kernels' dropout mask: h1, c1, the attention weights, [weighted ctx | drop(h1)], and in the backward d logits, dq (the
    dY rows of d W_in), d(weighted ctx) written back, d gates and d c0 -- ragged masks, B not a multiple of 8, parts narrower
    than a slice (H / 4 = 8 .. 128 columns), 1 .. 19 slabs, three launches in a row on ONE exchange buffer."""
    g = torch.Generator().manual_seed(B * 7 + S * 3 + H + n)
    p, seed = 0.5, 0xABCD
    tol = 1e-4 if cdt == torch.float32 else 1e-2
    sync = vln.ops.attn_sync_buffer(B, dev())
    assert vln._lib.load().vln_attn_textk_ok(vln.ops._dt(torch.empty(0, dtype=cdt)), B, S, H, sync.data_ptr(), sync.numel() * 4) == 1
    for it in range(3):
        ctx = (torch.randn(B, S, H, generator=g) * 0.5).to(cdt)
        w_in = torch.randn(H, H, generator=g) / H ** 0.5
        kctx = (ctx.double() @ w_in.double()).float()                              # K = ctx W_in  (t = W_in h; logits = ctx . t)
        mask = torch.zeros(B, S, dtype=torch.bool)
        for b in range(B):
            mask[b, max(1, S - (b % S)):] = True
        gates = torch.randn(n, B, 4 * H, generator=g) / n ** 0.5
        b_ih, b_hh = torch.randn(4 * H, generator=g) * 0.1, torch.randn(4 * H, generator=g) * 0.1
        c0 = torch.randn(B, H, generator=g) * 0.5
        off = 11 + it
        md = vln.ops.dropout_mask(B * H, seed, off, p, dev()).cpu().double().view(B, H)
        # fp64 reference
        pre = (gates.double().sum(0) + b_ih.double() + b_hh.double()).requires_grad_(True)
        c0r = c0.double().requires_grad_(True)
        i_, f_, g_, o_ = pre.split(H, 1)
        c1r = torch.sigmoid(f_) * c0r + torch.sigmoid(i_) * torch.tanh(g_)
        h1r = torch.sigmoid(o_) * torch.tanh(c1r)
        hd = h1r * md
        logits = torch.einsum("bsd,bd->bs", kctx.double(), hd).masked_fill(mask, -float("inf"))
        attn = torch.softmax(logits, 1)
        attn.retain_grad(); logits.retain_grad()
        wc = torch.einsum("bs,bsd->bd", attn, ctx.double())
        tcat_r = torch.cat((wc, hd), 1)
        h1_, c1_, act, tc, tcat, alpha = vln.ops.attn_textk_fwd(ctx.to(dev()), kctx.to(dev()), mask.to(dev()).view(torch.uint8), gates.to(dev()),
                                                                b_ih.to(dev()), b_hh.to(dev()), c0.to(dev()), sync, seed, off, p)
        check(h1_, h1r.detach(), 1e-5, "h1"); check(c1_, c1r.detach(), 1e-5, "c1")
        check(alpha, attn.detach(), tol, "alpha"); check(tcat, tcat_r.detach(), tol, "tcat")
        assert (alpha * mask.to(dev())).abs().max().item() == 0.0
        # backward: a random d tcat (in m slabs) + external gradients on h1 and c1
        m = 1 + (it + n) % 4
        dtcat = torch.randn(m, B, 2 * H, generator=g) / m ** 0.5
        dh1, dc1 = torch.randn(B, H, generator=g) * 0.3, torch.randn(B, H, generator=g) * 0.3
        loss = (tcat_r * dtcat.double().sum(0)).sum() + (h1r * dh1.double()).sum() + (c1r * dc1.double()).sum()
        dpre, dc0r = torch.autograd.grad(loss, (pre, c0r), retain_graph=True)
        dlr, = torch.autograd.grad(loss, logits, retain_graph=True)
        dq, dl, dwc, dg, dc0 = vln.ops.attn_textk_bwd(ctx.to(dev()), kctx.to(dev()), alpha, dtcat.to(dev()), dh1.to(dev()), dc1.to(dev()),
                                                      act, tc, c0.to(dev()), sync, seed, off, p)
        dlr = torch.nan_to_num(dlr)                                                    # masked positions: exactly zero
        check(dl, dlr, tol, "dl"); check(dq, torch.einsum("bs,bsd->bd", dlr, ctx.double()), tol, "dq")
        check(dwc[:, :H], dtcat.double().sum(0)[:, :H], 1e-5, "dwc written back")
        check(dg, dpre, tol, "dgates"); check(dc0, dc0r, tol, "dc0")
    torch.cuda.synchronize()
    vln._lib.check(vln._lib.load().vln_persistent_check(), "vln_persistent_check")
    assert int(sync[0].item()) == 6


def test_linear_fwd_accumulates_onto_its_output(vln):
    """VLN_ACT_ACCUM: Y += X W^T on the fused tall form (M = 5120: dctx += dK W_in^T of the projected context), on the split-K +
    reduce form (M = 64) and on a narrow output, plain and split-fp32 weights."""
    g = torch.Generator().manual_seed(77)
    for M, N, K in ((5120, 512, 512), (64, 512, 2048), (130, 12, 36)):
        x = torch.randn(M, K, generator=g); w = torch.randn(N, K, generator=g) / K ** 0.5; y0 = torch.randn(M, N, generator=g)
        ref = y0.double() + x.double() @ w.double().t()
        for split in (False, True):
            y = y0.clone().to(dev())
            vln.ops.linear_fwd(x.to(dev()), w.to(dev()), act=vln.ops.ACT_ACCUM, out=y, split=split)
            check(y, ref, 1e-4, f"y ({M},{N},{K}) split={split}")


def test_attention_dctx_deferred_second_output(vln):
    """`dk`: the (dl, q) half of the rollout's context gradient in its own tensor (the projected-context form), over more steps
    than one launch stages, with and without accumulation into dctx."""
    B, S, D, T = 8, 20, 512, 35
    g = torch.Generator().manual_seed(10)
    al = torch.rand(T, B, S, generator=g).to(dev()); dl = torch.randn(T, B, S, generator=g).to(dev())
    gw = torch.randn(T, B, 2 * D, generator=g).to(dev()); q = torch.randn(T, B, 2 * D, generator=g).to(dev())
    ref_a = torch.einsum("tbs,tbd->bsd", al.double(), gw[:, :, :D].double())
    ref_k = torch.einsum("tbs,tbd->bsd", dl.double(), q[:, :, D:].double())
    for acc in (False, True):
        out = torch.full((B, S, D), 0.5, device=dev()); dk = torch.full((B, S, D), 7.0, device=dev())
        vln.ops.attn_dctx_deferred([al[t].data_ptr() for t in range(T)], [dl[t].data_ptr() for t in range(T)],
                                   [gw[t].data_ptr() for t in range(T)], 2 * D, [q[t].data_ptr() + 4 * D for t in range(T)], 2 * D, out,
                                   accumulate=acc, dk=dk)
        check(out, ref_a + (0.5 if acc else 0.0), 1e-5, "dctx half"); check(dk, ref_k, 1e-5, "dk half")


@pytest.mark.parametrize("N,K", [(1024, 2176), (96, 300), (33, 64)])
def test_input_batchnorm_gradients_from_the_first_layers_weight_gradient(vln, N, K):
    """vln_bn0_grads_from_wgrad (round 5): with y0 = gamma * xhat + beta the first Linear layer's input and x itself carrying no
    gradient, d gamma / d beta of the input BatchNorm follow from that layer's dW = dz^T y0 and db = sum_r dz -- against the direct
    definition d gamma_k = sum_r (dz W)[r,k] xhat[r,k], d beta_k = sum_r (dz W)[r,k] in fp64, on a random batch; the layer's own
    dW / db are handed on into the accumulated gradients; a gamma of exactly 0 is reported, not divided by."""
    g = torch.Generator().manual_seed(N + K)
    R = 200
    xhat = torch.randn(R, K, generator=g).double()
    gamma = (torch.rand(K, generator=g) + 0.5).double() * torch.where(torch.rand(K, generator=g) < 0.5, -1.0, 1.0).double()
    beta = torch.randn(K, generator=g).double() * 0.3
    W = (torch.randn(N, K, generator=g) / K ** 0.5).double()
    dz = torch.randn(R, N, generator=g).double()
    y0 = xhat * gamma + beta
    dW, db = dz.t() @ y0, dz.sum(0)
    gfull = dz @ W
    ref_gg, ref_gb = (gfull * xhat).sum(0), gfull.sum(0)
    f = lambda t: t.float().to(dev())
    gW0, gb0, gg0, gbeta0 = torch.randn(N, K, generator=g), torch.randn(N, generator=g), torch.randn(K, generator=g), torch.randn(K, generator=g)
    gW, gb, gg, gbeta = f(gW0.double()), f(gb0.double()), f(gg0.double()), f(gbeta0.double())
    vln.ops.bn0_grads_from_wgrad(f(dW), f(db), f(W), f(gamma), f(beta), gW, gb, gg, gbeta, accumulate=True)
    check(gg - f(gg0.double()), ref_gg, 2e-5, "d gamma of the input BatchNorm")
    check(gbeta - f(gbeta0.double()), ref_gb, 2e-5, "d beta of the input BatchNorm")
    check(gW, gW0.double() + dW, 1e-6, "the layer's weight gradient handed on")
    check(gb, gb0.double() + db, 1e-6, "the layer's bias gradient handed on")
    gg2, gbeta2 = torch.empty(K, device=dev()), torch.empty(K, device=dev())
    vln.ops.bn0_grads_from_wgrad(f(dW), f(db), f(W), f(gamma), f(beta), torch.empty(N, K, device=dev()), torch.empty(N, device=dev()), gg2, gbeta2,
                                 accumulate=False)
    check(gg2, ref_gg, 2e-5, "d gamma (stored)")
    # a BatchNorm weight of exactly 0: no quotient -- the launch says so through the sticky status line and leaves that d gamma 0
    lib = vln._lib.load()
    assert lib.vln_persistent_check() == 0
    gz = f(gamma).clone(); gz[3] = 0.0
    vln.ops.bn0_grads_from_wgrad(f(dW), f(db), f(W), gz, f(beta), torch.empty(N, K, device=dev()), torch.empty(N, device=dev()), gg2, gbeta2, accumulate=False)
    torch.cuda.synchronize()
    assert float(gg2[3]) == 0.0 and torch.isfinite(gg2).all()
    assert lib.vln_persistent_check() != 0 and b"set_bn0_grads_from_wgrad" in lib.vln_last_error_string()
    assert lib.vln_persistent_check() == 0


def test_input_batchnorm_gradient_shortcut_reports_ill_conditioned_weights(vln):
    """d gamma_k = sum_n (dW[n,k] - beta_k db[n]) / gamma_k W[n,k] cancels catastrophically when |gamma_k| << |beta_k|,
    and the quotient amplifies the weight gradient's rounding by |beta| / |gamma|.  (a) |gamma| = |beta| / 32 -- inside the bound
    VLN_BN0_MAX_AMPLIFICATION = 64 --: from an fp32 dW (the fp32 mode) the shortcut's d gamma meets 1e-4 of the tensor's scale
    (measured 2e-6); from a dW carrying the rounding of the split-bf16 form (what the bf16 mode forms a BN-MLP's deferred weight
    gradients in: hi + lo planes, three MFMAs, ~2^-16) it meets 1e-3, a tenth of the bf16 bound (2e-4); nothing is reported.  The
    same dW from PLAIN bf16 operands (what precision 2 would be -- the library never uses it behind a BatchNorm, csrc/bn_mlp.hip)
    misses by orders of magnitude: recorded.  (b) |gamma| < |beta| / 64: the launch reports it through the sticky status line
    instead of handing the degraded d gamma on silently."""
    g = torch.Generator().manual_seed(77)
    N, K, R = 256, 192, 200
    xhat = torch.randn(R, K, generator=g).double()
    beta = (torch.rand(K, generator=g) + 0.5).double()                                          # O(1)
    gamma = beta / 32 * torch.where(torch.rand(K, generator=g) < 0.5, -1.0, 1.0).double()      # |gamma| = |beta| / 32
    W = (torch.randn(N, K, generator=g) / K ** 0.5).double()
    dz = torch.randn(R, N, generator=g).double()
    y0 = xhat * gamma + beta
    dW, db = dz.t() @ y0, dz.sum(0)
    gfull = dz @ W
    ref_gg = (gfull * xhat).sum(0)
    f = lambda t: t.float().to(dev())
    lib = vln._lib.load()
    assert lib.vln_persistent_check() == 0
    # dW as the split-bf16 wgrad forms it: both operands as hi + lo bf16 planes, the lo x lo product dropped, fp32 accumulate
    hi = lambda t: t.float().bfloat16().float()
    dzf, y0f = dz.float(), y0.float()
    dzh, y0h = hi(dzf), hi(y0f)
    dzl, y0l = hi(dzf - dzh), hi(y0f - y0h)
    dW_split = (dzh.t().double() @ y0h.double() + dzh.t().double() @ y0l.double() + dzl.t().double() @ y0h.double()).float()
    for name, dW_in, tol in (("an fp32 dW (fp32 mode)", dW.float(), 1e-4),
                             ("a split-bf16 dW (bf16 mode behind a BatchNorm)", dW_split, 1e-3),
                             ("a plain-bf16 dW (never used behind a BatchNorm; recorded)", (dzh.t().double() @ y0h.double()).float(), 1e9)):
        gg, gbeta = torch.empty(K, device=dev()), torch.empty(K, device=dev())
        vln.ops.bn0_grads_from_wgrad(dW_in.to(dev()), f(db), f(W), f(gamma), f(beta), torch.empty(N, K, device=dev()), torch.empty(N, device=dev()),
                                     gg, gbeta, accumulate=False)
        torch.cuda.synchronize()
        check(gg, ref_gg, tol, f"d gamma at |gamma| = |beta| / 32 from {name}")
    assert lib.vln_persistent_check() == 0                          # amplification 32 < 64: inside the bound, nothing reported
    gs = f(gamma).clone(); gs[5] = float(beta[5]) / 4096.0           # one weight 4096x below its bias
    gg, gbeta = torch.empty(K, device=dev()), torch.empty(K, device=dev())
    vln.ops.bn0_grads_from_wgrad(f(dW), f(db), f(W), gs, f(beta), torch.empty(N, K, device=dev()), torch.empty(N, device=dev()), gg, gbeta, accumulate=False)
    torch.cuda.synchronize()
    assert torch.isfinite(gg).all()
    assert lib.vln_persistent_check() != 0 and b"set_bn0_grads_from_wgrad" in lib.vln_last_error_string()
    assert lib.vln_persistent_check() == 0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_grouped_shadow_refresh_of_named_weights_equals_the_single_launches(vln, dtype):
    """functional._ShadowCache.ensure (round 5): a decoder names the weight shadows its steps will stream and the stale ones are
    cast / transposed by ONE `vln_shadow_refresh` launch -- the [W_ih | W_hh] pair without a concatenated copy.  Same bytes as the
    per-matrix launches of `get` / `_fused_lstm_weight`; afterwards those calls hit the cache (the very same tensors), and an
    in-place update of a parameter stales exactly its shadows."""
    F_ = vln.functional
    g = torch.Generator().manual_seed(5)
    mk = lambda *sh: torch.nn.Parameter(torch.randn(*sh, generator=g).to(dev()))
    W1, W2, Wih, Whh = mk(96, 200), mk(64, 64), mk(256, 132), mk(256, 64)
    ref_cache, cache = F_._ShadowCache(), F_._ShadowCache()
    wants = [(W1, "n", dtype), (W1, "t", dtype), (W2, "t", dtype)]
    refs = [ref_cache.get(w, k, d).clone() for w, k, d in wants]
    F_._fused_cache.clear()
    ref_f = [F_._fused_lstm_weight(Wih, Whh, dtype, tr).clone() for tr in (False, True)]
    F_._fused_cache.clear()
    cache.ensure(wants, [(Wih, Whh, dtype, False), (Wih, Whh, dtype, True)])
    got = [cache.get(w, k, d) for w, k, d in wants]
    for a, b, (w, k, d) in zip(got, refs, wants):
        assert a.dtype == (dtype if not (k == "n" and dtype == torch.float32) else torch.float32) and torch.equal(a, b), (k, d)
    got_f = [F_._fused_lstm_weight(Wih, Whh, dtype, tr) for tr in (False, True)]
    for a, b in zip(got_f, ref_f):
        assert a.shape == b.shape and torch.equal(a, b)
    # hits: the same objects, no new tensors
    assert all(cache.get(w, k, d) is t for (w, k, d), t in zip(wants, got))
    assert all(F_._fused_lstm_weight(Wih, Whh, dtype, tr) is t for tr, t in zip((False, True), got_f))
    with torch.no_grad():
        W1.mul_(2.0)
    cache.ensure(wants, [])
    assert torch.equal(cache.get(W1, "t", dtype).float(), (W1.detach().t().contiguous().to(dtype)).float())
    assert cache.get(W2, "t", dtype) is got[2]                       # untouched parameter: still the old shadow
    F_._fused_cache.clear()
