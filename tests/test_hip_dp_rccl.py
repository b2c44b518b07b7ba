"""GPU: the data-parallel exchange on the real backend.  A one-GPU box cannot run two RCCL ranks (one device per rank), so this
runs the product's collective path -- `dp.BucketReducer` through `optim.FusedRMSprop.start_allreduce / allreduce`,
`dp.allreduce_scalar`, `dp.gather_item_losses` -- on a ONE-rank `nccl` (= RCCL) group in a child process: library load,
communicator creation with `device_id`, asynchronous work objects and their stream ordering, with the world-size gate forced
open.  The two-rank numerics are covered over gloo (tests/test_dp_gloo.py)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys
sys.path.insert(0, os.environ["VLN_ROOT"])
import torch, torch.distributed as dist
import vln_amd as vln
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == "nccl"
vln.dp._dp_active = lambda group=None: True          # a one-rank group: run the collectives anyway
torch.manual_seed(0)
enc = torch.nn.Linear(64, 32).to(dev); dec = torch.nn.Linear(32, 16).to(dev)
opt = vln.optim.FusedRMSprop([list(enc.parameters()), list(dec.parameters())], lr=1e-3, clip_norm=[40.0, 40.0])
x = torch.randn(8, 64, device=dev)
opt.zero_grad()
dec(enc(x)).pow(2).sum().backward()
ref = [p.grad.clone() for p in list(enc.parameters()) + list(dec.parameters())]
opt.start_allreduce(1)                                 # the decoder's slice goes out early, asynchronously
assert len(opt._reducer.pending) == 1
opt.allreduce()                                        # the rest + wait
assert not opt._reducer.pending
torch.cuda.synchronize()
for p, r in zip(list(enc.parameters()) + list(dec.parameters()), ref):
    assert torch.equal(p.grad, r)                      # sum over one rank = identity
before = [p.detach().clone() for p in enc.parameters()]
opt.step()
torch.cuda.synchronize()
assert any(not torch.equal(a, b) for a, b in zip(before, enc.parameters()))
t = vln.dp.allreduce_scalar(torch.tensor([5.0], device=dev))
assert float(t) == 5.0
gi, gl = vln.dp.gather_item_losses(torch.arange(4, device=dev), torch.arange(4, device=dev).float() * 2)
assert gi.tolist() == [0, 1, 2, 3] and gl.tolist() == [0.0, 2.0, 4.0, 6.0]
dist.destroy_process_group()
print("RCCL-PATH-OK")
'''


CHILD_GRAPH = r'''
import os, sys
sys.path.insert(0, os.environ["VLN_ROOT"])
import torch, torch.distributed as dist
import vln_amd as vln, bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
vln.dp._dp_active = lambda group=None: True          # a one-rank group: run the collectives anyway
dtype = torch.bfloat16

def run(form):
    torch.manual_seed(77)
    store = vln.synthetic.build_store(dev, dtype, n_rows=300, seed=5)
    tapes = [vln.synthetic.tape_to(vln.synthetic.make_tape(16, 24, 4, 6, seed=500 + k, n_rows=store.N), dev, store=store) for k in range(5)]
    live = vln.LiveBatch(tapes, source="pull")
    torch.manual_seed(78)
    ag = vln.trainers.EnvDropILIteration(dev, dtype, 1, arena=True)
    ag.use_live(live)
    ag.clear_grads_in_step = True
    ag.enc.deterministic_embedding_grad = True
    ag.ride_gather = True
    ag.dec.ride_wgrads = False                       # a data-parallel rank wants the decoder's gradients final before the BPTT
    ag.use_clock(store)
    if form == "segments":
        ag.segmented = True
    elif form == "captured":                          # ONE graph, the process group's collectives are nodes of it
        ag.dec.grads_ready_hook = lambda: ag.opt.start_allreduce(1)
        ag.capture_error_mode = "thread_local"
    out = []
    def rec(loss):
        torch.cuda.synchronize()
        out.append((loss.detach().clone(), ag.opt.flat_p.clone(), ag.opt.sq.clone()))
    for k in range(2):
        rec(ag.iteration(live.load(k)))
    ag.capture(live.live)
    for k in range(2, 6):
        live.load(k)
        rec(ag.replay())
    vln._lib.check(vln._lib.load().vln_persistent_check(), "vln_persistent_check")
    return out

single = run("single")
for form in ("segments", "captured"):
    got = run(form)
    for i, (a, b) in enumerate(zip(single, got)):
        for x, y, what in zip(a, b, ("loss", "parameters", "RMSprop state")):
            assert torch.equal(x, y), f"{form}: iteration {i}: {what} differ from the single graph"
dist.destroy_process_group()
print("RCCL-GRAPH-OK")
'''


def _run_child(code):
    for attempt in range(2):            # the rendezvous port is picked by bind-and-close: one retry if it was taken meanwhile
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), VLN_ROOT=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_DEBUG="WARN")
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
        if r.returncode == 0 or "address already in use" not in r.stderr.lower():
            break
    if r.returncode != 0:               # keep the child's whole output where the caller of the suite can read it (pytest's repr truncates)
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "dp_rccl_child_failure.txt"), "a") as f:
            f.write(f"==== return code {r.returncode}\n---- stdout\n{r.stdout}\n---- stderr\n{r.stderr}\n")
    return r


def test_gradient_exchange_captured_inside_the_iteration_graph():
    """Round 5: the N > 1 iteration as ONE hipGraph -- the flat bucket's early decoder slice and the rest go
    through `torch.distributed.all_reduce` on a one-rank RCCL group INSIDE the capture (the collectives become nodes of the graph,
    the early slice on the process group's stream beside the encoder's BPTT) -- against round 4's three graph segments with
    host-issued collectives and against the plain single graph: losses, parameters and RMSprop state over six iterations (two
    eager, four replays) bit for bit."""
    r = _run_child(CHILD_GRAPH)
    assert r.returncode == 0 and "RCCL-GRAPH-OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


def test_collective_path_on_a_one_rank_rccl_group():
    for attempt in range(2):            # the rendezvous port is picked by bind-and-close: one retry if it was taken meanwhile
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), VLN_ROOT=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
        r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=300)
        if r.returncode == 0 or "address already in use" not in r.stderr.lower():
            break
    assert r.returncode == 0 and "RCCL-PATH-OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


@pytest.mark.parametrize("wgs", [32, 64, 160])
def test_persistent_recurrence_beside_a_resident_kernel(wgs):
    """The data-parallel design starts the decoder slice's all-reduce right before the encoder's BPTT, so a communication kernel
    is RESIDENT while the persistent recurrence -- whose workgroups spin on each other -- is launched.  Stand-in: a kernel on
    a second stream that holds `wgs` compute units (1024 threads + 160 KB of LDS each: nothing else fits beside it) for 3 ms.
    32 / 64 held CUs leave room for the recurrence's 128 workgroups; 160 do not: part of its grid can only start when the other
    kernel leaves.  In every case the instruction encoder's forward + backward must equal the undisturbed run bit for bit and no
    bounded wait may time out (a communication kernel finishes on its own: RCCL kernels of the ranks wait for each other, never
    for the recurrence)."""
    import time
    import torch
    import vln_amd as vln
    lib = vln._lib.load()
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    B, L = 64, 80
    enc = vln.EncoderLSTM(992, 256, 512, 0, 0.5, True, 1, compute_dtype=torch.bfloat16).to(dev).train()
    enc.deterministic_embedding_grad = True
    g = torch.Generator().manual_seed(6)
    lens = torch.sort(torch.randint(8, L + 1, (B,), generator=g), descending=True).values; lens[0] = L
    tokens = torch.zeros(B, L, dtype=torch.long)
    for i, n in enumerate(lens.tolist()):
        tokens[i, :n] = torch.randint(4, 992, (n,), generator=g)
    tokens = tokens.to(dev)
    r = torch.randn(B, L, 512, generator=g).to(dev)
    side = torch.cuda.Stream()

    def run(occupy):
        enc._calls = 0
        for p in enc.parameters():
            p.grad = None
        torch.cuda.synchronize()
        if occupy:
            with torch.cuda.stream(side):
                vln._lib.check(lib.vln_debug_occupy(wgs, 160 * 1024, 3000, side.cuda_stream), "vln_debug_occupy")
        t0 = time.perf_counter()
        ctx, h, c = enc(tokens, lens)
        ((ctx * r).sum() + h.sum() + c.sum()).backward()
        torch.cuda.current_stream().synchronize()
        dt = (time.perf_counter() - t0) * 1e3
        torch.cuda.synchronize()
        vln._lib.check(lib.vln_persistent_check(), "vln_persistent_check")
        return ctx.detach().clone(), [p.grad.clone() for p in enc.parameters()], dt

    run(False)
    ref_ctx, ref_g, t_alone = run(False)
    ctx, grads, t_beside = run(True)
    assert enc.persistent_status() == 0
    assert torch.equal(ctx, ref_ctx)
    for a, b in zip(grads, ref_g):
        assert torch.equal(a, b)
    print(f"\n[recurrence beside a resident kernel on {wgs} CUs] encoder fwd+bwd {t_alone:.3f} ms alone, {t_beside:.3f} ms beside it")
