"""What a dependent trivial kernel costs INSIDE a PyTorch process (companion of scripts/boundary_probe.hip, which measures
1.9-2.1 us per launch in a stand-alone HIP program): 48 dependent launches of the library's trivial kernel
(vln_debug_trivial_chain) captured as one graph by torch, replayed; and the same with tiny torch ops.
    python scripts/torch_boundary_probe.py [--no-lib]"""
import sys, time
sys.path.insert(0, ".")
import torch

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
use_lib = "--no-lib" not in sys.argv


def timed_graph(fn, reps=300, n=48):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (reps * n) * 1e6


x = torch.zeros(64 * 512, device=dev)


def torch_chain():
    for _ in range(48):
        x.add_(1.0)


print(f"torch ops  (x.add_ on 32768 floats, graph of 48): {timed_graph(torch_chain):.2f} us per launch", flush=True)
if use_lib:
    import vln_amd as vln
    lib = vln._lib.load()
    buf = torch.zeros(2, 64 * 512, device=dev)

    def lib_chain():
        vln._lib.check(lib.vln_debug_trivial_chain(buf[0].data_ptr(), buf[1].data_ptr(), 64 * 512, 48, 256, vln._lib.raw_stream()), "chain")

    print(f"library    (vln_debug_trivial_chain, graph of 48): {timed_graph(lib_chain):.2f} us per launch", flush=True)
    # after the library has created its host-mapped status word (first persistent launch does): again
    enc = vln.EncoderLSTM(992, 256, 512, 0, 0.5, True, 1, compute_dtype=torch.bfloat16).to(dev).train()
    tok = torch.randint(4, 992, (64, 80), device=dev)
    enc(tok, torch.full((64,), 80))
    torch.cuda.synchronize()
    print(f"library    after a persistent launch (host-mapped word exists): {timed_graph(lib_chain):.2f} us per launch", flush=True)
    print(f"torch ops  after a persistent launch: {timed_graph(torch_chain):.2f} us per launch", flush=True)
    big = torch.empty(1 << 29, device=dev)      # 2 GiB resident
    print(f"library    with 2 GiB more allocated: {timed_graph(lib_chain):.2f} us per launch", flush=True)
    pin = torch.empty(1 << 20, pin_memory=True)
    print(f"library    with pinned host memory allocated: {timed_graph(lib_chain):.2f} us per launch", flush=True)
