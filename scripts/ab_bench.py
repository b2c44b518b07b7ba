"""Interleaved A/B timing of training-iteration variants in ONE process (cdna guide rule 24: separate invocations and
separate boxes differ by >10 %).  Usage: python scripts/ab_bench.py [rounds] [bf16|fp32]"""
import statistics
import sys
import time

sys.path.insert(0, ".")
import torch
import bench
import vln_amd as vln

dev = torch.device("cuda:0")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dtype = torch.float32 if (len(sys.argv) > 2 and sys.argv[2] == "fp32") else torch.bfloat16
cpu_tape = vln.synthetic.make_tape(64, 80, 7, 8, 2020)
tape_t = vln.synthetic.tape_to(cpu_tape, dev)
tape_s = vln.synthetic.tape_to(cpu_tape, dev, store_dtype=dtype)
lib = vln._lib.load()

# variant -> {tunable id: value} on top of the defaults (vln_set_tunable; see csrc/vln_internal.h)
DEFAULT_TUN = {0: 384, 1: 1, 2: 1, 3: 512, 4: 0, 5: 0, 6: 0, 7: 0}
VARIANTS = {
    "base": {},
    "gemm_split_target256": {0: 256},
    "gemm_split_target384": {0: 384},
    "gemm_split_target448": {0: 448},
    "gemm_split_target512": {0: 512},
    "lstm_dispatch_order_map": {7: 1},
    "attn_two_kernels": {4: 1},
    "wgrad_fp32_exact": {6: 1},
    "overlap_wgrads": {},
    "per_step_logits_and_loss_branch": {},
}
torch.manual_seed(0)
agent = vln.trainers.EnvDropILIteration(dev, dtype, 1)


def configure(cfg, name=""):
    agent.dec.overlap_wgrads = (name == "overlap_wgrads")
    agent.dec.defer_logits = agent.dec.batch_logit_backward = (name != "per_step_logits_and_loss_branch")
    want = not name.startswith("noarena")
    if want != (agent.arena is not None):
        agent.use_arena(want)
    lib.vln_set_persistent(0 if name == "per_step_lstm" else 1)
    lib.vln_set_graphs(0 if name == "no_graphs" else 1)
    tun = dict(DEFAULT_TUN); tun.update(cfg)
    for k, v in tun.items():
        lib.vln_set_tunable(k, v)


times = {n: [] for n in VARIANTS}
for n, cfg in VARIANTS.items():
    configure(cfg, n)
    for _ in range(3):
        agent.iteration(tape_t if n == 'tensor_features' else tape_s)
torch.cuda.synchronize()
for r in range(rounds):
    for n, cfg in VARIANTS.items():
        configure(cfg, n)
        agent.iteration(tape_t if n == 'tensor_features' else tape_s)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            agent.iteration(tape_t if n == 'tensor_features' else tape_s)
        torch.cuda.synchronize()
        times[n].append((time.perf_counter() - t0) / 10 * 1e3)
configure(VARIANTS["base"], "base")
for n, v in times.items():
    print(f"{n:22s} median {statistics.median(v):.3f} ms  min {min(v):.3f}  max {max(v):.3f}")
import ctypes
st = (ctypes.c_int64 * 3)()
lib.vln_graph_stats(st)
print(f"graphs: {st[0]} replays, {st[1]} captures, {st[2]} chains switched off")
