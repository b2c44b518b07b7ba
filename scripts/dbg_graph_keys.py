"""Which fields of the decoder-step argument block change from iteration to iteration (hipGraph key stability)."""
import ctypes as C, sys
sys.path.insert(0, '.')
import torch, bench
import vln_amd as vln
from vln_amd import _lib
dev = torch.device('cuda:0')
agent = bench.GpuAgent(vln, dev, torch.bfloat16, 1)
tape = bench.tape_to(bench.make_tape(64, 80, 7, 8, 2020), dev, store_dtype=torch.bfloat16)
lib = _lib.load()
log = []
for name in ("vln_envdrop_step_fwd", "vln_envdrop_step_bwd"):
    f = getattr(lib, name)
    def g(*a, f=f, name=name):
        rec = {}
        for i, x in enumerate(a[:-1]):
            o = x._obj
            for fld, _ in o._fields_:
                rec[f"{i}.{fld}"] = getattr(o, fld)
        log.append((name, rec))
        return f(*a)
    setattr(lib, name, g)
its = []
for it in range(6):
    log.clear()
    agent.iteration(tape)
    torch.cuda.synchronize()
    its.append(list(log))
for a, b in ((3, 4), (4, 5)):
    print(f"--- iteration {a} vs {b}")
    for (n1, r1), (n2, r2) in zip(its[a], its[b]):
        diff = [k for k in r1 if r1[k] != r2[k] and not k.endswith(".offset")]
        if diff:
            print(n1, diff)
st = (C.c_int64 * 3)(); lib.vln_graph_stats(st); print("graph stats", list(st))
