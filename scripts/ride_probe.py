"""How long the feature gather rides: the headline's instruction-encoder launch (B = 64, L = 80, bf16) timed with HIP events
  alone            the persistent recurrence, no passengers
  with the ride    + the 7 x 2944 rows of the rollout's feature gather as passenger workgroups (the headline's launch)
  ride, L = 2      the same ride in a two-token recurrence: the passengers on their own
over tapes that rotate through the full-size resident table (rows come from HBM, as in bench.py).
  python scripts/ride_probe.py [--lib PATH]   (--lib: copy that build over the in-tree library first; A/B of gather_ride.h)"""
import argparse
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=None)
    ap.add_argument("--reps", type=int, default=40)
    ap.add_argument("--tunable", action="append", default=[], help="I=V: vln_set_tunable(I, V)")
    args = ap.parse_args()
    if args.lib:
        shutil.copy(args.lib, os.path.join(ROOT, "curriculum-learning-for-vln_amd", "libvln_hip.so"))
    import torch
    import bench
    import vln_amd as vln
    dev = torch.device("cuda:0")
    for tv in args.tunable:
        i, v = tv.split("=")
        vln._lib.check(vln._lib.load().vln_set_tunable(int(i), int(v)), "vln_set_tunable")
    dtype = torch.bfloat16
    store = vln.synthetic.build_store(dev, dtype, 10567)
    tapes = [vln.synthetic.tape_to(vln.synthetic.make_tape(64, 80, 7, 8, seed=2020 + k, n_rows=store.N), dev, store=store) for k in range(8)]
    enc = vln.EncoderLSTM(992, 256, 512, 0, 0.5, True, 1, compute_dtype=dtype).to(dev).train()

    def run(label, use_ride, L):
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.reps)]
        for i in range(args.reps + 4):
            t = tapes[i % len(tapes)]
            steps = [(s["rows"], s["vidx"], s["crow"], s["cview"], s["chead"], s["celev"]) for s in t["steps"]]
            ride = store.rollout_ride(steps, 0.3, want_bf16=True, want_f32=False) if use_ride else None
            tok, ln = t["tokens"][:, :L].contiguous(), t["lengths32"].clamp(max=L)
            torch.cuda.synchronize()
            if i >= 4:
                ev[i - 4][0].record()
            enc(tok, ln, ride=ride) if use_ride else enc(tok, ln)
            if i >= 4:
                ev[i - 4][1].record()
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(b) for a, b in ev)
        print(f"{label:16s} median {ms[len(ms) // 2] * 1e3:7.1f} us   min {ms[0] * 1e3:7.1f} us   (whole EncoderLSTM.forward: embed + input GEMM + recurrence launch)")

    run("alone", False, 80)
    run("with the ride", True, 80)
    run("ride, L = 2", True, 2)
    run("alone, L = 2", False, 2)
    vln._lib.check(vln._lib.load().vln_persistent_check(), "vln_persistent_check")


if __name__ == "__main__":
    main()
