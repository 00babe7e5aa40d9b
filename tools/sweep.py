#!/usr/bin/env python3
"""Throughput over formats x FFT sizes on one GPU (device-resident operands, frame loop + finish kernel), to spot variants that fall off:
   ps per butterfly = time / (frames * n/2 * log2 n).      python3 tools/sweep.py [log2 samples] [FMT,FMT,...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from __graft_entry__ import load_package
pkg = load_package()
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 24
S = 1 << lg
WIDTH = {"CU4": 1, "CU8": 2, "CS8": 2, "CU12": 3, "CS12": 3, "CU16": 4, "CS16": 4, "CF32": 8, "CS32": 8, "CF64": 16, "CS64": 16}
if len(sys.argv) > 2:
    WIDTH = {f: WIDTH[f] for f in sys.argv[2].split(",")}
CH = os.environ.get("SWEEP_CH") == "1"          # L/R channel split (fft_nayuki.js:103-119) instead of I/Q
LO = int(os.environ.get("SWEEP_LOG2N_MIN", "6"))
ctx = pkg.Context(0)
lut = bench.load_cmap("viridis")
L = len(lut)
print("%-5s" % "n" + "".join("%14s" % f for f in WIDTH))
for ln in range(LO, 14):
    n = 1 << ln
    row = "%-5d" % n
    for fmt, sw in WIDTH.items():
        W = S // n
        win, weight = pkg.window("hann", n)
        plan = ctx.plan(fmt, n, win, 1.0 / weight, 6.0, 30.0, lut, CH)
        d_in = torch.empty(S * sw, dtype=torch.uint8, device="cuda")
        ctx.synth_trinoise(d_in.data_ptr(), fmt, 0, S, bench.GEN["seed"], bench.GEN["step"], bench.GEN["gshift"], bench.GEN["amp"], bench.GEN["namp"])
        rgba = torch.empty(4 * W * n, dtype=torch.uint8, device="cuda")
        g = torch.empty(3 * W, dtype=torch.uint8, device="cuda")
        rec = torch.zeros(L + 1002, dtype=torch.int64, device="cuda")
        p = rec.data_ptr()
        def run():
            plan.execute(d_in.data_ptr(), S * sw, W, rgba.data_ptr(), g.data_ptr(), g.data_ptr() + W, g.data_ptr() + 2 * W, p, p + 8 * L, p + 8 * (L + 1000))
        for _ in range(30):
            run()
        torch.cuda.synchronize()
        reps = 60
        t0 = time.perf_counter()
        for _ in range(reps):
            run()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        ps = dt / (W * (n / 2) * ln) * 1e12
        row += "%8.0f us %4.2f" % (dt * 1e6, ps)
        plan.close()
        del d_in, rgba
    print(row, flush=True)
print("(us per launch of 2^%d samples, ps per butterfly; kernel: %s)" % (lg, "frames unless the plan falls back"))
