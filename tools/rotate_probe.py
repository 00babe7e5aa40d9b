#!/usr/bin/env python3
"""Which side of config 2 leans on the Infinity Cache: the capture's reads or the image's stores?  The same launch over K capture buffers
and / or K image buffers in rotation (K = 1: the one-set loop), 2000 back-to-back launches each, wall time per launch.
   python3 tools/rotate_probe.py [K]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from __graft_entry__ import load_package
pkg = load_package()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 3
fmt, lg, n, window, cmap, frames, desc = bench.CONFIGS["cfg2"]
S = 1 << lg; sw = bench.SAMPLE_WIDTH[fmt]; W = S // n
ctx = pkg.Context(0)
win, weight = pkg.window(window, n)
lut = bench.load_cmap(cmap)
plan = ctx.plan(fmt, n, win, 1.0 / weight, 6.0, 30.0, lut)
ins = [torch.empty(S * sw, dtype=torch.uint8, device="cuda") for _ in range(K)]
outs = [torch.empty(4 * W * n, dtype=torch.uint8, device="cuda") for _ in range(K)]
for k in range(K):
    ctx.synth_trinoise(ins[k].data_ptr(), fmt, k * S, S, bench.GEN["seed"], bench.GEN["step"], bench.GEN["gshift"], bench.GEN["amp"], bench.GEN["namp"])
g = torch.empty(3 * W, dtype=torch.uint8, device="cuda")
rec = torch.zeros(len(lut) + 1002, dtype=torch.int64, device="cuda")
p, L = rec.data_ptr(), len(lut)
def run(i, rin, rout):
    a, b = ins[i % K if rin else 0], outs[i % K if rout else 0]
    plan.execute(a.data_ptr(), S * sw, W, b.data_ptr(), g.data_ptr(), g.data_ptr() + W, g.data_ptr() + 2 * W, p, p + 8 * L, p + 8 * (L + 1000))
for i in range(3000):
    run(i, True, True)
ctx.synchronize()
for rep in range(2):
    for name, rin, rout in (("one set", False, False), ("captures rotate", True, False), ("images rotate", False, True), ("both rotate", True, True)):
        for i in range(200):
            run(i, rin, rout)
        ctx.synchronize()
        t0 = time.perf_counter()
        for i in range(2000):
            run(i, rin, rout)
        ctx.synchronize()
        print("%-16s %.2f us per launch" % (name, (time.perf_counter() - t0) / 2000 * 1e6))
