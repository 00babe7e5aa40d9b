#!/usr/bin/env python3
"""Diagnostic: share of a config's pixels in the end bins of the two histograms (clipped colour values / centi-bel levels).
   python3 tools/hist_ends.py [cfg ...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import torch
from __graft_entry__ import load_package
pkg = load_package()
for cfg in (sys.argv[1:] or ["cfg1", "cfg2", "cfg3", "cfg4", "cfg5"]):
    fmt, lg, n, window, cmap, frames, desc = bench.CONFIGS[cfg]
    S = 1 << lg; sw = bench.SAMPLE_WIDTH[fmt]; W = frames if frames else S // n
    ctx = pkg.Context(0)
    win, weight = pkg.window(window, n)
    lut = bench.load_cmap(cmap)
    plan = ctx.plan(fmt, n, win, 1.0 / weight, 6.0, 30.0, lut)
    d_in = torch.empty(S * sw, dtype=torch.uint8, device="cuda")
    ctx.synth_trinoise(d_in.data_ptr(), fmt, 0, S, bench.GEN["seed"], bench.GEN["step"], bench.GEN["gshift"], bench.GEN["amp"], bench.GEN["namp"])
    L = len(lut)
    rgba = torch.empty(4 * W * n, dtype=torch.uint8, device="cuda")
    g = torch.empty(3 * W, dtype=torch.uint8, device="cuda")
    rec = torch.zeros(L + 1002, dtype=torch.int64, device="cuda")
    p = rec.data_ptr()
    plan.execute(d_in.data_ptr(), S * sw, W, rgba.data_ptr(), g.data_ptr(), g.data_ptr() + W, g.data_ptr() + 2 * W, p, p + 8 * L, p + 8 * (L + 1000))
    torch.cuda.synchronize()
    r = rec.cpu().numpy()
    c, cb = r[:L], r[L:L + 1000]
    tot = float(W) * n
    mm = np.frombuffer(r[L + 1000:L + 1002].tobytes(), dtype=np.float64)
    print("%s: colour 0: %.3f  colour %d: %.3f | cB bin 0: %.4f  bin 999: %.4f | dBfs range %s" % (cfg, c[0] / tot, L - 1, c[-1] / tot, cb[0] / tot, cb[999] / tot, mm))
    plan.close(); del ctx
