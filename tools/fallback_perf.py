#!/usr/bin/env python3
"""What the portable kernel (k_scratch_radix2) costs: the requests the frame loop does not take - n > 8192, n < 64, colour maps longer
than 256 entries, index edges outside the f32 range, non-finite tapers - next to the frame loop on a request both can run.
Device-resident operands, frame loop + finish kernel, us per launch and ps per butterfly.      python3 tools/fallback_perf.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from __graft_entry__ import load_package
pkg = load_package()
ctx = pkg.Context(0)


def run(label, fmt, sw, lg, n, lut, force=None, gain=6.0, rng=30.0, taper=None):
    S = 1 << lg
    W = S // n
    win, weight = pkg.window("hann", n)
    if taper is not None:
        win = taper(win)
    plan = ctx.plan(fmt, n, win, 1.0 / weight, gain, rng, lut)
    if force:
        plan.force_kernel(force)
    L = len(lut)
    d_in = torch.empty(S * sw, dtype=torch.uint8, device="cuda")
    ctx.synth_trinoise(d_in.data_ptr(), fmt, 0, S, bench.GEN["seed"], bench.GEN["step"], bench.GEN["gshift"], bench.GEN["amp"], bench.GEN["namp"])
    rgba = torch.empty(4 * W * n, dtype=torch.uint8, device="cuda")
    g = torch.empty(3 * W, dtype=torch.uint8, device="cuda")
    rec = torch.zeros(L + 1002, dtype=torch.int64, device="cuda")
    p = rec.data_ptr()
    ex = lambda: plan.execute(d_in.data_ptr(), S * sw, W, rgba.data_ptr(), g.data_ptr(), g.data_ptr() + W, g.data_ptr() + 2 * W, p, p + 8 * L, p + 8 * (L + 1000))  # noqa: E731
    for _ in range(5):
        ex()
    torch.cuda.synchronize()
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        ex()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    lv = int(np.log2(n))
    print("%-58s %-15s %9.0f us  %6.2f ps/butterfly  %7.1f M frames/s" % (label, plan.kernel_name(), dt * 1e6, dt / (W * (n / 2) * lv) * 1e12, W / dt / 1e6), flush=True)
    plan.close()


viridis = bench.load_cmap("viridis")
i = np.arange(1024)
long_lut = np.stack([i & 255, (i >> 2) & 255, 255 - (i & 255)], axis=1).astype(np.uint8)
print("2^24 samples per launch (2^22 for n >= 65536)")
run("cf32 n=1024 viridis (frame loop, for scale)", "CF32", 8, 24, 1024, viridis)
run("cf32 n=1024 viridis, portable kernel forced", "CF32", 8, 24, 1024, viridis, force="scratch")
run("cf32 n=1024, 1024-entry colour map", "CF32", 8, 24, 1024, long_lut)
run("cf32 n=1024, gain 2500 (edges outside the f32 range)", "CF32", 8, 24, 1024, viridis, gain=2500.0)
run("cf32 n=1024, taper with an infinity", "CF32", 8, 24, 1024, viridis, taper=lambda w: np.concatenate([[np.inf], w[1:]]))
run("cu8  n=32 viridis", "CU8", 2, 24, 32, viridis)
run("cf32 n=8192 viridis (frame loop, for scale)", "CF32", 8, 24, 8192, viridis)
run("cf32 n=16384 viridis", "CF32", 8, 24, 16384, viridis)
run("cf32 n=65536 viridis", "CF32", 8, 22, 65536, viridis)
run("cf32 n=1048576 viridis", "CF32", 8, 22, 1 << 20, viridis)
