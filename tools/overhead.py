"""Kernel-time vs. frame count for the frame-loop kernels (HIP events), to separate fixed launch cost from per-frame cost."""
import sys, os, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package
pkg = load_package()
ctx = pkg.Context(0)
n = 1024
fmt = sys.argv[1] if len(sys.argv) > 1 else "CF32"
kern = sys.argv[2] if len(sys.argv) > 2 else "auto"
sw = pkg.parse_format(fmt)[1]
win, weight = pkg.window("blackmanHarris", n)
i = np.arange(256)
lut = np.stack([i, 255 - i, (i * 7) & 255], axis=1).astype(np.uint8)
plan = ctx.plan(fmt, n, win, 1.0 / weight, 6.0, 30.0, lut, bool(int(os.environ.get("SP_CH", "0"))))
plan.force_kernel(kern)
Smax = 24576 * n
d_in = ctx.alloc(Smax * sw)
ctx.synth_trinoise(d_in, fmt, 0, Smax, 0x5EED0001, 7321, 11, 0.5, 0.02)
sizes = [4 * 24576 * n, 24576, 24576, 24576, 8 * 256, 8000, 16]
ptrs = [ctx.alloc(s) for s in sizes]
ctx.enable_timing(True)
Ws = [int(w) for w in os.environ["SP_WIDTHS"].split(",")] if os.environ.get("SP_WIDTHS") else [12, 3072, 12288, 16384, 16416, 16512, 17408, 24576]
for W in Ws:
    ms = []
    for r in range(12):
        plan.execute(d_in, W * n * sw, W, *ptrs)
        ctx.synchronize()
        ms.append(ctx.last_kernel_ms())
    print(plan.kernel_name(), fmt, "W=%6d  kernel_us=%8.1f  (min %.1f)" % (W, 1e3 * float(np.median(ms[2:])), 1e3 * min(ms)))
