#!/usr/bin/env python3
"""Frame-loop kernel time per sample format and FFT size (HIP event pairs around single launches, inputs resident in HBM):
   python3 tools/format_sweep.py [log2 samples=24] [n,n,n,n]      - a quick look for a format or size that falls out of line."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from __graft_entry__ import load_package
pkg = load_package()
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 24
SIZES = tuple(int(x) for x in sys.argv[2].split(",")) if len(sys.argv) > 2 else (256, 1024, 2048, 8192)
S = 1 << lg
ctx = pkg.Context(0)
lut = bench.load_cmap("viridis")
L = len(lut)
WIDTH = {"CU4": 1, "CS4": 1, "CU8": 2, "CS8": 2, "CU12": 3, "CS12": 3, "CU16": 4, "CS16": 4, "CU32": 8, "CS32": 8, "CF32": 8, "CU64": 16, "CS64": 16, "CF64": 16}
# clock spin-up before the first measurement (an idle MI355X runs its first few thousand launches 5-15 % slower: bench.py does the same)
_w, _ww = pkg.window("hann", 1024)
_p = ctx.plan("CU8", 1024, _w, 1.0 / _ww, 6.0, 30.0, lut)
_d = ctx.alloc(S * 2)
_o = [ctx.alloc(max(s, 16)) for s in (4 * S, S // 1024, S // 1024, S // 1024, 8 * L, 8000, 16)]
for _ in range(4000):
    _p.execute(_d, S * 2, S // 1024, *_o)
ctx.synchronize()
print("us per launch over 2^%d samples (event pair, incl. ~6 us of dispatch latency); layout spectrogram, Blackman-Harris" % lg)
print("%-6s" % "fmt" + "".join("%10s" % ("n=%d" % n) for n in SIZES) + "     channel mode: n=1024, n=2048")
for fmt, sw in WIDTH.items():
    d_in = ctx.alloc(S * sw)
    ctx.synth_trinoise(d_in, fmt, 0, S, bench.GEN["seed"], bench.GEN["step"], bench.GEN["gshift"], bench.GEN["amp"], bench.GEN["namp"])
    row = []
    for (n, ch) in tuple((n, False) for n in SIZES) + ((1024, True), (2048, True)):
        W = S // n
        win, weight = pkg.window("blackmanHarris", n)
        plan = ctx.plan(fmt, n, win, 1.0 / weight, 6.0, 30.0, lut, channel_mode=ch)
        ptrs = [ctx.alloc(max(s, 16)) for s in (4 * W * n, W, W, W, 8 * L, 8000, 16)]
        for _ in range(20):
            plan.execute(d_in, S * sw, W, *ptrs)
        ctx.synchronize()
        ctx.enable_timing(True)
        ts = []
        for _ in range(10):
            for _ in range(4):
                plan.execute(d_in, S * sw, W, *ptrs)
            ctx.synchronize()
            ts.append(ctx.last_kernel_ms())
        ctx.enable_timing(False)
        row.append(1e3 * float(np.median(ts)))
        for p in ptrs:
            ctx.free(p)
        del plan
    ctx.free(d_in)
    print("%-6s" % fmt + "".join("%10.1f" % t for t in row[:len(SIZES)]) + "     " + "".join("%10.1f" % t for t in row[len(SIZES):]))
