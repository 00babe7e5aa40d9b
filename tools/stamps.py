#!/usr/bin/env python3
"""Diagnostic: per-wave phase clocks of k_frames from a library built with -DSP_STAMPS (tools/build_variant.sh stamps -DSP_STAMPS).
   SP_EXPERIMENT_KNOBS=1 SP_LIB_VARIANT=stamps python3 tools/stamps.py [cfg | FORMAT:log2samples:n]"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from __graft_entry__ import load_package
pkg = load_package()
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
if ":" in cfg:   # FORMAT:log2(samples):n, e.g. CU4:24:1024 (bytes per sample from the library's format table)
    fmt, lg, n = cfg.split(":")[0], int(cfg.split(":")[1]), int(cfg.split(":")[2])
    window, cmap, frames = "blackmanHarris", "viridis", None
    bench.SAMPLE_WIDTH[fmt] = pkg.parse_format(fmt)[1]
else:
    fmt, lg, n, window, cmap, frames, desc = bench.CONFIGS[cfg]
S = 1 << lg; sw = bench.SAMPLE_WIDTH[fmt]; W = frames if frames else S // n
ctx = pkg.Context(0)
win, weight = pkg.window(window, n)
lut = bench.load_cmap(cmap)
plan = ctx.plan(fmt, n, win, 1.0 / weight, 6.0, 30.0, lut)
plan.force_kernel("frames")
d_in = ctx.alloc(S * sw)
ctx.synth_trinoise(d_in, fmt, 0, S, bench.GEN["seed"], bench.GEN["step"], bench.GEN["gshift"], bench.GEN["amp"], bench.GEN["namp"])
L = len(lut)
ptrs = [ctx.alloc(max(s, 16)) for s in (4 * W * n, W, W, W, 8 * L, 8000, 16)]
# SP_STAMPS_ROTATE=K: K capture / image sets in rotation (the stamped launch - the last one - then streams through HBM instead of
# finding its capture and its image in the Infinity Cache)
ROT = int(os.environ.get("SP_STAMPS_ROTATE", "1"))
sets = [(d_in, ptrs[0])]
for k in range(1, ROT):
    di = ctx.alloc(S * sw)
    ctx.synth_trinoise(di, fmt, k * S, S, bench.GEN["seed"], bench.GEN["step"], bench.GEN["gshift"], bench.GEN["amp"], bench.GEN["namp"])
    sets.append((di, ctx.alloc(4 * W * n)))
for i in range(300):
    di, img = sets[i % ROT]
    plan.execute(di, S * sw, W, img, *ptrs[1:])
ctx.synchronize()
waves = 8 if n > 1024 else int(os.environ.get("SP_WAVES", "8"))
cnt = 256 * waves * 24
buf = (ctypes.c_ulonglong * cnt)()
lib = ctx.lib.L
lib.sp_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
assert lib.sp_debug_read_stamps(ctx.h, buf, cnt) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(256, waves, 24).astype(np.float64)
names = ["head+decode", "pass1+xch1(+drain0)", "pass2+xch2", "pass3", "drain+barrier", "epilogue", "loop total", "start clock"]
tot = a[:, :, 6].mean()
print("%s: mean cycles per wave over the frame loop (%d frames per wave): %.0f" % (cfg, W // (256 * waves), tot))
for k in range(6):
    print("  %-22s %9.0f  %5.1f %%   (min %.0f max %.0f over waves)" % (names[k], a[:, :, k].mean(), 100 * a[:, :, k].mean() / tot, a[:, :, k].min(), a[:, :, k].max()))
waves_per_frame = max(1, n // 1024)
print("  exact-edge path taken %.2f times per frame per wave (8 batches of 2 bins per frame); per-wave loop total min %.0f max %.0f"
      % (a[:, :, 7].mean() / (W * waves_per_frame / (256.0 * waves)), a[:, :, 6].min(), a[:, :, 6].max()))
print("  by wave index (mean loop total):", np.round(a[:, :, 6].mean(axis=0)))
for k in range(6):
    print("  by wave index %-20s" % names[k], np.round(a[:, :, k].mean(axis=0)))
ghz = a[:, :, 9].mean() / (a[:, :, 10].mean() * 10.0) 
print("  whole wave: %.0f shader cycles = %.2f us of the 100 MHz clock -> %.2f GHz; prologue %.0f cycles (%.2f us), loop %.0f (%.2f us), tail %.0f (%.2f us)"
      % (a[:, :, 9].mean(), a[:, :, 10].mean() / 100.0, ghz, a[:, :, 8].mean(), a[:, :, 8].mean() / ghz / 1e3, tot, tot / ghz / 1e3,
         (a[:, :, 9] - a[:, :, 8] - a[:, :, 6]).mean(), (a[:, :, 9] - a[:, :, 8] - a[:, :, 6]).mean() / ghz / 1e3))
# (n <= 1024: the first request is issued behind the table loads, so the argument segment's first fetch and the request's issue show in the second figure)
print("  prologue: arguments + first request %.0f cycles, tables arrive %.0f, zeroing + barrier %.0f" % (a[:, :, 16].mean(), a[:, :, 17].mean(), a[:, :, 18].mean()))
print("  inside the frame meetings (n >= 2048): %.0f cycles per wave = %.1f %% of the loop; by wave:" % (a[:, :, 19].mean(), 100 * a[:, :, 19].mean() / tot), np.round(a[:, :, 19].mean(axis=0)))
hw = a[:, :, 13].astype(np.int64)
print("  SIMD of wave index 0..7 (workgroup 0, 1, 100):", [list((hw[b] >> 4) & 3) for b in (0, 1, 100)], " wave slot:", list(hw[0] & 15))
simd = (hw >> 4) & 3
print("  mean loop total by SIMD:", [round(float(a[:, :, 6][simd == q].mean())) for q in range(4)])
for k, nm in ((12, "barrier after the loop"), (20, "last write-out issued (not by wave sets)"), (21, "cells -> prefix sums (two barriers)"),
              (14, "adds issued, last side outputs, barrier"), (15, "range adds, outstanding memory ops")):
    print("  tail: %-42s %7.0f cycles (%.2f us)  by wave:" % (nm, a[:, :, k].mean(), a[:, :, k].mean() / ghz / 1e3), np.round(a[:, :, k].mean(axis=0)))
st = a[:, :, 11]
print("  wave start times: spread %.2f us over the launch; last wave end - first wave start = %.2f us"
      % ((st.max() - st.min()) / 100.0, ((st + a[:, :, 10]).max() - st.min()) / 100.0))
print("  start offset by workgroup (us, first 16):", np.round((st[:16, 0] - st.min()) / 100.0, 2))
