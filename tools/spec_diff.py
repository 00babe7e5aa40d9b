"""Prototype check: renders bench config slices with the library selected by SP_LIB_VARIANT and writes the outputs to an .npz;
with two .npz files, prints how many pixels / counts differ.
    python3 tools/spec_diff.py dump out.npz [cfg] [log2 samples]   |   python3 tools/spec_diff.py cmp a.npz b.npz"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def dump(path, cfg="cfg2", lg=None):
    import bench
    import siggen
    from __graft_entry__ import load_package
    pkg = load_package()
    fmt, lg0, n, window, cmap, frames, _ = bench.CONFIGS[cfg]
    lg = int(lg) if lg else min(lg0, 22)
    S = 1 << lg
    W = S // n if not frames else (frames >> (lg0 - lg))
    gen = dict(kind="trinoise", **bench.GEN)
    data = siggen.generate(fmt, gen, S)
    win, weight = pkg.window(window, n)
    lut = bench.load_cmap(cmap)
    ctx = pkg.Context(0)
    got = ctx.render(fmt, data, n, win, 1.0 / weight, 6.0, 30.0, lut, W)
    np.savez(path, **got)
    ctx.close()


def cmp(a, b):
    A, B = np.load(a), np.load(b)
    px = (A["rgba"].view(np.uint32) != B["rgba"].view(np.uint32)).sum()
    print("pixels: %d of %d differ (%.3g)" % (px, A["rgba"].size // 4, px / (A["rgba"].size // 4)))
    for k in ("c_hist", "cB_hist", "gauge_mins", "gauge_maxs", "gauge_amps"):
        print(k, "differing entries:", int((A[k] != B[k]).sum()), "abs sum", int(np.abs(A[k].astype(np.int64) - B[k].astype(np.int64)).sum()))
    for k in ("dBfs_min", "dBfs_max"):
        print(k, float(A[k]), float(B[k]), "rel", abs(float(A[k]) - float(B[k])) / max(abs(float(A[k])), 1e-300))


if __name__ == "__main__":
    if sys.argv[1] == "dump":
        dump(*sys.argv[2:])
    else:
        cmp(sys.argv[2], sys.argv[3])
