"""Soak run (not collected by pytest; run by hand on the GPU box): many more seeded random requests than the suite holds, plus
long large-n requests (n = 2048 .. 8192, up to 700 frames: many groups per workgroup, partial last groups), every output bit-exact
against the C oracle.
    python3 tests/soak_gpu.py [cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import siggen
from __graft_entry__ import load_package
from oracle import pyoracle
from test_gpu_parity import _random_cases


def big_cases(count, seed):
    rs = np.random.RandomState(seed)
    fmts = ["CU8", "CS8", "CU12", "CS12", "CU16", "CS16", "CF32", "CS32", "CF64"]
    out = []
    for _ in range(count):
        n = int(rs.choice([1024, 2048, 2048, 4096, 8192]))
        frames = int(rs.randint(100, 700))
        hop = max(1, n // int(rs.choice([1, 1, 2, 4, 8])))
        samples = n + (frames - 1) * hop + int(rs.randint(0, 5))
        out.append(dict(fmt=str(rs.choice(fmts)), n=n, width=frames, samples=samples, win="hann", gain=float(rs.randint(0, 40)),
                        rng=float(rs.choice([30, 60, 90])), ch=bool(rs.randint(4) == 0), wf=bool(rs.randint(4) == 0), lut_len=256,
                        seed=int(rs.randint(1 << 30)), amp=float(rs.choice([0.05, 0.5])), kind="trinoise"))
    return out


def check(ctx, c):
    kind = c["kind"] if not c["fmt"].startswith("CF") else "trinoise"
    gen = {"kind": kind, "seed": c["seed"], "step": 4099, "gshift": 9, "amp": c["amp"], "namp": 0.02}
    data = siggen.generate(c["fmt"], gen, c["samples"])
    win, weight = pyoracle.window(c["win"], c["n"])
    if weight == 0:          # hann / bartlett / blackman at n = 2: an all-zero taper
        return []
    i = np.arange(c["lut_len"])
    lut = np.stack([(i * 5) & 255, (i * 11 + 3) & 255, (255 - i) & 255], axis=1).astype(np.uint8)
    want = pyoracle.render(c["fmt"], data, c["n"], win, 1.0 / weight, c["gain"], c["rng"], lut, c["width"], c["ch"], c["wf"])
    got = ctx.render(c["fmt"], data, c["n"], win, 1.0 / weight, c["gain"], c["rng"], lut, c["width"], c["ch"], c["wf"])
    bad = [k for k in ("rgba", "gauge_mins", "gauge_maxs", "gauge_amps") if not np.array_equal(got[k], want[k])]
    bad += [k for k in ("c_hist", "cB_hist") if not np.array_equal(got[k].astype(np.int64), want[k])]
    for k in ("dBfs_min", "dBfs_max"):
        a, b = np.float64(got[k]), np.float64(want[k])
        if not ((a != a and b != b) or a.view(np.uint64) == b.view(np.uint64)):
            bad.append(k)
    return bad


def main():
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 7
    pkg = load_package()
    ctx = pkg.Context(0)
    cases = _random_cases(count, seed) + big_cases(max(1, count // 8), seed + 1)
    failed = 0
    for k, c in enumerate(cases):
        bad = check(ctx, c)
        if bad:
            failed += 1
            print("MISMATCH", bad, c, flush=True)
    print("soak: %d cases (%d long large-n), %d mismatches, seed %d" % (len(cases), max(1, count // 8), failed, seed))
    ctx.close()
    sys.exit(1 if failed else 0)


if __name__ == "__main__":
    main()
