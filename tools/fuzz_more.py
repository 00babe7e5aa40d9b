"""One-off extended fuzz: 640 more seeded random requests (all formats, n = 2..8192, both layouts) through sp_render against the C oracle."""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as T
from __graft_entry__ import load_package
from oracle import pyoracle
import siggen
pkg = load_package(); ctx = pkg.Context(0)
bad = 0; tot = 0
for seed in (1, 2, 3, 4, 5, 6, 7, 8):
    for c in T._random_cases(80, 1000 + seed):
        kind = c["kind"] if not c["fmt"].startswith("CF") else "trinoise"
        gen = {"kind": kind, "seed": c["seed"], "step": 4099, "gshift": 9, "amp": c["amp"], "namp": 0.02}
        data = siggen.generate(c["fmt"], gen, c["samples"])
        win, weight = pyoracle.window(c["win"], c["n"])
        with np.errstate(divide="ignore"):
            bn = float(np.float64(1.0) / np.float64(weight))
        i = np.arange(c["lut_len"])
        lut = np.stack([(i * 5) & 255, (i * 11 + 3) & 255, (255 - i) & 255], axis=1).astype(np.uint8)
        want = pyoracle.render(c["fmt"], data, c["n"], win, bn, c["gain"], c["rng"], lut, c["width"], c["ch"], c["wf"])
        try:
            got = ctx.render(c["fmt"], data, c["n"], win, bn, c["gain"], c["rng"], lut, c["width"], c["ch"], c["wf"])
        except Exception as e:
            bad += 1; tot += 1
            print("RENDER ERROR", c, "block_norm", bn, repr(e)[:200])
            continue
        tot += 1
        try:
            T._assert_same(got, want)
        except AssertionError as e:
            bad += 1
            print("MISMATCH", c, "block_norm", bn, str(e)[:100])
        except Exception as e:
            bad += 1
            print("ERROR", c, "block_norm", bn, repr(e)[:200])
print("cases", tot, "mismatches", bad)
