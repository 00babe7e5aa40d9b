"""The two CPU oracles against each other on 10 000 seeded requests (no GPU): oracle/sp_oracle.c — the checker of every GPU parity test,
whose Math.cos / sin / log10 are the fdlibm restatement oracle/v8math.h — and oracle/js/worker_oracle.js under Node, i.e. on the real V8
routines the reference itself runs on.  Pins the restated engine math (twiddles, tapers, one log10 per pixel, 73 million pixels) beyond
the 24 576 known-answer points of tests/golden/math_*.bin.  Every output is compared bit for bit through a digest."""
import hashlib
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

import siggen
from __graft_entry__ import ROOT
from oracle import pyoracle

pytestmark = pytest.mark.skipif(shutil.which("node") is None, reason="node not installed")


def soak_cases(count, seed):
    rs = np.random.RandomState(seed)
    fmts = ["CU4", "CS4", "CU8", "CS8", "CU12", "CS12", "CU16", "CS16", "CU32", "CS32", "CU64", "CS64", "CF32", "CF64"]
    wins = ["rectangular", "bartlett", "hamming", "hann", "blackman", "blackmanHarris"]
    out = []
    for i in range(count):
        big = i % 40 == 0                                  # every 40th request is a large transform
        n = 1 << (int(rs.randint(10, 14)) if big else int(rs.randint(1, 10)))
        frames = int(rs.randint(1, 12 if big else 120))
        mode = rs.randint(4)                               # hop = n, fractional, overlap, sparse
        if mode == 0:
            samples = n * frames
        elif mode == 1:
            samples = n * frames + int(rs.randint(1, n))
        elif mode == 2:
            samples = n + (frames - 1) * max(1, n // int(rs.randint(2, 9))) + int(rs.randint(0, 3))
        else:
            samples = n * frames * int(rs.randint(2, 4)) + int(rs.randint(0, 7))
        out.append(dict(fmt=fmts[rs.randint(len(fmts))], n=n, width=frames, samples=max(samples, n), win=wins[rs.randint(len(wins))],
                        gain=float(rs.randint(-10, 60)), rng=float(rs.choice([6, 12, 30, 30, 45.5, 90, 120])), ch=bool(rs.randint(2)),
                        wf=bool(rs.randint(2)), lut_len=int(rs.choice([2, 3, 17, 64, 255, 256, 256, 300])), seed=int(rs.randint(1 << 30)),
                        amp=float(rs.choice([0.05, 0.5, 0.9])), kind=str(rs.choice(["trinoise", "trinoise", "bytes"]))))
        if out[-1]["fmt"].startswith("CF"):
            out[-1]["kind"] = "trinoise"                   # random bytes read as floats are mostly NaN / huge: not what this run is for
    return out


def c_digest(c):
    gen = {"kind": c["kind"], "seed": c["seed"], "step": 4099, "gshift": 9, "amp": c["amp"], "namp": 0.02}
    data = siggen.generate(c["fmt"], gen, c["samples"])
    win, weight = pyoracle.window(c["win"], c["n"])
    if weight == 0:
        return "skip"
    i = np.arange(c["lut_len"])
    lut = np.stack([(i * 5) & 255, (i * 11 + 3) & 255, (255 - i) & 255], axis=1).astype(np.uint8)
    r = pyoracle.render(c["fmt"], data, c["n"], win, 1.0 / weight, c["gain"], c["rng"], lut, c["width"], c["ch"], c["wf"])
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(r["rgba"]).tobytes())
    for k in ("gauge_mins", "gauge_maxs", "gauge_amps"):
        h.update(np.ascontiguousarray(r[k]).tobytes())
    h.update(np.asarray(r["c_hist"], dtype=np.float64).tobytes())
    h.update(np.asarray(r["cB_hist"], dtype=np.float64).tobytes())
    for k in ("dBfs_min", "dBfs_max"):
        v = np.float64(r[k])
        h.update(b"nan" if v != v else v.tobytes())
    return h.hexdigest()


def test_c_oracle_equals_js_oracle_under_v8_on_10000_requests(tmp_path):
    cases = soak_cases(10000, 20261003)
    f = tmp_path / "cases.json"
    f.write_text(json.dumps(cases))
    node = subprocess.Popen(["node", os.path.join(ROOT, "tests", "js", "soak_oracles.js"), str(f)], stdout=subprocess.PIPE, text=True)
    mine = [c_digest(c) for c in cases]                    # the C oracle works while Node does
    out, _ = node.communicate(timeout=900)
    assert node.returncode == 0
    theirs = out.split()
    assert len(theirs) == len(mine) == 10000
    bad = [(i, cases[i]) for i in range(len(cases)) if mine[i] != theirs[i]]
    assert not bad, bad[:5]
    assert sum(1 for d in mine if d != "skip") > 9000
