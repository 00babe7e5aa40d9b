"""Pins both oracles (oracle/sp_oracle.c and oracle/js/worker_oracle.js) against the vectors produced by the real
reference worker.  CPU only."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

import goldenlib
import siggen
from goldenlib import f64_from_hex, same_f64, sha256
from oracle import pyoracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_input_generator_matches_js(golden):
    for c in golden.spec["worker_cases"]:
        assert sha256(golden.input(c)) == golden.expected[c["name"]]["input_sha256"], c["name"]


def test_c_oracle_reproduces_worker_vectors(golden):
    bad = []
    for c in golden.spec["worker_cases"]:
        e = golden.expected[c["name"]]
        data = golden.input(c)
        win, weight = pyoracle.window(c["window"], c["n"]) if "throws" not in e or c["n"] & (c["n"] - 1) == 0 else (np.ones(c["n"]), 1.0)
        bn = 1.0 / weight
        lut = golden.lut(c)
        kw = dict(channel_mode=c["channelMode"], waterfall=c["waterfall"])
        if "throws" in e:
            with pytest.raises(pyoracle.OracleError):
                pyoracle.render(c["format"], data, c["n"], win, bn, c["gain"], c["range"], lut, c["width"], **kw)
            continue
        assert same_f64(bn, e["block_norm"]), c["name"]
        if "reply" in e:
            r = pyoracle.render(c["format"], data, c["n"], win, bn, c["gain"], c["range"], lut, c["width"], **kw)
            bad += goldenlib.check_reply(r, e["reply"], c["name"] + ": ")
            continue
        # the caller's slice + merge (spectroplot.js:1206-1244)
        k = c["slices"]
        sw = pyoracle.decode(c["format"], np.zeros(0, np.uint8), 0, 0)[2]
        sl_w = c["width"] // k
        n = c["n"]
        W = c["width"]
        merged = np.zeros((W, n, 4) if c["waterfall"] else (n, W, 4), dtype=np.uint8)
        c_hist = np.zeros(len(lut), dtype=np.int64)
        dmin, dmax = 0.0, -200.0
        for i in range(k):
            b0, b1 = pyoracle.slice_bounds(len(data), sw, i, k)
            assert b1 - b0 == e["slices"][i]["slice_bytes"]
            r = pyoracle.render(c["format"], data[b0:b1], n, win, bn, c["gain"], c["range"], lut, sl_w, **kw)
            bad += goldenlib.check_reply(r, e["slices"][i], "%s[%d]: " % (c["name"], i))
            c_hist += r["c_hist"]
            dmin = r["dBfs_min"] if r["dBfs_min"] < dmin else dmin
            dmax = r["dBfs_max"] if r["dBfs_max"] > dmax else dmax
            off = i * sl_w
            if c["waterfall"]:
                y0 = W - sl_w - off
                merged[y0:y0 + sl_w] = r["rgba"].reshape(sl_w, n, 4)
            else:
                merged[:, off:off + sl_w] = r["rgba"].reshape(n, sl_w, 4)
        assert sha256(merged) == e["merged"]["rgba_sha256"], c["name"]
        assert [int(v) for v in c_hist] == e["merged"]["c_hist"], c["name"]
        assert same_f64(dmin, e["merged"]["dBfs_min"]) and same_f64(dmax, e["merged"]["dBfs_max"]), c["name"]
    assert not bad, bad[:20]


def test_c_oracle_windows(golden):
    idx = json.load(open(golden.file("windows.json")))
    blob = open(golden.file("windows.bin"), "rb").read()
    for e in idx:
        w, weight = pyoracle.window(e["name"], e["n"])
        assert sha256(w) == e["sha256"], (e["name"], e["n"])
        assert same_f64(weight, e["weight"]), (e["name"], e["n"])
        if e["offset"] >= 0:
            assert w.tobytes() == blob[e["offset"]:e["offset"] + 8 * e["n"]]


def test_c_oracle_twiddles_and_fft(golden):
    F = json.load(open(golden.file("fft.json")))
    for t in F["twiddles"]:
        c, s = pyoracle.twiddles(t["n"])
        assert sha256(np.concatenate([c, s])) == t["sha256"], t["n"]
    kept = np.fromfile(golden.file("twiddles_8192.bin"), dtype=np.float64)
    c, s = pyoracle.twiddles(8192)
    assert np.array_equal(np.concatenate([c, s]).view(np.uint64), kept.view(np.uint64))
    for k in F["cases"]:
        n = k["n"]
        re = np.zeros(n)
        im = np.zeros(n)
        if k["kind"] == "impulse":
            re[k["pos"]], im[k["pos"]] = 1.0, -0.5
        elif k["kind"] == "dc":
            re[:], im[:] = 0.75, -0.25
        else:
            i = np.arange(n, dtype=np.uint32)
            re = siggen.hash32(k["seed"], 2 * i).astype(np.float64) / 2147483648.0 - 1.0
            im = siggen.hash32(k["seed"], 2 * i + 1).astype(np.float64) / 2147483648.0 - 1.0
        ro, io = pyoracle.fft(re, im, split=bool(k.get("split")))
        assert sha256(np.concatenate([ro, io])) == k["sha256"], k
    with pytest.raises(pyoracle.OracleError):
        pyoracle.twiddles(12)


def test_c_oracle_decode(golden):
    D = {e["name"]: e for e in json.load(open(golden.file("decode.json")))}
    for k in golden.spec["decode_kat"]:
        e = D[k["name"]]
        if "hex" in k:
            data = np.frombuffer(bytes.fromhex(k["hex"]), dtype=np.uint8)
        else:
            sw = siggen.SAMPLE_WIDTH.get(k["gen_format"], 2)
            data = siggen.generate(k["gen_format"], {"kind": "bytes", "seed": k["seed"]}, -(-k["bytes"] // sw))[:k["bytes"]]
        if "throws" in e:
            with pytest.raises(pyoracle.OracleError):
                pyoracle.decode(k["format"], data, 0, 1)
            continue
        vals, count, sw = pyoracle.decode(k["format"], data, k["pos_lo"], k["pos_hi"] - k["pos_lo"])
        assert sw == e["sampleWidth"] and same_f64(count, e["sampleCount"]), k["name"]
        flat = vals.reshape(-1)
        assert len(flat) == len(e["values"])
        for v, h in zip(flat, e["values"]):
            assert same_f64(float(v), h), (k["name"], v, f64_from_hex(h))


def test_engine_math_restatement(golden):
    lg = np.fromfile(golden.file("math_log10.bin"), dtype=np.float64)
    n = len(lg) // 2
    L = pyoracle.lib()
    for x, e in zip(lg[:n], lg[n:]):
        r = L.spo_log10(float(x))
        assert (r != r and e != e) or np.float64(r).view(np.uint64) == np.float64(e).view(np.uint64), x
    tg = np.fromfile(golden.file("math_trig.bin"), dtype=np.float64)
    m = len(tg) // 3
    for t, c, s in zip(tg[:m], tg[m:2 * m], tg[2 * m:]):
        assert np.float64(L.spo_cos(float(t))).view(np.uint64) == np.float64(c).view(np.uint64), t
        assert np.float64(L.spo_sin(float(t))).view(np.uint64) == np.float64(s).view(np.uint64), t


@pytest.mark.skipif(shutil.which("node") is None, reason="node not installed")
def test_js_oracle_reproduces_all_vectors():
    out = subprocess.run(["node", os.path.join(ROOT, "oracle/js/check_golden.js")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "bit-for-bit" in out.stdout
