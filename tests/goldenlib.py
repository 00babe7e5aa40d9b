"""Access to the committed golden vectors (tests/golden/, produced by oracle/gen_golden.js from the real reference)."""
import hashlib
import json
import os
import struct

import numpy as np

import siggen

GDIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def f64_from_hex(h):
    return struct.unpack("<d", struct.pack("<Q", int(h, 16)))[0]


def f64_hex(v):
    return "%016x" % struct.unpack("<Q", struct.pack("<d", v))[0]


def same_f64(v, h):
    e = f64_from_hex(h)
    if e != e:
        return v != v
    return f64_hex(v) == h


def sha256(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


class Golden:
    def __init__(self):
        with open(os.path.join(GDIR, "cases.json")) as f:
            self.spec = json.load(f)
        with open(os.path.join(GDIR, "worker_expected.json")) as f:
            self.expected = {e["name"]: e for e in json.load(f)}
        with open(os.path.join(GDIR, "cmaps.json")) as f:
            self.cmap_index = {e["name"]: e for e in json.load(f)}
        self.cmap_bin = np.fromfile(os.path.join(GDIR, "cmaps.bin"), dtype=np.uint8)
        self.cases = {c["name"]: c for c in self.spec["worker_cases"]}

    def lut(self, case_or_name, force_ends=None):
        """(len, 3) uint8 LUT for a case (or a bare cmap name), with the caller's end forcing applied when asked."""
        if isinstance(case_or_name, str):
            name, force = case_or_name, bool(force_ends)
        else:
            name, force = case_or_name["cmap"], case_or_name["force_ends"] if force_ends is None else force_ends
        if name.startswith("custom:"):
            n = int(name.split(":")[1])
            i = np.arange(n)
            lut = np.stack([(i * 7) & 255, (i * 13 + 5) & 255, (255 - i) & 255], axis=1).astype(np.uint8)
        else:
            e = self.cmap_index[name + "_cmap"]
            lut = self.cmap_bin[e["offset"]:e["offset"] + 3 * e["length"]].reshape(-1, 3).copy()
        if force:
            lut[0] = (0, 0, 0)
            lut[-1] = (255, 255, 255)
        return lut

    def input(self, case):
        return siggen.case_input(case)

    def file(self, name):
        return os.path.join(GDIR, name)


def check_reply(r, e, what=""):
    """Compares a render result dict (pyoracle.render layout) with a golden reply digest; returns a list of mismatches."""
    bad = []
    if sha256(r["rgba"]) != e["rgba_sha256"]:
        bad.append("rgba")
    for k in ("gauge_mins", "gauge_maxs", "gauge_amps"):
        if bytes(r[k]).hex() != e[k]:
            bad.append(k)
    if [int(v) for v in r["c_hist"]] != e["c_hist"]:
        bad.append("c_hist")
    cb = {str(i): int(v) for i, v in enumerate(r["cB_hist"]) if v}
    if cb != e["cB_hist"]:
        bad.append("cB_hist")
    if not same_f64(r["dBfs_min"], e["dBfs_min"]):
        bad.append("dBfs_min %r vs %s" % (r["dBfs_min"], e["dBfs_min_num"]))
    if not same_f64(r["dBfs_max"], e["dBfs_max"]):
        bad.append("dBfs_max %r vs %s" % (r["dBfs_max"], e["dBfs_max_num"]))
    return ["%s%s" % (what, b) for b in bad]
