"""sp_group's MULTI-member RCCL gather on one GPU, behind a test double of librccl (tests/cpp/rccl_shim.cpp).

The real library refuses two ranks on one device, and the test boxes have one GPU: without the double the branch that north_star names -
"an RCCL gather of the RGBA strip over xGMI" with more than one rank: comms[r], the receive addresses of both layouts, the strips staged
beside the image and re-tiled - would run for the first time on the driver's 8-GPU node.  The double has RCCL's point-to-point semantics
(grouped sends and receives matched pair by pair, byte counts checked, stream order on both sides) and counts what it moved.
Everything is compared bit-for-bit with the merged vectors of the real reference run with the same number of workers
(lib/spectroplot.js:1206-1244, lib/samples.js:253-258)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import goldenlib
from __graft_entry__ import ROOT, load_package
from oracle import pyoracle
from test_gpu_parity import _check_group_against_merged_vectors

pytestmark = pytest.mark.gpu

SHIM = os.path.join(ROOT, "tests", "cpp", "_build", "librccl_shim.so")
STAT = ("comms_made", "comms_destroyed", "comms_aborted", "sends", "receives", "groups", "pairs", "bytes", "cross_device", "injected",
        "largest", "inits")


@pytest.fixture(scope="module")
def pkg():
    return load_package()


class Shim:
    def __init__(self):
        if not os.path.exists(SHIM):
            subprocess.check_call(["make", "-s", "-C", os.path.dirname(os.path.dirname(SHIM))])
        self.lib = C.CDLL(SHIM)    # the same mapping the product's dlopen of this path gets: its counters are the ones read here

    def reset(self):
        self.lib.rccl_shim_reset()

    def stats(self):
        a = (C.c_uint64 * 12)()
        self.lib.rccl_shim_stats(a)
        return dict(zip(STAT, [int(v) for v in a]))


@pytest.fixture()
def shim(monkeypatch):
    s = Shim()
    s.reset()
    monkeypatch.setenv("SPECTROPLOT_HIP_FORCE_RCCL", "1")
    monkeypatch.setenv("SPECTROPLOT_HIP_RCCL_LIB", SHIM)
    for k in ("RCCL_SHIM_FAIL_INIT", "RCCL_SHIM_FAIL_SEND", "RCCL_SHIM_FAIL_RECV", "RCCL_SHIM_FAIL_GROUPEND", "RCCL_SHIM_REFUSE_DUPLICATES",
              "SPECTROPLOT_HIP_NO_RCCL", "SPECTROPLOT_HIP_ASSUME_NO_PEER"):
        monkeypatch.delenv(k, raising=False)
    return s


def _merged_cases(golden, members):
    return [c for c in golden.spec["worker_cases"] if "merged" in golden.expected[c["name"]] and c["slices"] == members]


@pytest.mark.parametrize("members", [2, 3, 8])
def test_multi_member_rccl_gather_matches_the_callers_merge(pkg, golden, shim, members):
    """2, 3 and 8 members, both layouts: transport "rccl", results bit-exact.  Per render the members other than the root each send one
    strip and one record block (the root's own strip is placed by a device copy, first_sender = 1), the byte counts are the strips' and
    the blocks', and the root stages exactly (members - 1) strips - and only for the spectrogram layout, whose column bands a contiguous
    receive cannot write; the waterfall's receives land in the image's row bands."""
    g = pkg.Group([0] * members)
    _check_group_against_merged_vectors(pkg, golden, g, members, "device", "rccl")
    assert g.transport_note() == "", g.transport_note()
    info = g.rccl_info()
    assert "librccl_shim" in info and "%d communicator" % members in info and "29999" in info, info
    cases = _merged_cases(golden, members)
    st = shim.stats()
    assert st["inits"] == 1 and st["comms_made"] == members, st
    assert st["groups"] == len(cases) and st["sends"] == st["receives"] == 2 * (members - 1) * len(cases) == st["pairs"], st
    want_bytes, want_stage = 0, 0
    for c in cases:
        sw = c["width"] // members
        strip = 4 * sw * c["n"]
        small = ((len(golden.lut(c)) + 1000 + 2) * 8 + 3 * sw + 15) & ~15
        want_bytes += (members - 1) * (strip + small)
        if not c["waterfall"]:
            want_stage = max(want_stage, (members - 1) * strip)
    assert st["bytes"] == want_bytes, (st, want_bytes)
    assert st["cross_device"] == 0 and st["injected"] == 0
    image_bytes, staging_bytes = g.root_bytes()
    assert staging_bytes == (want_stage + 16 + 256 if want_stage else 0), (staging_bytes, want_stage)
    g.close()
    st = shim.stats()
    assert st["comms_destroyed"] == members and st["comms_aborted"] == 0, st


def test_waterfall_only_group_stages_nothing(pkg, golden, shim):
    """The waterfall layout's strips are row bands: received straight into the image, no block beside it."""
    g = pkg.Group([0, 0, 0])
    c = golden.cases["slice_CF32_wf_3"]
    win, weight = pyoracle.window(c["window"], c["n"])
    m = g.render(c["format"], golden.input(c), c["n"], win, 1.0 / weight, c["gain"], c["range"], golden.lut(c, force_ends=c["force_ends"]),
                 c["width"], c["channelMode"], True)
    assert goldenlib.sha256(m["rgba"]) == golden.expected[c["name"]]["merged"]["rgba_sha256"]
    assert g.transport() == "rccl" and g.root_bytes()[1] == 0
    g.close()


@pytest.mark.parametrize("what,k", [("SEND", 1), ("SEND", 2), ("SEND", 4), ("RECV", 1), ("RECV", 3), ("GROUPEND", 1), ("INIT", 1)])
def test_an_rccl_failure_in_a_multi_member_exchange_ends_in_peer_copies(pkg, golden, shim, monkeypatch, what, k):
    """The k-th ncclSend / ncclRecv of the exchange fails, ncclGroupEnd fails with half of its copies started, ncclCommInitAll refuses:
    every render still returns the reference's merged result (peer copies redo the gather), the communicators are aborted, the note has
    ONE entry naming the call, and the next renders do not go back to RCCL."""
    monkeypatch.setenv("RCCL_SHIM_FAIL_" + what, str(k))
    g = pkg.Group([0, 0, 0])
    # (the staging block of the attempted spectrogram exchange stays allocated: staged_before)
    _check_group_against_merged_vectors(pkg, golden, g, 3, "device", "peer", staged_before=True)
    note = g.transport_note()
    where = {"SEND": "ncclSend", "RECV": "ncclSend / ncclRecv", "GROUPEND": "ncclGroupEnd", "INIT": "ncclCommInitAll"}[what]
    assert note.count("RCCL not used") == 1 and where in note and "(shim)" in note, note
    st = shim.stats()
    assert st["injected"] == 1 and st["inits"] == 1, st
    assert st["comms_aborted"] == (0 if what == "INIT" else 3) and st["comms_destroyed"] == 0, st
    assert st["groups"] == (0 if what == "INIT" else 1), st      # one exchange was tried, the later renders tried none
    g.close()


def test_real_duplicate_refusal_is_modelled_too(pkg, golden, shim, monkeypatch):
    """With RCCL_SHIM_REFUSE_DUPLICATES the double behaves like the real library on this box (a device listed twice is refused)."""
    monkeypatch.setenv("RCCL_SHIM_REFUSE_DUPLICATES", "1")
    g = pkg.Group([0, 0])
    _check_group_against_merged_vectors(pkg, golden, g, 2, "device", "peer")
    assert "ncclCommInitAll" in g.transport_note()
    g.close()


@pytest.mark.parametrize("rccl", [False, True], ids=["peer_copies", "rccl_then_none"])
def test_members_the_root_cannot_address_are_staged_and_retiled(pkg, golden, shim, monkeypatch, rccl):
    """SPECTROPLOT_HIP_ASSUME_NO_PEER=1: every other member is treated as one whose memory the root cannot address - its spectrogram
    strip goes into the staging block (hipMemcpyPeerAsync between devices) and is re-tiled by the root (sp_group.hip's `restage`
    branch).  Under RCCL nothing changes: its receives never needed peer access."""
    monkeypatch.setenv("SPECTROPLOT_HIP_ASSUME_NO_PEER", "1")
    if not rccl:
        monkeypatch.setenv("SPECTROPLOT_HIP_NO_RCCL", "1")
    g = pkg.Group([0] * 8)
    _check_group_against_merged_vectors(pkg, golden, g, 8, "device", "rccl" if rccl else "peer", staged_before=True)
    assert "SPECTROPLOT_HIP_ASSUME_NO_PEER" in g.transport_note()
    strips = max(7 * 4 * (c["width"] // 8) * c["n"] for c in _merged_cases(golden, 8) if not c["waterfall"])
    assert g.root_bytes()[1] == strips + 16 + 256, g.root_bytes()
    assert shim.stats()["pairs"] == (0 if not rccl else 2 * 7 * len(_merged_cases(golden, 8)))
    g.close()


def test_shim_gather_of_a_seeded_capture_against_the_oracle(pkg, shim):
    """Beyond the golden shapes: 2^21 cs16 samples, n = 2048, five members, width 1003 (three columns stay blank), both layouts -
    every strip against the oracle's render of that slice with its own stride (lib/worker.js:50), through the double's exchange."""
    import siggen
    fmt, n, width, members = "CS16", 2048, 1003, 5
    data = siggen.generate(fmt, {"kind": "trinoise", "seed": 77, "step": 5557, "gshift": 12, "amp": 0.4, "namp": 0.03}, 1 << 21)
    win, weight = pyoracle.window("hann", n)
    lut = np.stack([np.arange(256), (np.arange(256) * 5) & 255, 255 - np.arange(256)], axis=1).astype(np.uint8)
    g = pkg.Group([0] * members)
    sw = width // members
    for waterfall in (False, True):
        m = g.render(fmt, data, n, win, 1.0 / weight, 6.0, 40.0, lut, width, False, waterfall, dirty=0x5A)
        assert g.transport() == "rccl", g.transport_note()
        img = m["rgba"].reshape(width, n, 4) if waterfall else m["rgba"].reshape(n, width, 4)
        c_sum = np.zeros(256, np.int64)
        for r in range(members):
            b0, b1 = pkg.slice_bounds(data.size, 4, r, members)
            want = pyoracle.render(fmt, data[b0:b1], n, win, 1.0 / weight, 6.0, 40.0, lut, sw, False, waterfall)
            if waterfall:
                got = img[width - sw - sw * r: width - sw * r]
                assert np.array_equal(got.reshape(-1), want["rgba"]), (r, "waterfall")
            else:
                got = img[:, sw * r: sw * (r + 1)]
                assert np.array_equal(got, want["rgba"].reshape(n, sw, 4)), (r, "spectrogram")
            c_sum += want["c_hist"]
            for k in ("gauge_mins", "gauge_maxs", "gauge_amps"):
                assert np.array_equal(m[k][sw * r: sw * (r + 1)], want[k]), (r, k)
        assert np.array_equal(m["c_hist"].astype(np.int64), c_sum)
        rest = img[:width - sw * members] if waterfall else img[:, sw * members:]
        assert not rest.any()
    g.close()
