"""GPU parity tests: the HIP path, called through the C ABI, against the golden vectors of the real reference worker
and against the C oracle on seeded inputs.  Bit-exact: RGBA bytes, histograms, gauges and the f64 dBfs range.
Tolerance: none (every compared quantity is an integer, a byte, or an f64 that is required to be bit-identical)."""
import numpy as np
import pytest

import goldenlib
import siggen
from __graft_entry__ import load_package
from oracle import pyoracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    return load_package()


@pytest.fixture(scope="module")
def ctx(pkg):
    c = pkg.Context(0)
    yield c
    c.close()


def _plan_render(pkg, ctx, kernel, fmt, data, n, win, bn, gain, rng, lut, width, channel_mode, waterfall):
    """Renders through sp_plan_execute with a forced kernel variant (device-resident operands)."""
    plan = ctx.plan(fmt, n, win, bn, gain, rng, lut, channel_mode, waterfall)
    try:
        plan.force_kernel(kernel)
    except pkg.SpectroplotError:
        plan.close()
        return None
    W, L = int(width), len(lut)
    d_in = ctx.alloc(max(data.size, 16))
    ctx.upload(d_in, data) if data.size else None
    sizes = [4 * W * n, W, W, W, 8 * L, 8000, 16]
    ptrs = [ctx.alloc(max(s, 16)) for s in sizes]
    for p, s in zip(ptrs, sizes):
        ctx.memset(p, 0, max(s, 16))
    plan.execute(d_in, data.size, W, *ptrs)
    ctx.synchronize()
    out = {"rgba": ctx.download(ptrs[0], sizes[0]), "gauge_mins": ctx.download(ptrs[1], W), "gauge_maxs": ctx.download(ptrs[2], W),
           "gauge_amps": ctx.download(ptrs[3], W), "c_hist": ctx.download(ptrs[4], 8 * L, np.uint64),
           "cB_hist": ctx.download(ptrs[5], 8000, np.uint64)}
    mm = ctx.download(ptrs[6], 16, np.float64)
    out["dBfs_min"], out["dBfs_max"] = float(mm[0]), float(mm[1])
    out["kernel"] = plan.kernel_name()
    for p in ptrs + [d_in]:
        ctx.free(p)
    plan.close()
    return out


def _case_args(golden, c):
    data = golden.input(c)
    win, weight = pyoracle.window(c["window"], c["n"])
    return data, win, 1.0 / weight, golden.lut(c)


def test_golden_worker_cases_sp_render(pkg, ctx, golden):
    """Every worker vector of the real reference through sp_render (host buffers, automatic kernel choice)."""
    bad = []
    for c in golden.spec["worker_cases"]:
        e = golden.expected[c["name"]]
        if "reply" not in e and "throws" not in e:
            continue
        if "throws" in e and c["n"] & (c["n"] - 1):
            data, win, bn, lut = golden.input(c), np.ones(c["n"]), 1.0, golden.lut(c)
        else:
            data, win, bn, lut = _case_args(golden, c)
        args = (c["format"], data, c["n"], win, bn, c["gain"], c["range"], lut, c["width"], c["channelMode"], c["waterfall"])
        if "throws" in e:
            with pytest.raises(pkg.SpectroplotError) as ei:
                ctx.render(*args)
            want = -2 if "power of 2" in e["throws"] else -3
            assert ei.value.status == want, (c["name"], ei.value.status)
            continue
        r = ctx.render(*args)
        bad += goldenlib.check_reply(r, e["reply"], c["name"] + ": ")
    assert not bad, bad[:30]


@pytest.mark.parametrize("kernel", ["scratch", "frames"])
def test_golden_worker_cases_each_kernel(pkg, ctx, golden, kernel):
    """The same vectors through sp_plan_execute with each device kernel forced (device-resident operands)."""
    bad, ran = [], 0
    for c in golden.spec["worker_cases"]:
        e = golden.expected[c["name"]]
        if "reply" not in e or c["width"] == 0:
            continue
        data, win, bn, lut = _case_args(golden, c)
        r = _plan_render(pkg, ctx, kernel, c["format"], data, c["n"], win, bn, c["gain"], c["range"], lut, c["width"],
                         c["channelMode"], c["waterfall"])
        if r is None:
            continue   # this kernel does not cover the case's n
        ran += 1
        bad += goldenlib.check_reply(r, e["reply"], "%s[%s]: " % (c["name"], r["kernel"]))
    assert ran > 40
    assert not bad, bad[:30]


def test_golden_slices_through_worker_mirror(pkg, golden):
    """The caller's slice + merge (spectroplot.js:1206-1244) over HipWorker instances, against the merged vectors."""
    w = pkg.HipWorker(0)
    for c in golden.spec["worker_cases"]:
        e = golden.expected[c["name"]]
        if "merged" not in e:
            continue
        data = golden.input(c)
        win, weight = pyoracle.window(c["window"], c["n"])
        cmap = golden.lut(c, force_ends=False).tolist()
        m = pkg.render_sliced(w.render, data, c["format"], c["n"], c["width"], c["slices"], win, weight, cmap, c["gain"], c["range"],
                              c["channelMode"], c["waterfall"], force_ends=c["force_ends"])
        assert goldenlib.sha256(m["data"]) == e["merged"]["rgba_sha256"], c["name"]
        assert [int(v) for v in m["c_hist"]] == e["merged"]["c_hist"], c["name"]
        assert goldenlib.same_f64(m["dBfs_min"], e["merged"]["dBfs_min"]) and goldenlib.same_f64(m["dBfs_max"], e["merged"]["dBfs_max"])
        assert m["slice_width"] == e["merged"]["slice_width"]
        for i, r in enumerate(m["replies"]):
            rr = {"rgba": r["imageData"]["data"], "gauge_mins": r["gauge_mins"], "gauge_maxs": r["gauge_maxs"],
                  "gauge_amps": r["gauge_amps"], "c_hist": r["c_hist"], "cB_hist": r["cB_hist"], "dBfs_min": r["dBfs_min"],
                  "dBfs_max": r["dBfs_max"]}
            assert not goldenlib.check_reply(rr, e["slices"][i]), (c["name"], i)
    w.terminate()


def _check_group_against_merged_vectors(pkg, golden, g, members, gather, expect_transport, staged_before=False):
    ran = 0
    for c in golden.spec["worker_cases"]:
        e = golden.expected[c["name"]]
        if "merged" not in e or c["slices"] != members:
            continue
        data = golden.input(c)
        win, weight = pyoracle.window(c["window"], c["n"])
        lut = golden.lut(c, force_ends=c["force_ends"])
        # (the caller's buffers arrive dirty: what no slice draws must come back cleared, as on the caller's fresh canvas)
        m = g.render(c["format"], data, c["n"], win, 1.0 / weight, c["gain"], c["range"], lut, c["width"], c["channelMode"], c["waterfall"],
                     gather=gather, dirty=0xAB)
        assert goldenlib.sha256(m["rgba"]) == e["merged"]["rgba_sha256"], c["name"]
        assert [int(v) for v in m["c_hist"]] == e["merged"]["c_hist"], c["name"]
        assert goldenlib.same_f64(m["dBfs_min"], e["merged"]["dBfs_min"]) and goldenlib.same_f64(m["dBfs_max"], e["merged"]["dBfs_max"]), c["name"]
        sw = m["slice_width"]
        assert sw == e["merged"]["slice_width"]
        cb = np.zeros(1000, dtype=np.int64)
        for i, es in enumerate(e["slices"]):
            for k in ("gauge_mins", "gauge_maxs", "gauge_amps"):
                assert bytes(m[k][i * sw:(i + 1) * sw]).hex() == es[k], (c["name"], i, k)
            for b, v in es["cB_hist"].items():
                cb[int(b)] += v
        assert np.array_equal(m["cB_hist"].astype(np.int64), cb), c["name"]
        for k in ("gauge_mins", "gauge_maxs", "gauge_amps"):
            assert not m[k][members * sw:].any(), (c["name"], k)
        assert g.transport() == expect_transport, (g.transport(), g.transport_note())
        t = g.timings()
        assert t[0] > 0 and all(v >= 0 for v in t), t
        image_bytes, staging_bytes = g.root_bytes()
        if gather == "device" and expect_transport in ("none", "peer") and not staged_before:
            # copies land in the image itself: the root holds the image and nothing beside it
            assert staging_bytes == 0 and image_bytes >= m["rgba"].size, (image_bytes, staging_bytes)
        ran += 1
    assert ran >= 1, "no golden case with %d slices" % members


@pytest.mark.parametrize("gather", ["device", "host"])
@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0], [0, 0, 0, 0, 0, 0, 0, 0]], ids=lambda d: "%d_members" % len(d))
def test_group_render_merges_on_the_device_like_the_callers_merge(pkg, golden, devices, gather):
    """sp_group_render_ex: the caller's slice + merge (lib/spectroplot.js:1206-1244) from one process, against the merged vectors of the
    real reference for the same worker count, both layouts, widths that leave columns un-rendered.  gather = device: strips meet in the
    root member's HBM (one member needs no transport; several members on this box's one GPU take peer copies, straight into their bands
    of the image).  gather = host: every member writes its band of the host image itself and the side outputs are merged on the host."""
    g = pkg.Group(devices)
    _check_group_against_merged_vectors(pkg, golden, g, len(devices), gather,
                                        "host" if gather == "host" else "none" if len(devices) == 1 else "peer")
    g.close()


def test_group_forced_rccl_moves_the_roots_strip_through_a_self_exchange(pkg, golden, monkeypatch):
    """SPECTROPLOT_HIP_FORCE_RCCL=1: what one GPU can execute of north_star's transport.  A one-member group loads librccl,
    makes a one-rank communicator (ncclCommInitAll) and moves its own strip and record block through a grouped ncclSend / ncclRecv to
    itself - the dlopen table, the signatures, ncclUint8, the group call and the stream ordering of the multi-GPU gather - instead of
    two device copies; both layouts (the waterfall's receive lands in the image, the spectrogram's beside it)."""
    monkeypatch.setenv("SPECTROPLOT_HIP_FORCE_RCCL", "1")
    g = pkg.Group([0])
    _check_group_against_merged_vectors(pkg, golden, g, 1, "device", "rccl")
    assert g.transport_note() == "", g.transport_note()
    info = g.rccl_info()               # e.g. "/opt/rocm/lib/librccl.so.1, ncclGetVersion 22707, 1 communicator(s)"
    print("RCCL in use:", info)
    assert "rccl" in info.lower() and "1 communicator" in info and int(info.split("ncclGetVersion ")[1].split(",")[0]) >= 20000, info
    image_bytes, staging_bytes = g.root_bytes()
    assert staging_bytes > 0          # the spectrogram layout's strip arrived beside the image
    g.close()


def test_group_rccl_failures_end_in_peer_copies_with_a_note(pkg, golden, monkeypatch):
    """No RCCL failure may fail a render or be retried on every render.  (a) A library that cannot be loaded (bogus name through
    SPECTROPLOT_HIP_RCCL_LIB); (b) a real librccl whose ncclCommInitAll refuses the member list (two members on one device: 'duplicate
    GPU').  Both: results from peer copies, bit-exact against the merged vectors, the reason in transport_note(), and the second
    render does not try again (the note stays one entry long)."""
    monkeypatch.setenv("SPECTROPLOT_HIP_FORCE_RCCL", "1")
    monkeypatch.setenv("SPECTROPLOT_HIP_RCCL_LIB", "/nonexistent/librccl_bogus.so")
    g = pkg.Group([0, 0])
    _check_group_against_merged_vectors(pkg, golden, g, 2, "device", "peer")
    note = g.transport_note()
    assert "RCCL not used" in note and "librccl_bogus" in note and note.count("RCCL not used") == 1, note
    g.close()
    monkeypatch.delenv("SPECTROPLOT_HIP_RCCL_LIB")
    g = pkg.Group([0, 0])
    _check_group_against_merged_vectors(pkg, golden, g, 2, "device", "peer")
    note = g.transport_note()
    assert "ncclCommInitAll" in note and note.count("RCCL not used") == 1, note
    g.close()


def test_batched_merge_equals_the_merge_render_by_render(pkg, ctx):
    """sp_merge_replies_batch (one launch per collective in bench.py --gpus N) against sp_merge_replies called per render on the
    re-ordered records: sums of the histograms, min / max of the dBfs ranges (lib/spectroplot.js:1229-1238), for 1 ... 8 ranks and
    batches of 1, 3 and 16 renders, LUTs of 2, 64 and 256 entries."""
    rs = np.random.RandomState(5)
    for ranks, renders, L in ((1, 1, 256), (2, 3, 64), (3, 16, 256), (8, 16, 256), (8, 5, 2)):
        P = L + 1000 + 2
        g = rs.randint(0, 1 << 40, size=(ranks, renders, P)).astype(np.uint64)
        mm = -200.0 * rs.rand(ranks, renders, 2)
        g[:, :, L + 1000:] = mm.view(np.uint64)
        d_g, d_out, d_one, d_rec = ctx.alloc(g.nbytes), ctx.alloc(8 * renders * P), ctx.alloc(8 * P), ctx.alloc(8 * ranks * P)
        ctx.upload(d_g, g.reshape(-1).view(np.uint8))
        ctx.merge_replies_batch(d_g, ranks, renders, L, d_out)
        ctx.synchronize()
        got = ctx.download(d_out, 8 * renders * P, np.uint64).reshape(renders, P)
        for j in range(renders):
            ctx.upload(d_rec, np.ascontiguousarray(g[:, j, :]).reshape(-1).view(np.uint8))
            ctx.merge_replies(d_rec, ranks, L, d_one, d_one + 8 * L, d_one + 8 * (L + 1000))
            ctx.synchronize()
            one = ctx.download(d_one, 8 * P, np.uint64)
            assert np.array_equal(got[j], one), (ranks, renders, L, j)
            assert np.array_equal(got[j][:L + 1000], g[:, j, :L + 1000].sum(axis=0, dtype=np.uint64))
            assert got[j][L + 1000:].view(np.float64)[0] == mm[:, j, 0].min() and got[j][L + 1000:].view(np.float64)[1] == mm[:, j, 1].max()
        for p_ in (d_g, d_out, d_one, d_rec):
            ctx.free(p_)


def test_group_render_rejects_what_it_cannot_read(pkg):
    """A request with a null taper / colour map is an invalid argument, not a crash - also once a plan is cached."""
    import ctypes as C
    L = pkg.Library.get().L
    g = pkg.Group([0])
    win, weight = pyoracle.window("hann", 64)
    lut = np.zeros((4, 3), np.uint8)
    data = np.zeros(2 * 64 * 8, np.uint8)
    g.render("CU8", data, 64, win, 1.0 / weight, 6.0, 30.0, lut, 8)
    _Request, _Reply = pkg.binding._Request, pkg.binding._Reply
    req = _Request(2, 64, 0, 0, 4, 0, 1.0 / weight, 6.0, 30.0, None, None)
    rep = _Reply(None, None, None, None, None, None, None)
    for mode in (0, 1):
        assert L.sp_group_render_ex(g.h, C.byref(req), data.ctypes.data_as(C.c_void_p), data.size, 8, C.byref(rep), mode) == -1
    assert L.sp_group_render_ex(g.h, C.byref(req), data.ctypes.data_as(C.c_void_p), data.size, 8, C.byref(rep), 7) == -1
    g.close()


SEEDED = [
    # fmt, log2 samples, n, width, window, gain, range, channelMode, waterfall
    ("CF32", 18, 1024, 256, "blackmanHarris", 6, 30, False, False),
    ("CF32", 17, 1024, 333, "hann", 10, 50, False, True),
    ("CS16", 18, 2048, 128, "hann", 6, 30, False, False),
    ("CU8", 18, 1024, 1000, "blackmanHarris", 6, 30, True, False),
    ("CS12", 17, 8192, 64, "blackmanHarris", 6, 30, False, False),
    ("CU8", 16, 512, 2048, "hann", 6, 30, False, False),
    ("CS8", 16, 64, 700, "hamming", 0, 60, False, True),
    ("CU16", 16, 128, 515, "bartlett", 20, 90, True, True),
    ("CF64", 15, 4096, 40, "blackman", 6, 30, False, False),
    ("CS32", 15, 256, 129, "rectangular", 3, 12, False, False),
]


@pytest.mark.parametrize("case", SEEDED, ids=lambda c: "%s_n%d_w%d" % (c[0], c[2], c[3]))
def test_seeded_inputs_against_oracle(pkg, ctx, golden, case):
    fmt, lg, n, width, wname, gain, rng, ch, wf = case
    gen = {"kind": "trinoise", "seed": 99 + n + width, "step": 9173, "gshift": 12, "amp": 0.45, "namp": 0.03}
    data = siggen.generate(fmt, gen, 1 << lg)
    win, weight = pyoracle.window(wname, n)
    lut = golden.lut("viridis", force_ends=True)
    want = pyoracle.render(fmt, data, n, win, 1.0 / weight, gain, rng, lut, width, ch, wf)
    got = ctx.render(fmt, data, n, win, 1.0 / weight, gain, rng, lut, width, ch, wf)
    for k in ("rgba", "gauge_mins", "gauge_maxs", "gauge_amps"):
        assert np.array_equal(got[k], want[k]), k
    assert np.array_equal(got["c_hist"].astype(np.int64), want["c_hist"])
    assert np.array_equal(got["cB_hist"].astype(np.int64), want["cB_hist"])
    assert np.float64(got["dBfs_min"]).view(np.uint64) == np.float64(want["dBfs_min"]).view(np.uint64)
    assert np.float64(got["dBfs_max"]).view(np.uint64) == np.float64(want["dBfs_max"]).view(np.uint64)


def test_device_synth_matches_cpu_generator(pkg, ctx):
    gen = dict(seed=0x5EED0001, step=7321, gshift=11, amp=0.5, namp=0.02)
    for fmt in pkg.FORMATS:
        sw = siggen.SAMPLE_WIDTH[fmt]
        count, t0 = 5000, 123456
        d = ctx.alloc(count * sw)
        ctx.synth_trinoise(d, fmt, t0, count, gen["seed"], gen["step"], gen["gshift"], gen["amp"], gen["namp"])
        ctx.synchronize()
        got = ctx.download(d, count * sw)
        ctx.free(d)
        want = siggen.generate(fmt, dict(kind="trinoise", **gen), count, t0)
        assert np.array_equal(got, want), fmt


def test_full_size_config2_sampled_frames(pkg, ctx, golden):
    """BASELINE config 2 at full size (16 MSample cf32, n=1024, Blackman-Harris, viridis) on device-generated input:
    size-independent checks (histogram totals, alpha channel) plus bit-exact comparison of sampled frames with the oracle."""
    n, S, fmt = 1024, 1 << 24, "CF32"
    W = S // n
    gen = dict(seed=0x5EED0001, step=7321, gshift=11, amp=0.5, namp=0.02)
    win, weight = pyoracle.window("blackmanHarris", n)
    # an injective LUT so that the image's colour counts can be compared with c_hist (viridis repeats two colours)
    i = np.arange(256)
    lut = np.stack([i, 255 - i, (i * 7) & 255], axis=1).astype(np.uint8)
    lut[0], lut[-1] = (0, 0, 0), (255, 255, 255)
    d_in = ctx.alloc(S * 8)
    ctx.synth_trinoise(d_in, fmt, 0, S, gen["seed"], gen["step"], gen["gshift"], gen["amp"], gen["namp"])
    plan = ctx.plan(fmt, n, win, 1.0 / weight, 6.0, 30.0, lut)
    sizes = [4 * W * n, W, W, W, 8 * 256, 8000, 16]
    ptrs = [ctx.alloc(s) for s in sizes]
    for p, s in zip(ptrs, sizes):
        ctx.memset(p, 0, s)
    plan.execute(d_in, S * 8, W, *ptrs)
    ctx.synchronize()
    rgba = ctx.download(ptrs[0], sizes[0]).reshape(n, W, 4)
    c_hist = ctx.download(ptrs[4], 8 * 256, np.uint64)
    cb_hist = ctx.download(ptrs[5], 8000, np.uint64)
    gmin, gmax, gamp = (ctx.download(ptrs[i], W) for i in (1, 2, 3))
    assert int(c_hist.sum()) == W * n
    assert int(cb_hist.sum()) <= W * n
    assert (rgba[:, :, 3] == 255).all()
    # colour histogram of the image equals c_hist (a checksum of checksums)
    packed = rgba.reshape(-1, 4).view(np.uint32).reshape(-1)
    lut32 = np.concatenate([lut, np.full((256, 1), 255, np.uint8)], axis=1).view(np.uint32).reshape(-1)
    assert len(np.unique(lut32)) == 256
    order = np.argsort(lut32)
    idx = order[np.searchsorted(lut32[order], packed)]
    assert np.array_equal(np.bincount(idx, minlength=256).astype(np.uint64), c_hist)
    rs = np.random.RandomState(7)
    for x in list(rs.randint(0, W, size=24)) + [0, W - 1]:
        frame = siggen.generate(fmt, dict(kind="trinoise", **gen), n, t0=int(x) * n)   # stride == n exactly
        o = pyoracle.render(fmt, frame, n, win, 1.0 / weight, 6.0, 30.0, lut, 1)
        assert np.array_equal(rgba[:, x, :].reshape(-1), o["rgba"]), x
        assert gmin[x] == o["gauge_mins"][0] and gmax[x] == o["gauge_maxs"][0] and gamp[x] == o["gauge_amps"][0], x
    for p in ptrs + [d_in]:
        ctx.free(p)
    plan.close()


def _random_cases(count, seed):
    rs = np.random.RandomState(seed)
    fmts = ["CU4", "CS4", "CU8", "CS8", "CU12", "CS12", "CU16", "CS16", "CU32", "CS32", "CU64", "CS64", "CF32", "CF64"]
    wins = ["rectangular", "bartlett", "hamming", "hann", "blackman", "blackmanHarris"]
    out = []
    for i in range(count):
        fmt = fmts[rs.randint(len(fmts))]
        n = 1 << rs.randint(1, 14)                      # 2 .. 8192 (scratch kernel below 64)
        frames = int(rs.randint(1, 80 if n >= 2048 else 300))
        mode = rs.randint(4)                            # hop = n, fractional, overlap, sparse
        if mode == 0:
            samples = n * frames
        elif mode == 1:
            samples = n * frames + int(rs.randint(1, n))
        elif mode == 2:
            samples = n + (frames - 1) * max(1, n // int(rs.randint(2, 9))) + int(rs.randint(0, 3))
        else:
            samples = n * frames * int(rs.randint(2, 5)) + int(rs.randint(0, 7))
        samples = min(samples, 1 << 19)
        if samples < n:
            samples = n
        lut_len = int(rs.choice([2, 3, 17, 64, 255, 256, 256, 256, 300]))
        out.append(dict(fmt=fmt, n=n, width=frames, samples=samples, win=wins[rs.randint(len(wins))], gain=float(rs.randint(-10, 60)),
                        rng=float(rs.choice([6, 12, 30, 30, 45.5, 90, 120])), ch=bool(rs.randint(2)), wf=bool(rs.randint(2)), lut_len=lut_len,
                        seed=int(rs.randint(1 << 30)), amp=float(rs.choice([0.05, 0.5, 0.9])), kind=str(rs.choice(["trinoise", "trinoise", "bytes"]))))
    return out


@pytest.mark.parametrize("case", _random_cases(60, 20261002), ids=lambda c: "%s_n%d_w%d_%s" % (c["fmt"], c["n"], c["width"], "lr" if c["ch"] else "iq"))
def test_random_requests_against_oracle(pkg, ctx, case):
    """Seeded random requests (all formats, n = 2 .. 8192, every stride regime, both layouts, L/R split, odd LUT lengths)
    through sp_render against the C oracle: every output bit-exact."""
    c = case
    kind = c["kind"] if not c["fmt"].startswith("CF") else "trinoise"
    gen = {"kind": kind, "seed": c["seed"], "step": 4099, "gshift": 9, "amp": c["amp"], "namp": 0.02}
    data = siggen.generate(c["fmt"], gen, c["samples"])
    win, weight = pyoracle.window(c["win"], c["n"])
    i = np.arange(c["lut_len"])
    lut = np.stack([(i * 5) & 255, (i * 11 + 3) & 255, (255 - i) & 255], axis=1).astype(np.uint8)
    want = pyoracle.render(c["fmt"], data, c["n"], win, 1.0 / weight, c["gain"], c["rng"], lut, c["width"], c["ch"], c["wf"])
    got = ctx.render(c["fmt"], data, c["n"], win, 1.0 / weight, c["gain"], c["rng"], lut, c["width"], c["ch"], c["wf"])
    for k in ("rgba", "gauge_mins", "gauge_maxs", "gauge_amps"):
        assert np.array_equal(got[k], want[k]), k
    assert np.array_equal(got["c_hist"].astype(np.int64), want["c_hist"])
    assert np.array_equal(got["cB_hist"].astype(np.int64), want["cB_hist"])
    for k in ("dBfs_min", "dBfs_max"):
        a, b = np.float64(got[k]), np.float64(want[k])
        assert (a != a and b != b) or a.view(np.uint64) == b.view(np.uint64), k


@pytest.mark.parametrize("cfg", [
    # name, format, log2 samples, n, frames (None = S/n), window
    ("config3_cs16_n2048", "CS16", 25, 2048, None, "hann"),
    ("config4_slice_cu8_n1024", "CU8", 26, 1024, None, "blackmanHarris"),
    ("config5_cs12_n8192_zoom8", "CS12", 23, 8192, (1 << 23) // 8192 * 8, "blackmanHarris"),
    # BASELINE configs 3 and 5 at full size (1 GiB in / 1 GiB out; 192 MiB in / 2 GiB out)
    ("config3_full_cs16_n2048", "CS16", 28, 2048, None, "hann"),
    ("config5_full_cs12_n8192_zoom8", "CS12", 26, 8192, (1 << 26) // 8192 * 8, "blackmanHarris"),
    # one full slice of BASELINE config 4: 2^28 cu8 samples (1/8 of the 2 GSample capture), 262 144 frames, 1 GiB of image
    ("config4_full_slice_cu8_n1024", "CU8", 28, 1024, None, "blackmanHarris"),
], ids=lambda c: c[0])
def test_baseline_config_shapes_sampled_frames(pkg, ctx, golden, cfg):
    """The other BASELINE.json configurations (scaled down and at full size: format, n, hop / 8x overlap, window) on
    device-generated input: histogram totals plus bit-exact comparison of sampled frames (image column, gauges) with the
    oracle, which renders each sampled frame from its own n samples (frames are independent given their start)."""
    name, fmt, lg, n, frames, wname = cfg
    S = 1 << lg
    sw = siggen.SAMPLE_WIDTH[fmt]
    W = frames or S // n
    gen = dict(seed=0x5EED0001, step=7321, gshift=11, amp=0.5, namp=0.02)
    win, weight = pyoracle.window(wname, n)
    lut = golden.lut("cube1", force_ends=True)
    d_in = ctx.alloc(S * sw)
    ctx.synth_trinoise(d_in, fmt, 0, S, gen["seed"], gen["step"], gen["gshift"], gen["amp"], gen["namp"])
    plan = ctx.plan(fmt, n, win, 1.0 / weight, 6.0, 30.0, lut)
    sizes = [4 * W * n, W, W, W, 8 * 256, 8000, 16]
    ptrs = [ctx.alloc(s) for s in sizes]
    for p, s in zip(ptrs, sizes):
        ctx.memset(p, 0, s)
    plan.execute(d_in, S * sw, W, *ptrs)
    ctx.synchronize()
    rgba = ctx.download(ptrs[0], sizes[0]).reshape(n, W, 4)
    c_hist = ctx.download(ptrs[4], 8 * 256, np.uint64)
    gmin, gmax, gamp = (ctx.download(ptrs[i], W) for i in (1, 2, 3))
    assert int(c_hist.sum()) == W * n
    assert (rgba[:, ::97, 3] == 255).all()
    stride = (S - n) / (W - 1)
    rs = np.random.RandomState(11)
    for x in list(rs.randint(0, W, size=10)) + [0, W - 1]:
        start = int(0.5 + stride * int(x))                      # ~~(0.5 + stride * x), positive here
        frame = siggen.generate(fmt, dict(kind="trinoise", **gen), n, t0=start)
        o = pyoracle.render(fmt, frame, n, win, 1.0 / weight, 6.0, 30.0, lut, 1)
        assert np.array_equal(rgba[:, x, :].reshape(-1), o["rgba"]), (name, x)
        assert gmin[x] == o["gauge_mins"][0] and gmax[x] == o["gauge_maxs"][0] and gamp[x] == o["gauge_amps"][0], (name, x)
    for p in ptrs + [d_in]:
        ctx.free(p)
    plan.close()


def test_pixels_straddling_every_index_edge(pkg, ctx):
    """k_frames decides colour index and centi-bel bin from an f32 value and sends only the lanes within a proven error margin of
    a step to the exact edge tables (sp_host.cpp).  This test puts |X|^2 onto both sides of EVERY step of both scales, a few ulps
    apart: a frame whose only non-zero sample is (x, 0) under a rectangular taper has |X|^2 = x*x in all its bins, exactly, and
    x runs over the doubles around the square root of every edge.  The edges are located with the oracle's own log10."""
    import ctypes
    n, fmt = 64, "CF64"
    gain, rng, L = 6.0, 30.0, 256
    block_norm = 1.0 / n
    log10 = pyoracle.lib().spo_log10
    bndb = 10 * log10(block_norm)
    cmax, color_norm = L - 1, L / -rng

    def gray(a2):
        u = cmax - (5 * log10(a2) + bndb + gain) * color_norm
        return int(0.5 + (0 if u < 0 else cmax if u > cmax else u))

    def cbin(a2):
        v = 0.5 + ((5 * log10(a2) + bndb + gain) - gain) * -10
        return min(int(v), 999) if v > -1 else -1 if int(v) < 0 else 0

    def edges(f, lo, hi):
        """smallest doubles in [lo, hi] at which f changes, by bisection over bit patterns (f is monotone there)"""
        out = []
        a, b = np.float64(lo).view(np.uint64), np.float64(hi).view(np.uint64)

        def rec(a, fa, b, fb):
            if fa == fb:
                return
            if b - a == 1:
                out.append(np.uint64(b).view(np.float64))
                return
            m = a + (b - a) // 2
            fm = f(float(np.uint64(m).view(np.float64)))
            rec(a, fa, m, fm)
            rec(m, fm, b, fb)
        rec(int(a), f(float(lo)), int(b), f(float(hi)))
        return out

    es = edges(gray, 1e-18, 1e5) + edges(cbin, 1e-18, 1e5)
    assert len(es) > 1200
    xs = []
    for e in es:
        r = np.sqrt(np.float64(e))
        for k in range(-2, 3):
            xs.append(np.uint64(int(r.view(np.uint64)) + k).view(np.float64))
    xs = np.array(xs, dtype=np.float64)
    W = len(xs)
    cap = np.zeros((W, n, 2), dtype=np.float64)
    cap[:, 0, 0] = xs
    data = cap.reshape(-1).view(np.uint8)
    win = np.ones(n, dtype=np.float64)
    i = np.arange(L)
    lut = np.stack([i, 255 - i, (i * 7) & 255], axis=1).astype(np.uint8)
    want = pyoracle.render(fmt, data, n, win, block_norm, gain, rng, lut, W)
    for kernel in ("frames", "scratch"):
        got = _plan_render(pkg, ctx, kernel, fmt, data, n, win, block_norm, gain, rng, lut, W, False, False)
        assert got is not None and got["kernel"] == {"frames": "frames", "scratch": "scratch_radix2"}[kernel]
        assert np.array_equal(got["rgba"], want["rgba"]), kernel
        assert np.array_equal(got["c_hist"].astype(np.int64), want["c_hist"]), kernel
        assert np.array_equal(got["cB_hist"].astype(np.int64), want["cB_hist"]), kernel
        assert np.float64(got["dBfs_min"]).view(np.uint64) == np.float64(want["dBfs_min"]).view(np.uint64), kernel
    # the squares really fall on both sides of the steps: both colours of most edges occur
    assert len(np.unique(want["rgba"].reshape(-1, 4)[:, 0])) > 250


@pytest.mark.parametrize("fmt,lg,n,width,wf", [("CF32", 21, 1024, 2048, False), ("CF32", 21, 1024, 2000, True), ("CU8", 23, 512, 16384, False),
                                                ("CS16", 22, 2048, 3001, False)],
                         ids=["cf32_hop", "cf32_frac_wf", "cu8_8chunks", "cs16_overlap"])
def test_large_host_requests_render_in_overlapped_chunks(pkg, ctx, fmt, lg, n, width, wf):
    """sp_render walks a large request in chunks of frames (samples in, render, image band out on three streams).  Whole
    reply against the oracle: chunk seams (frame groups, copied byte ranges, column / row bands) must not show."""
    S = 1 << lg
    data = siggen.generate(fmt, {"kind": "trinoise", "seed": 321, "step": 7321, "gshift": 11, "amp": 0.5, "namp": 0.02}, S)
    win, weight = pyoracle.window("hann", n)
    i = np.arange(256)
    lut = np.stack([i, 255 - i, (i * 7) & 255], axis=1).astype(np.uint8)
    want = pyoracle.render(fmt, data, n, win, 1.0 / weight, 6.0, 30.0, lut, width, False, wf)
    for _ in range(2):                                   # twice: the second request reuses streams, events and staging buffers
        got = ctx.render(fmt, data, n, win, 1.0 / weight, 6.0, 30.0, lut, width, False, wf)
        _assert_same(got, want)


SPARSE = [(f, 256, 18, 131, False) for f in ("CU4", "CS4", "CU8", "CS8", "CU12", "CS12", "CU16", "CS16", "CU32", "CS32", "CU64", "CS64", "CF32", "CF64")] + [
    ("CS12", 64, 16, 300, True),          # short frames, many rows per pitched copy
    ("CU8", 512, 0, 200, False),          # an integer stride: 3 * n (the length is made to fit below)
    ("CF32", 1024, 22, 2048, False),      # the reference's interactive shape, scaled: four chunks, fractional stride (drift inside a chunk)
    ("CS16", 1024, 24, 4096, True),       # ... waterfall layout, hop 4 n
    ("CS12", 8192, 23, 96, False),        # a frame = the whole workgroup
    ("CF32", 128, 21, 1500, False),       # stride ~ 11 n
]


@pytest.mark.parametrize("fmt,n,lg,width,wf", SPARSE, ids=lambda v: str(v))
def test_sparse_requests_upload_only_the_frames_they_read(pkg, ctx, monkeypatch, fmt, n, lg, width, wf):
    """stride > n (lib/worker.js:50, 70-75: the reference's loop skips the samples between frames): sp_render copies the frames as rows
    of pitched copies into a packed device buffer and renders from there.  Bit-exact against the oracle, identical to the contiguous
    upload (SPECTROPLOT_HIP_NO_PACKED_UPLOAD), and the bytes over the link are the frames' own plus the rows' widening (at most half)."""
    S = (1 << lg) if lg else n + (width - 1) * 3 * n
    data = siggen.generate(fmt, {"kind": "trinoise", "seed": 77 + n, "step": 7321, "gshift": 9, "amp": 0.5, "namp": 0.02}, S)
    win, weight = pyoracle.window("blackmanHarris", n)
    i = np.arange(256)
    lut = np.stack([i, 255 - i, (i * 7) & 255], axis=1).astype(np.uint8)
    want = pyoracle.render(fmt, data, n, win, 1.0 / weight, 6.0, 30.0, lut, width, False, wf)
    sw = pkg.parse_format(fmt)[1]
    for _ in range(2):
        got = ctx.render(fmt, data, n, win, 1.0 / weight, 6.0, 30.0, lut, width, False, wf)
        _assert_same(got, want)
        sent = ctx.last_upload_bytes()
        assert width * n * sw <= sent <= width * n * sw * 3 // 2 and sent <= data.size * 3 // 4, (sent, width * n * sw, data.size)
    monkeypatch.setenv("SPECTROPLOT_HIP_NO_PACKED_UPLOAD", "1")
    got = ctx.render(fmt, data, n, win, 1.0 / weight, 6.0, 30.0, lut, width, False, wf)
    _assert_same(got, want)
    assert ctx.last_upload_bytes() == data.size


FROM_HOST = [
    # fmt, n, log2 samples (0: 3 n hop), width, waterfall - what the path takes
    ("CU8", 1024, 23, 8192, False),       # contiguous upload in four chunks under the renders (16 MiB in)
    ("CS16", 2048, 22, 3001, True),       # overlapping frames, waterfall, four chunks
    ("CF32", 1024, 22, 2048, False),      # sparse: only the frames' samples travel, four packed chunks
    ("CS12", 256, 18, 131, False),        # sparse, one packed chunk
    ("CU8", 512, 0, 200, False),          # sparse, integer stride
    ("CF32", 64, 12, 7, False),           # a tiny request: one plain copy
]


@pytest.mark.parametrize("fmt,n,lg,width,wf", FROM_HOST, ids=lambda v: str(v))
def test_plan_execute_from_host_matches_the_oracle(pkg, ctx, fmt, n, lg, width, wf):
    """sp_plan_execute_from_host (what a group member does with its slice): the capture in host memory, uploaded in chunks of frames
    under the renders - a sparse request only the samples its frames read - and every output left in HBM.  Queued twice back to back
    WITHOUT a synchronisation in between (the second request's first copy must wait for the first request's kernels: they share the
    context's staging buffer), then everything is compared with the oracle."""
    S = (1 << lg) if lg else n + (width - 1) * 3 * n
    data = siggen.generate(fmt, {"kind": "trinoise", "seed": 99 + n, "step": 7321, "gshift": 9, "amp": 0.5, "namp": 0.02}, S)
    other = siggen.generate(fmt, {"kind": "trinoise", "seed": 1234, "step": 4099, "gshift": 9, "amp": 0.3, "namp": 0.05}, S)
    win, weight = pyoracle.window("blackmanHarris", n)
    i = np.arange(256)
    lut = np.stack([i, 255 - i, (i * 7) & 255], axis=1).astype(np.uint8)
    want = pyoracle.render(fmt, data, n, win, 1.0 / weight, 6.0, 30.0, lut, width, False, wf)
    plan = ctx.plan(fmt, n, win, 1.0 / weight, 6.0, 30.0, lut, False, wf)
    W = width
    sizes = [4 * W * n, W, W, W, 8 * 256, 8000, 16]
    ptrs = [ctx.alloc(max(s_, 16)) for s_ in sizes]
    for p_, s_ in zip(ptrs, sizes):
        ctx.memset(p_, 0xA5, max(s_, 16))              # (dirty: the kernel clears the reply itself)
    keep = [plan.execute_from_host(other, W, *ptrs), plan.execute_from_host(data, W, *ptrs)]
    ctx.synchronize()
    sw = pkg.parse_format(fmt)[1]
    sent = ctx.last_upload_bytes()
    stride = (S - n) / (W - 1)
    if stride > n and W >= 2:
        assert W * n * sw <= sent <= W * n * sw * 3 // 2 and sent <= data.size * 3 // 4, (sent, W * n * sw, data.size)
    else:
        assert sent == data.size
    got = {"rgba": ctx.download(ptrs[0], sizes[0]), "gauge_mins": ctx.download(ptrs[1], W), "gauge_maxs": ctx.download(ptrs[2], W),
           "gauge_amps": ctx.download(ptrs[3], W), "c_hist": ctx.download(ptrs[4], 8 * 256, np.uint64),
           "cB_hist": ctx.download(ptrs[5], 8000, np.uint64)}
    mm = ctx.download(ptrs[6], 16, np.float64)
    got["dBfs_min"], got["dBfs_max"] = float(mm[0]), float(mm[1])
    _assert_same(got, want)
    del keep
    for p_ in ptrs:
        ctx.free(p_)
    plan.close()


def test_nonfinite_taper_is_exact(pkg, ctx):
    """A caller-supplied taper may hold infinities or NaN (options.windowF is any function).  Inf * 1 + Inf * 0 is NaN in the
    reference's first butterfly, so the kernels that skip the products of (1, 0) butterflies must not serve such a plan."""
    n, W, fmt = 1024, 40, "CS16"
    data = siggen.generate(fmt, {"kind": "trinoise", "seed": 5, "step": 4099, "gshift": 9, "amp": 0.5, "namp": 0.02}, n * W)
    win, weight = pyoracle.window("hann", n)
    i = np.arange(256)
    lut = np.stack([i, 255 - i, (i * 7) & 255], axis=1).astype(np.uint8)
    for bad in ([(n // 2, np.inf)], [(3, np.nan)], [(0, -np.inf), (n - 1, np.inf)]):
        w = win.copy()
        for pos, v in bad:
            w[pos] = v
        plan = ctx.plan(fmt, n, w, 1.0 / weight, 6.0, 30.0, lut)
        assert plan.kernel_name() == "scratch_radix2"
        plan.close()
        want = pyoracle.render(fmt, data, n, w, 1.0 / weight, 6.0, 30.0, lut, W)
        got = ctx.render(fmt, data, n, w, 1.0 / weight, 6.0, 30.0, lut, W)
        _assert_same(got, want)


def _assert_same(got, want):
    for k in ("rgba", "gauge_mins", "gauge_maxs", "gauge_amps"):
        assert np.array_equal(got[k], want[k]), k
    assert np.array_equal(got["c_hist"].astype(np.int64), want["c_hist"])
    assert np.array_equal(got["cB_hist"].astype(np.int64), want["cB_hist"])
    for k in ("dBfs_min", "dBfs_max"):
        a, b = np.float64(got[k]), np.float64(want[k])
        assert (a != a and b != b) or a.view(np.uint64) == b.view(np.uint64), k


@pytest.mark.parametrize("gain,rng,bn_scale", [(6.0, 30.0, 1.0), (2500.0, 30.0, 1.0), (-2500.0, 30.0, 1.0), (6.0, 3000.0, 1.0),
                                               (6.0, 30.0, 1e-140), (6.0, 30.0, 1e140), (40.0, 0.75, 1.0)],
                         ids=["plain", "gain+2500", "gain-2500", "range3000", "norm1e-140", "norm1e140", "range0.75"])
def test_extreme_gain_range_norm(pkg, ctx, gain, rng, bn_scale):
    """Requests whose colour / centi-bel edges leave the f32 range (the LDS kernel's first guess) or whose colour steps are
    finer than its guess tolerates must still be bit-exact (they run on the scratch kernel)."""
    n, W, fmt = 1024, 96, "CF32"
    data = siggen.generate(fmt, {"kind": "trinoise", "seed": 99, "step": 4099, "gshift": 9, "amp": 0.5, "namp": 0.02}, n * W)
    win, weight = pyoracle.window("hann", n)
    i = np.arange(256)
    lut = np.stack([i, 255 - i, (i * 7) & 255], axis=1).astype(np.uint8)
    bn = bn_scale / weight
    want = pyoracle.render(fmt, data, n, win, bn, gain, rng, lut, W)
    got = ctx.render(fmt, data, n, win, bn, gain, rng, lut, W)
    _assert_same(got, want)


@pytest.mark.parametrize("fmt", ["CF32", "CF64", "CS16"])
def test_silence_and_nonfinite_samples(pkg, ctx, fmt):
    """Frames of exact zeros (|X|^2 = 0: -inf dB, centi-bel key 0), frames with infinities and NaNs (float formats) and ordinary
    frames in one capture: the rare-path branches of the histogram code against the oracle."""
    n, W = 1024, 64
    data = siggen.generate(fmt, {"kind": "trinoise", "seed": 5, "step": 4099, "gshift": 9, "amp": 0.5, "namp": 0.02}, n * W)
    sw = data.size // (n * W)
    data = data.copy()
    data[8 * n * sw:24 * n * sw] = 0                                  # 16 silent frames
    if fmt.startswith("CF"):
        v = data.view(np.float32 if fmt == "CF32" else np.float64)
        per = 2 * n
        v[30 * per + 17] = np.inf
        v[31 * per + 400] = -np.inf
        v[33 * per + 1] = np.nan
        v[40 * per:41 * per] = 1e30 if fmt == "CF32" else 1e200      # overflow of |X|^2 in f32 / towards inf
        v[41 * per:42 * per] = 1e-30 if fmt == "CF32" else 1e-200    # underflow of the f32 first guess
    win, weight = pyoracle.window("blackmanHarris", n)
    i = np.arange(256)
    lut = np.stack([i, 255 - i, (i * 7) & 255], axis=1).astype(np.uint8)
    for ch in (False, True):
        want = pyoracle.render(fmt, data, n, win, 1.0 / weight, 6.0, 30.0, lut, W, ch)
        got = ctx.render(fmt, data, n, win, 1.0 / weight, 6.0, 30.0, lut, W, ch)
        _assert_same(got, want)


def test_reply_is_overwritten_not_accumulated(pkg, ctx):
    """Two executions of one plan into the same reply buffers leave the counts of ONE request (include/spectroplot_hip.h),
    for both kernels, and a zero-width request clears them."""
    n, W, fmt = 1024, 40, "CU8"
    data = siggen.generate(fmt, {"kind": "trinoise", "seed": 3, "step": 4099, "gshift": 9, "amp": 0.5, "namp": 0.02}, n * W)
    win, weight = pyoracle.window("hann", n)
    i = np.arange(256)
    lut = np.stack([i, 255 - i, (i * 7) & 255], axis=1).astype(np.uint8)
    want = pyoracle.render(fmt, data, n, win, 1.0 / weight, 6.0, 30.0, lut, W)
    for kernel in ("frames", "scratch"):
        plan = ctx.plan(fmt, n, win, 1.0 / weight, 6.0, 30.0, lut)
        plan.force_kernel(kernel)
        d_in = ctx.alloc(data.size)
        ctx.upload(d_in, data)
        sizes = [4 * W * n, W, W, W, 8 * 256, 8000, 16]
        ptrs = [ctx.alloc(s) for s in sizes]
        for p, s in zip(ptrs, sizes):
            ctx.memset(p, 0xA5, s)                                     # stale contents must not survive
        for _ in range(3):
            plan.execute(d_in, data.size, W, *ptrs)
        ctx.synchronize()
        assert np.array_equal(ctx.download(ptrs[4], 8 * 256, np.uint64).astype(np.int64), want["c_hist"]), kernel
        assert np.array_equal(ctx.download(ptrs[5], 8000, np.uint64).astype(np.int64), want["cB_hist"]), kernel
        mm = ctx.download(ptrs[6], 16, np.float64)
        assert mm[0] == want["dBfs_min"] and mm[1] == want["dBfs_max"], kernel
        plan.execute(d_in, data.size, 0, *ptrs)
        ctx.synchronize()
        assert not ctx.download(ptrs[4], 8 * 256, np.uint64).any() and not ctx.download(ptrs[5], 8000, np.uint64).any(), kernel
        mm = ctx.download(ptrs[6], 16, np.float64)
        assert mm[0] == 0.0 and mm[1] == -200.0, kernel
        for p in ptrs + [d_in]:
            ctx.free(p)
        plan.close()


def test_merge_replies_matches_callers_merge(pkg, ctx):
    """sp_merge_replies against the reference caller's merge (element-wise histogram sums, running min / max of the dBfs range,
    lib/spectroplot.js:1229-1238) on random records."""
    rs = np.random.RandomState(11)
    for count, L in ((1, 256), (2, 256), (8, 64), (5, 300)):
        P = L + 1000 + 2
        rec = np.zeros((count, P), dtype=np.uint64)
        rec[:, :L + 1000] = rs.randint(0, 1 << 40, size=(count, L + 1000)).astype(np.uint64)
        mins = -rs.rand(count) * 150.0
        maxs = -rs.rand(count) * 100.0 - 1.0
        mins[0], maxs[-1] = 0.0, -200.0                        # the worker's initial values are legal replies
        rec[:, P - 2] = mins.view(np.uint64)
        rec[:, P - 1] = maxs.view(np.uint64)
        d_rec, d_out = ctx.alloc(rec.nbytes), ctx.alloc(8 * P)
        ctx.upload(d_rec, rec.view(np.uint8).reshape(-1))
        ctx.memset(d_out, 0xEE, 8 * P)
        ctx.merge_replies(d_rec, count, L, d_out, d_out + 8 * L, d_out + 8 * (L + 1000))
        ctx.synchronize()
        out = ctx.download(d_out, 8 * P, np.uint64)
        assert np.array_equal(out[:L + 1000], rec[:, :L + 1000].sum(axis=0))
        mm = out[L + 1000:].view(np.float64)
        assert mm[0] == mins.min() and mm[1] == maxs.max()
        ctx.free(d_rec)
        ctx.free(d_out)


@pytest.mark.parametrize("fmt", ["CU12", "CS12", "CU4", "CS4"])
@pytest.mark.parametrize("n,frames,extra", [(64, 40, 0), (256, 33, 0), (1024, 70, 0), (1024, 70, 5), (2048, 9, 0), (1024, 1, 0), (512, 2, 0)])
def test_three_byte_samples_to_the_last_byte(pkg, ctx, fmt, n, frames, extra):
    """12-bit formats are fetched as unaligned dwords; a frame that ends with the buffer is fetched one byte low and shifted.
    Captures whose last frame ends exactly at the last byte (extra = 0), a few samples before it, single frames, and S == n."""
    samples = n * frames + extra
    data = siggen.generate(fmt, {"kind": "trinoise", "seed": 77 + n + frames, "step": 4099, "gshift": 9, "amp": 0.6, "namp": 0.02}, samples)
    assert data.size == (3 if "12" in fmt else 1) * samples
    win, weight = pyoracle.window("hann", n)
    i = np.arange(256)
    lut = np.stack([i, 255 - i, (i * 7) & 255], axis=1).astype(np.uint8)
    for width in sorted({frames, max(1, frames // 2), 2 * frames}):
        want = pyoracle.render(fmt, data, n, win, 1.0 / weight, 6.0, 30.0, lut, width)
        got = ctx.render(fmt, data, n, win, 1.0 / weight, 6.0, 30.0, lut, width)
        _assert_same(got, want)


@pytest.mark.parametrize("fmt", ["CU64", "CS64", "CF64"])
@pytest.mark.parametrize("n,frames", [(64, 300), (256, 150), (1024, 70), (2048, 40), (8192, 9)])
def test_sixteen_byte_samples(pkg, ctx, fmt, n, frames):
    """The 16-byte formats (the frame loop's generic loader: one 16-byte load per sample when the frame starts) at sizes of every
    synchronisation regime, I/Q and L/R, hop = n / overlapping / sparse, with frames that are silent, and for cf64 frames that hold
    infinities, NaNs and finite values from 2^1017 up."""
    samples = n * frames
    data = siggen.generate(fmt, {"kind": "trinoise", "seed": 1234 + n, "step": 4099, "gshift": 9, "amp": 0.6, "namp": 0.02}, samples).copy()
    assert data.size == 16 * samples
    data[2 * n * 16:4 * n * 16] = 0
    if fmt == "CF64":
        v = data.view(np.float64)
        per = 2 * n
        v[5 * per + 3] = np.inf
        v[6 * per + per - 1] = np.nan
        v[7 * per + 10] = -2.0 ** 1018
        v[8 * per:9 * per] = 1e200
    win, weight = pyoracle.window("blackmanHarris", n)
    i = np.arange(256)
    lut = np.stack([i, 255 - i, (i * 7) & 255], axis=1).astype(np.uint8)
    for width, ch in ((frames, False), (frames, True), (2 * frames - 1, False), (max(1, frames // 3), False)):
        want = pyoracle.render(fmt, data, n, win, 1.0 / weight, 6.0, 30.0, lut, width, ch)
        got = ctx.render(fmt, data, n, win, 1.0 / weight, 6.0, 30.0, lut, width, ch)
        _assert_same(got, want)


def test_full_size_repeatable(pkg, ctx):
    """Config 2 at full size, executed three times into separate buffers: every output byte identical between runs (a race
    between waves, tiles or accumulators would show up as a difference)."""
    n, S, fmt = 1024, 1 << 24, "CF32"
    W = S // n
    win, weight = pyoracle.window("blackmanHarris", n)
    i = np.arange(256)
    lut = np.stack([i, 255 - i, (i * 7) & 255], axis=1).astype(np.uint8)
    d_in = ctx.alloc(S * 8)
    ctx.synth_trinoise(d_in, fmt, 0, S, 0x5EED0001, 7321, 11, 0.5, 0.02)
    plan = ctx.plan(fmt, n, win, 1.0 / weight, 6.0, 30.0, lut)
    sizes = [4 * W * n, W, W, W, 8 * 256, 8000, 16]
    runs = []
    for _ in range(3):
        ptrs = [ctx.alloc(s) for s in sizes]
        plan.execute(d_in, S * 8, W, *ptrs)
        ctx.synchronize()
        runs.append([ctx.download(p, s) for p, s in zip(ptrs, sizes)])
        for p in ptrs:
            ctx.free(p)
    for r in runs[1:]:
        for a, b in zip(runs[0], r):
            assert np.array_equal(a, b)
    ctx.free(d_in)
    plan.close()


@pytest.mark.parametrize("fmt,n,lg,ch", [("CS16", 2048, 24, False), ("CF32", 2048, 23, True), ("CU8", 4096, 24, False), ("CF32", 4096, 23, False),
                                         ("CS12", 8192, 24, False), ("CF32", 8192, 23, True)])
def test_frames_shared_by_several_waves_are_race_free(pkg, ctx, fmt, n, lg, ch):
    """n >= 2048: a frame's waves meet around the one re-distribution that crosses them and nowhere else (the first one - and at
    n = 8192 the second - stays inside a wave).  A launch that keeps every CU busy for dozens of groups, four times over, must give the
    same bytes every time, and the bytes of the portable kernel, which shares none of that synchronisation."""
    S = 1 << lg
    W = S // n
    sw = siggen.SAMPLE_WIDTH[fmt]
    win, weight = pyoracle.window("hann", n)
    i = np.arange(256)
    lut = np.stack([i, 255 - i, (i * 7) & 255], axis=1).astype(np.uint8)
    d_in = ctx.alloc(S * sw)
    ctx.synth_trinoise(d_in, fmt, 0, S, 0x5EED0002, 7321, 11, 0.5, 0.02)
    sizes = [4 * W * n, W, W, W, 8 * 256, 8000, 16]
    runs = []
    for kernel in ("scratch", "frames", "frames", "frames", "frames"):
        plan = ctx.plan(fmt, n, win, 1.0 / weight, 6.0, 30.0, lut, ch)
        plan.force_kernel(kernel)
        ptrs = [ctx.alloc(s) for s in sizes]
        plan.execute(d_in, S * sw, W, *ptrs)
        ctx.synchronize()
        assert plan.kernel_name() == ("scratch_radix2" if kernel == "scratch" else "frames")
        runs.append([ctx.download(p, s) for p, s in zip(ptrs, sizes)])
        for p in ptrs:
            ctx.free(p)
        plan.close()
    names = ["rgba", "gauge_mins", "gauge_maxs", "gauge_amps", "c_hist", "cB_hist", "dBfs"]
    for k, r in enumerate(runs[1:]):
        for name, a, b in zip(names, runs[0], r):
            assert np.array_equal(a, b), (name, k)
    ctx.free(d_in)


def test_requests_by_name_match_the_golden_vectors_and_reuse_their_plan(pkg, golden):
    """sp_render_named: the request given as option names, resolved inside the library with the reference's lookup rules and defaults
    (lib/utils.js:25-40, lib/spectroplot.js:238-264, 1113-1146; the keys are the ones tests/golden/parse.json recorded from the reference).
    Every golden worker case whose taper and colour map the reference ships is rendered by names only and compared with the real
    worker's reply; spellings that resolve to the same table entry give the same reply; a repeated request builds no new plan."""
    import json
    import os
    parse = json.load(open(os.path.join(goldenlib.GDIR, "parse.json")))
    win_hits = {e["key"]: e["hit"] for e in parse["lookups"]}
    cmap_hits = {e["key"]: e["hit"] for e in parse["clookups"]}
    ctx = pkg.Context(0)
    bad, ran = [], 0
    for c in golden.spec["worker_cases"]:
        e = golden.expected[c["name"]]
        if "reply" not in e or c["cmap"].startswith("custom:") or not c["force_ends"] or c["width"] == 0:
            continue
        # the table keys themselves: a bare 'blackman' is a prefix of the sorted table's 'blackmanHarrisWindow' too and resolves to that one
        r = ctx.render_named(c["format"], golden.input(c), c["n"], c["window"] + "Window", c["cmap"] + "_cmap", c["gain"], c["range"],
                             c["width"], c["channelMode"], c["waterfall"])
        bad += goldenlib.check_reply(r, e["reply"], c["name"] + " by name: ")
        ran += 1
    assert ran > 60 and not bad, (ran, bad[:20])

    # spellings: every recorded key against the array path with what the reference resolved it to
    c = golden.cases["cfg2_scaled"]
    data = golden.input(c)

    def by_arrays(window_hit, cmap_hit):
        wname = (window_hit or "blackmanHarrisWindow")[:-len("Window")]
        win, weight = pyoracle.window(wname, c["n"])
        lut = golden.lut((cmap_hit or "cube1_cmap")[:-len("_cmap")], force_ends=True)
        return ctx.render(c["format"], data, c["n"], win, 1.0 / weight, c["gain"], c["range"], lut, c["width"])

    def same(a, b):
        return all(np.array_equal(a[k], b[k]) for k in ("rgba", "gauge_mins", "gauge_maxs", "gauge_amps", "c_hist", "cB_hist")) \
            and a["dBfs_min"] == b["dBfs_min"] and a["dBfs_max"] == b["dBfs_max"]

    win_hits["blackman"] = "blackmanHarrisWindow"   # first prefix hit in the reference's (sorted) key order
    for key in ("hann", "Hann", "black", "blackman", "nosuch", "ha", "b", "rect", "HANNWINDOW"):
        got = ctx.render_named(c["format"], data, c["n"], key, "viridis", c["gain"], c["range"], c["width"])
        assert same(got, by_arrays(win_hits[key], "viridis_cmap")), "window key %r" % key
    for key in ("v", "VIRIDIS", "nosuch", "p", "pl", "h", "gist", "inferno_cmap", "cube1"):
        got = ctx.render_named(c["format"], data, c["n"], "blackmanHarris", key, c["gain"], c["range"], c["width"])
        assert len(got["c_hist"]) == len(golden.lut((cmap_hits[key] or "cube1_cmap")[:-5])), key
        assert same(got, by_arrays("blackmanHarrisWindow", cmap_hits[key])), "cmap key %r" % key

    # the plan stays while names and numbers repeat; a changed name or number builds one new plan
    ctx.render_named("cf32", data, c["n"], "hann", "v", 6.0, 30.0, c["width"])
    n0 = ctx.plan_creations()
    for _ in range(3):
        ctx.render_named("cf32", data, c["n"], "hann", "v", 6.0, 30.0, c["width"])
    assert ctx.plan_creations() == n0
    ctx.render_named("cf32", data, c["n"], "hann", "v", 7.0, 30.0, c["width"])
    assert ctx.plan_creations() == n0 + 1
    ctx.render_named("cf32", data, c["n"], "Hann", "v", 7.0, 30.0, c["width"])       # another spelling: re-resolved, one more plan at most
    assert ctx.plan_creations() <= n0 + 2
    ctx.close()
