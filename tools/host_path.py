"""Host-buffer path (sp_render: what HipWorker calls): PCIe-inclusive time of a config-2-sized request through the C ABI, for four
kinds of host buffers: fresh pageable reply (what a naive caller does), reused pageable, and page-locked request / reply
(sp_host_alloc).  DESIGN.md section 8."""
import ctypes as C, sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package
import siggen
pkg = load_package()
from spectroplot_js_amd import binding as B
ctx = pkg.Context(0)
L = ctx.lib.L
# python3 tools/host_path.py [cfg2 | cfg1]   (cfg1: 1 MSample cu8, n = 512: the small-message case)
CFG = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
# sparse: the reference's interactive shape - config 2's capture at a screen-wide image (2048 frames, stride ~ 8 n): sp_render uploads the
#         frames' own samples only; `sparse-off`: the same with SPECTROPLOT_HIP_NO_PACKED_UPLOAD=1 (the capture as it is)
if CFG == "sparse-off":
    os.environ["SPECTROPLOT_HIP_NO_PACKED_UPLOAD"] = "1"
n, fmt, W = (512, "CU8", 2048) if CFG == "cfg1" else (1024, "CF32", 2048) if CFG.startswith("sparse") else (1024, "CF32", 16384)
S = 1 << 24 if CFG.startswith("sparse") else W * n
data = siggen.generate(fmt, {"kind": "trinoise", "seed": 0x5EED0001, "step": 7321, "gshift": 11, "amp": 0.5, "namp": 0.02}, S)
win, weight = pkg.window("hann" if CFG == "cfg1" else "blackmanHarris", n)
i = np.arange(256)
lut = np.stack([i, 255 - i, (i * 7) & 255], axis=1).astype(np.uint8)
fid, _ = B.parse_format(fmt)
req, keep = B._make_request(fid, n, win, 1.0 / weight, 6.0, 30.0, lut, False, False)
L.sp_host_alloc.argtypes = [C.c_size_t, C.POINTER(C.c_void_p)]


def pinned(nbytes):
    p = C.c_void_p()
    assert L.sp_host_alloc(nbytes, C.byref(p)) == 0
    return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(nbytes,))


def run(inp, out_rgba, fresh_out, reps=10 if CFG == "cfg2" else 200):
    small = [np.zeros(W, np.uint8) for _ in range(3)] + [np.zeros(256, np.uint64), np.zeros(1000, np.uint64), np.zeros(2)]
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    ts = []
    for r in range(reps + 3):
        if fresh_out:
            out_rgba = np.empty(4 * W * n, np.uint8)
        rep = B._Reply(p(out_rgba), p(small[0]), p(small[1]), p(small[2]), p(small[3]), p(small[4]), p(small[5]))
        t0 = time.perf_counter()
        assert L.sp_render(ctx.h, C.byref(req), p(inp), inp.size, W, C.byref(rep)) == 0
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts[3:])) * 1e3, int(small[3].sum())


pin_in = pinned(data.size); pin_in[:] = data
pin_out = pinned(4 * W * n)
page_out = np.zeros(4 * W * n, np.uint8)
for name, inp, out, fresh in (("pageable request, fresh pageable reply", data, None, True), ("pageable request, reused pageable reply", data, page_out, False),
                              ("pageable request, page-locked reply", data, pin_out, False), ("page-locked request and reply", pin_in, pin_out, False)):
    ms, total = run(inp, out, fresh)
    assert total == W * n
    sent = ctx.last_upload_bytes()
    print("sp_render " + CFG + ", %-42s %6.3f ms per request = %5.1f GB/s over PCIe both ways (%.1f MiB of samples sent)" % (name + ":", ms, (sent + 4 * W * n) / ms / 1e6, sent / 2**20))
