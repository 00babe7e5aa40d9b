"""Host-buffer path (sp_render: what HipWorker calls): PCIe-inclusive time for a config-2-sized request, measured, for DESIGN.md section 5."""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package
import siggen
pkg = load_package()
ctx = pkg.Context(0)
n, fmt = 1024, "CF32"
W = 16384; S = W * n
data = siggen.generate(fmt, {"kind": "trinoise", "seed": 0x5EED0001, "step": 7321, "gshift": 11, "amp": 0.5, "namp": 0.02}, S)
win, weight = pkg.window("blackmanHarris", n)
i = np.arange(256)
lut = np.stack([i, 255 - i, (i * 7) & 255], axis=1).astype(np.uint8)
for _ in range(3):
    ctx.render(fmt, data, n, win, 1.0 / weight, 6.0, 30.0, lut, W)
ts = []
for _ in range(10):
    t0 = time.perf_counter()
    ctx.render(fmt, data, n, win, 1.0 / weight, 6.0, 30.0, lut, W)
    ts.append(time.perf_counter() - t0)
t = float(np.median(ts))
print("sp_render cfg2: %.2f ms per request (%d MiB in, %d MiB out): %.1f M frames/s, %.1f GB/s over PCIe both ways" %
      (t * 1e3, data.nbytes >> 20, (4 * W * n) >> 20, W / t / 1e6, (data.nbytes + 4 * W * n) / t / 1e9))
