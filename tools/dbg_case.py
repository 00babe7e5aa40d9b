import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import goldenlib
from __graft_entry__ import load_package
from oracle import pyoracle
pkg = load_package()
g = goldenlib.Golden()
ctx = pkg.Context(0)
for name in sys.argv[1:]:
    c = g.cases[name]
    data = g.input(c)
    win, weight = pyoracle.window(c["window"], c["n"])
    lut = g.lut(c)
    want = pyoracle.render(c["format"], data, c["n"], win, 1.0 / weight, c["gain"], c["range"], lut, c["width"], c["channelMode"], c["waterfall"], planes=True)
    got = ctx.render(c["format"], data, c["n"], win, 1.0 / weight, c["gain"], c["range"], lut, c["width"], c["channelMode"], c["waterfall"])
    n, W = c["n"], c["width"]
    a = got["rgba"].reshape(-1, 4); b = want["rgba"].reshape(-1, 4)
    diff = np.nonzero((a != b).any(axis=1))[0]
    print(name, "pixels differing:", len(diff), "of", len(a))
    for p in diff[:10]:
        y, x = divmod(int(p), W) if not c["waterfall"] else (None, None)
        i = (n // 2 - y) % n if y is not None else None
        print("  px", p, "x", x, "bin", i, "got", a[p], "want", b[p], "abs2", want["abs2"][x, i] if x is not None else None)
    print("  c_hist diff idx:", np.nonzero(got["c_hist"].astype(np.int64) != want["c_hist"])[0][:10], got["c_hist"][:3], want["c_hist"][:3], got["c_hist"][-2:], want["c_hist"][-2:])
