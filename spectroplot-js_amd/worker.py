"""Python mirror of the reference's worker interface (same field names, same error behaviour), over the C ABI.

`HipWorker` has the shape of the object the reference injects through `options.workerOrUrl`
(lib/spectroplot.js:100-116): `postMessage(message)` in, `onmessage({'data': reply})` out, exactly one reply per
render request, in order.  Messages without a `buffer` are ignored like the reference's guard (lib/worker.js:158-163).
The production host layer is the JavaScript `HipWorker` (js/hip_worker.js); this mirror exists so that the parity
tests can drive the same ABI from pytest.
"""
import numpy as np

from . import binding


def _lut_bytes(cmap):
    """cmap entries -> uint8, with Uint8ClampedArray store semantics for non-integer entries (worker.js:118-121)."""
    a = np.asarray(cmap, dtype=np.float64).reshape(-1, 3)
    a = np.where(np.isnan(a), 0.0, a)
    return np.clip(np.rint(a), 0, 255).astype(np.uint8)


class HipWorker:
    def __init__(self, device=0):
        self.ctx = binding.Context(device)
        self.onmessage = None
        self.onerror = None

    def terminate(self):
        self.ctx.close()

    def postMessage(self, message, transfer=None):  # noqa: N802 (reference spelling)
        if not message or message.get("buffer") is None:
            return                                                        # worker.js:159
        try:
            reply = self.render(message)
        except binding.SpectroplotError as e:
            # the reference has no error channel (a throw leaves the caller's promise pending); surface it instead
            if self.onerror:
                self.onerror({"message": str(e), "status": e.status})
                return
            raise
        if self.onmessage:
            self.onmessage({"data": reply})

    def render(self, m):
        data = np.frombuffer(m["buffer"], dtype=np.uint8) if not isinstance(m["buffer"], np.ndarray) else m["buffer"]
        r = self.ctx.render(m["format"], data, m["n"], m["windowc"], m["block_norm"], m["gain"], m["range"], _lut_bytes(m["cmap"]),
                            m["width"], bool(m.get("channelMode")), bool(m.get("waterfall")))
        return {"cB_hist": r["cB_hist"], "c_hist": r["c_hist"], "dBfs_min": r["dBfs_min"], "dBfs_max": r["dBfs_max"],
                "offset": m.get("offset"), "gauge_mins": r["gauge_mins"], "gauge_maxs": r["gauge_maxs"],
                "gauge_amps": r["gauge_amps"], "imageData": {"data": r["rgba"]}}


def render_sliced(render_fn, data, fmt, n, width, workers, windowc, weight, cmap, gain, rng, channel_mode=False,
                  waterfall=False, force_ends=True):
    """The data half of the caller's processData (lib/spectroplot.js:1113-1130, :1206-1244).

    Splits the capture into `workers` contiguous slices (remainder samples and columns dropped), renders each through
    `render_fn(message) -> reply` and merges strips, histograms and the dBfs range.  `render_fn` is normally
    `HipWorker.render`; multi-GPU drivers pass one worker per device.  Un-rendered columns stay zero (canvas default).
    """
    data = np.ascontiguousarray(data, dtype=np.uint8)
    _, sw = binding.parse_format(fmt)
    cmap = [list(c) for c in cmap]
    if force_ends:
        cmap[0] = [0, 0, 0]
        cmap[-1] = [255, 255, 255]
    slice_w = width // workers
    block_norm = 1.0 / weight
    merged = np.zeros((width, n, 4) if waterfall else (n, width, 4), dtype=np.uint8)
    c_hist = np.zeros(len(cmap), dtype=np.uint64)
    cb_hist = np.zeros(binding.SP_CB_HIST_SIZE, dtype=np.uint64)
    dmin, dmax = 0.0, -200.0
    replies = []
    fns = render_fn if isinstance(render_fn, (list, tuple)) else [render_fn] * workers
    for i in range(workers):
        b0, b1 = binding.slice_bounds(data.size, sw, i, workers)
        r = fns[i]({"block_norm": block_norm, "gain": gain, "range": rng, "cmap": cmap, "n": n, "windowc": windowc,
                    "width": slice_w, "offset": i * slice_w, "buffer": data[b0:b1], "format": fmt,
                    "channelMode": channel_mode, "waterfall": waterfall})
        replies.append(r)
        if r["dBfs_min"] < dmin:
            dmin = r["dBfs_min"]
        if r["dBfs_max"] > dmax:
            dmax = r["dBfs_max"]
        c_hist += np.asarray(r["c_hist"], dtype=np.uint64)
        cb_hist += np.asarray(r["cB_hist"], dtype=np.uint64)
        img = np.asarray(r["imageData"]["data"], dtype=np.uint8)
        off = r["offset"]
        if waterfall:
            y0 = width - slice_w - off
            merged[y0:y0 + slice_w] = img.reshape(slice_w, n, 4)
        else:
            merged[:, off:off + slice_w] = img.reshape(n, slice_w, 4)
    return {"data": merged.reshape(-1), "c_hist": c_hist, "cB_hist": cb_hist, "dBfs_min": dmin, "dBfs_max": dmax,
            "slice_width": slice_w, "replies": replies}
