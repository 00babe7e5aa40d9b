"""ctypes binding of include/spectroplot_hip.h.  No compute happens in Python."""
import ctypes as C
import os
import weakref
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

SP_CB_HIST_SIZE = 1000
FORMATS = ["CU4", "CS4", "CU8", "CS8", "CU12", "CS12", "CU16", "CS16", "CU32", "CS32", "CU64", "CS64", "CF32", "CF64"]

SP_ERR_NOT_POW2 = -2
SP_ERR_BYTE_LENGTH = -3
SP_ERR_UNSUPPORTED = -4
SP_ERR_NO_DEVICE = -5


class SpectroplotError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("%s (status %d)" % (message, status))
        self.status = status


class _Request(C.Structure):
    _fields_ = [("format", C.c_int32), ("n", C.c_int32), ("channel_mode", C.c_int32), ("waterfall", C.c_int32),
                ("lut_len", C.c_int32), ("reserved", C.c_int32), ("block_norm", C.c_double), ("gain", C.c_double),
                ("range", C.c_double), ("windowc", C.c_void_p), ("lut_rgb", C.c_void_p)]


class _NamedRequest(C.Structure):
    _fields_ = [("format", C.c_char_p), ("window", C.c_char_p), ("cmap", C.c_char_p), ("n", C.c_int32), ("channel_mode", C.c_int32),
                ("waterfall", C.c_int32), ("gain", C.c_double), ("range", C.c_double)]


class _Reply(C.Structure):
    _fields_ = [("rgba", C.c_void_p), ("gauge_mins", C.c_void_p), ("gauge_maxs", C.c_void_p), ("gauge_amps", C.c_void_p),
                ("c_hist", C.c_void_p), ("cb_hist", C.c_void_p), ("dbfs_minmax", C.c_void_p)]


def lib_path():
    # Kernel experiments only (tools/build_variant.sh, tools/ab_variants.sh): with SP_EXPERIMENT_KNOBS=1 in the environment,
    # SP_LIB_VARIANT=<name> loads lib/variants/<name>.so, a copy of the library built with other compile-time options.
    v = os.environ.get("SP_LIB_VARIANT") if os.environ.get("SP_EXPERIMENT_KNOBS") == "1" else None
    if v:
        return os.path.join(_HERE, "lib", "variants", v + ".so")
    return os.path.join(_HERE, "lib", "libspectroplot_hip.so")


def build_library(force=False):
    """Compiles the HIP library in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
    target = lib_path()
    if force:
        subprocess.check_call(["make", "-s", "-C", _HERE, "clean"])
    subprocess.check_call(["make", "-s", "-C", _HERE])
    return target


class Library:
    """The loaded shared library.  Fails loudly when it has not been built: there is no fallback path."""
    _instance = None

    def __init__(self, path=None):
        path = path or lib_path()
        if not os.path.exists(path):
            raise SpectroplotError(SP_ERR_NO_DEVICE, "HIP library %s is missing: run __graft_entry__.build() first" % path)
        L = self.L = C.CDLL(path)
        self.path = path
        vp, i32, sz, dbl = C.c_void_p, C.c_int32, C.c_size_t, C.c_double
        L.sp_version.restype = C.c_int
        L.sp_status_string.restype = C.c_char_p
        L.sp_status_string.argtypes = [C.c_int]
        L.sp_last_error.restype = C.c_char_p
        L.sp_last_error.argtypes = [vp]
        L.sp_format_parse.argtypes = [C.c_char_p, C.POINTER(i32), C.POINTER(i32)]
        L.sp_format_element_size.argtypes = [i32]
        L.sp_slice_bounds.argtypes = [sz, i32, i32, i32, C.POINTER(sz), C.POINTER(sz)]
        L.sp_window.argtypes = [C.c_char_p, i32, vp, C.POINTER(dbl)]
        L.sp_twiddles.argtypes = [i32, vp, vp]
        L.sp_js_log10.restype = dbl
        L.sp_js_log10.argtypes = [dbl]
        L.sp_cmap_count.restype = C.c_int
        L.sp_cmap_key.restype = C.c_char_p
        L.sp_cmap_key.argtypes = [i32]
        L.sp_cmap.argtypes = [C.c_char_p, vp, i32, C.POINTER(i32)]
        L.sp_cmap_generate.argtypes = [C.c_char_p, i32, vp]
        L.sp_plan_execute_from_host.argtypes = [vp, vp, sz, i32, C.POINTER(_Reply)]
        L.sp_device_count.argtypes = [C.POINTER(i32)]
        L.sp_context_create.argtypes = [i32, C.POINTER(vp)]
        L.sp_context_destroy.argtypes = [vp]
        L.sp_context_destroy.restype = None
        L.sp_context_set_stream.argtypes = [vp, vp]
        L.sp_context_synchronize.argtypes = [vp]
        L.sp_render.argtypes = [vp, C.POINTER(_Request), vp, sz, i32, C.POINTER(_Reply)]
        L.sp_render_named.argtypes = [vp, C.POINTER(_NamedRequest), vp, sz, i32, C.POINTER(_Reply)]
        L.sp_named_resolve.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(i32)]
        L.sp_context_plan_creations.argtypes = [vp, C.POINTER(C.c_int64)]
        L.sp_plan_create.argtypes = [vp, C.POINTER(_Request), C.POINTER(vp)]
        L.sp_plan_destroy.argtypes = [vp]
        L.sp_plan_destroy.restype = None
        L.sp_plan_execute.argtypes = [vp, vp, sz, i32, C.POINTER(_Reply)]
        L.sp_plan_kernel_name.restype = C.c_char_p
        L.sp_plan_kernel_name.argtypes = [vp]
        L.sp_plan_force_kernel.argtypes = [vp, i32]
        L.sp_device_alloc.argtypes = [vp, sz, C.POINTER(vp)]
        L.sp_device_free.argtypes = [vp, vp]
        L.sp_device_upload.argtypes = [vp, vp, vp, sz]
        L.sp_device_download.argtypes = [vp, vp, vp, sz]
        L.sp_device_memset.argtypes = [vp, vp, C.c_int, sz]
        L.sp_synth_trinoise.argtypes = [vp, vp, i32, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, dbl, dbl]
        L.sp_context_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]
        L.sp_context_enable_timing.argtypes = [vp, i32]
        L.sp_merge_replies.argtypes = [vp, vp, i32, i32, vp, vp, vp]
        L.sp_merge_replies_batch.argtypes = [vp, vp, i32, i32, i32, vp]
        L.sp_context_event_pair_overhead_ms.argtypes = [vp, C.POINTER(C.c_float)]
        L.sp_place_strips.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32]
        if not hasattr(L, "sp_group_create"):
            return          # (an older build loaded as an experiment variant, tools/ab_variants.sh: everything above is all it has)
        L.sp_group_create.argtypes = [C.POINTER(i32), i32, C.POINTER(vp)]
        L.sp_group_destroy.argtypes = [vp]
        L.sp_group_destroy.restype = None
        L.sp_group_size.argtypes = [vp]
        L.sp_group_render.argtypes = [vp, C.POINTER(_Request), vp, sz, i32, C.POINTER(_Reply)]
        L.sp_group_transport.restype = C.c_char_p
        L.sp_group_transport.argtypes = [vp]
        L.sp_group_last_error.restype = C.c_char_p
        L.sp_group_last_error.argtypes = [vp]
        if hasattr(L, "sp_group_render_ex"):
            L.sp_group_render_ex.argtypes = [vp, C.POINTER(_Request), vp, sz, i32, C.POINTER(_Reply), i32]
            L.sp_group_transport_note.restype = C.c_char_p
            L.sp_group_transport_note.argtypes = [vp]
            L.sp_group_last_timings.argtypes = [vp, C.POINTER(dbl), C.POINTER(dbl), C.POINTER(dbl)]
            L.sp_group_root_bytes.argtypes = [vp, C.POINTER(sz), C.POINTER(sz)]
            if hasattr(L, "sp_group_rccl_info"):
                L.sp_group_rccl_info.argtypes = [vp, C.c_char_p, sz]
            L.sp_render_strip.argtypes = [vp, C.POINTER(_Request), vp, sz, i32, C.POINTER(_Reply), i32]
        if hasattr(L, "sp_context_last_upload_bytes"):
            L.sp_context_last_upload_bytes.argtypes = [vp, C.POINTER(sz)]
        L.sp_context_get_stream.argtypes = [vp, C.POINTER(vp)]

    @classmethod
    def get(cls):
        if cls._instance is None:
            cls._instance = Library()
        return cls._instance

    def check(self, status, ctx=None):
        if status == 0:
            return
        msg = self.L.sp_last_error(ctx).decode() if ctx else ""
        raise SpectroplotError(status, msg or self.L.sp_status_string(status).decode())

    def device_count(self):
        n = C.c_int32(0)
        self.L.sp_device_count(C.byref(n))
        return n.value


def parse_format(name):
    """(format id, bytes per complex sample) for a reference format name (lib/samples.js:22-162)."""
    f, w = C.c_int32(), C.c_int32()
    Library.get().L.sp_format_parse(str(name).encode(), C.byref(f), C.byref(w))
    return f.value, w.value


def slice_bounds(nbytes, sample_width, index, count):
    b, e = C.c_size_t(), C.c_size_t()
    lib = Library.get()
    lib.check(lib.L.sp_slice_bounds(nbytes, sample_width, index, count, C.byref(b), C.byref(e)))
    return b.value, e.value


def window(name, n):
    lib = Library.get()
    out = np.empty(n, dtype=np.float64)
    w = C.c_double()
    lib.check(lib.L.sp_window(name.encode(), n, out.ctypes.data_as(C.c_void_p), C.byref(w)))
    return out, w.value


def twiddles(n):
    lib = Library.get()
    c = np.empty(max(n // 2, 1), dtype=np.float64)
    s = np.empty(max(n // 2, 1), dtype=np.float64)
    lib.check(lib.L.sp_twiddles(n, c.ctypes.data_as(C.c_void_p), s.ctypes.data_as(C.c_void_p)))
    return c[:n // 2], s[:n // 2]


def named_resolve(window, cmap):
    """(taper name as sp_window takes it, colour-map key, entry count) for two option names, defaults included
    (lib/spectroplot.js:238-264, lib/utils.js:25-40)."""
    w, k, n = C.c_char_p(), C.c_char_p(), C.c_int32()
    lib = Library.get()
    lib.check(lib.L.sp_named_resolve(str(window).encode(), str(cmap).encode(), C.byref(w), C.byref(k), C.byref(n)))
    return w.value.decode(), k.value.decode(), n.value


def _make_request(fmt_id, n, windowc, block_norm, gain, rng, lut, channel_mode, waterfall):
    windowc = np.ascontiguousarray(windowc, dtype=np.float64)
    lut = np.ascontiguousarray(lut, dtype=np.uint8).reshape(-1, 3)
    req = _Request(fmt_id, int(n), int(bool(channel_mode)), int(bool(waterfall)), len(lut), 0, float(block_norm), float(gain),
                   float(rng), windowc.ctypes.data_as(C.c_void_p), lut.ctypes.data_as(C.c_void_p))
    return req, (windowc, lut)


class Context:
    """One device + stream; the analogue of one reference Worker instance."""

    def __init__(self, device=0):
        self.lib = Library.get()
        h = C.c_void_p()
        self.lib.check(self.lib.L.sp_context_create(device, C.byref(h)))
        self.h = h
        self.device = device
        self._plans = weakref.WeakSet()   # live Plan objects of this context

    def close(self):
        if self.h:
            # plans hold a pointer to their context: they go first (a Plan object that outlives this call is inert)
            for p in list(self._plans):
                p.close()
            self.lib.L.sp_context_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, status):
        self.lib.check(status, self.h)

    def set_stream(self, stream_handle):
        self._chk(self.lib.L.sp_context_set_stream(self.h, C.c_void_p(stream_handle)))

    def get_stream(self):
        """The stream handle bound with set_stream (0 while the context uses its own): what set_stream takes to restore it."""
        s = C.c_void_p()
        self._chk(self.lib.L.sp_context_get_stream(self.h, C.byref(s)))
        return s.value or 0

    def synchronize(self):
        self._chk(self.lib.L.sp_context_synchronize(self.h))

    def last_upload_bytes(self):
        """Bytes of samples the last render() sent over the host link (a sparse request - stride > n - sends its frames only)."""
        v = C.c_size_t()
        self._chk(self.lib.L.sp_context_last_upload_bytes(self.h, C.byref(v)))
        return v.value

    def enable_timing(self, on=True):
        self._chk(self.lib.L.sp_context_enable_timing(self.h, int(on)))

    def merge_replies(self, d_records, count, lut_len, d_c_hist=0, d_cb_hist=0, d_minmax=0):
        """Device-side merge of `count` slice records [c_hist | cB_hist | dBfs_min, dBfs_max] (the caller's merge, spectroplot.js:1229-1238)."""
        self._chk(self.lib.L.sp_merge_replies(self.h, C.c_void_p(d_records), int(count), int(lut_len), C.c_void_p(d_c_hist or None),
                                              C.c_void_p(d_cb_hist or None), C.c_void_p(d_minmax or None)))

    def merge_replies_batch(self, d_gathered, ranks, renders, lut_len, d_merged):
        """The same merge for a batch of renders in ONE launch: d_gathered = [rank][render][record] as an all-gather of the ranks' batches
        delivers it, d_merged = `renders` merged records end to end (sp_merge_replies_batch)."""
        self._chk(self.lib.L.sp_merge_replies_batch(self.h, C.c_void_p(d_gathered), int(ranks), int(renders), int(lut_len), C.c_void_p(d_merged)))

    def place_strips(self, d_image, d_strips, count, n, width, slice_width, waterfall=False):
        """Device-side putImageData of `count` gathered strips (laid end to end at d_strips) into the merged image (spectroplot.js:1241-1244)."""
        self._chk(self.lib.L.sp_place_strips(self.h, C.c_void_p(d_image), C.c_void_p(d_strips), int(count), int(n), int(width),
                                             int(slice_width), int(bool(waterfall))))

    def event_pair_overhead_ms(self):
        ms = C.c_float()
        self._chk(self.lib.L.sp_context_event_pair_overhead_ms(self.h, C.byref(ms)))
        return float(ms.value)

    def last_kernel_ms(self):
        ms = C.c_float()
        self._chk(self.lib.L.sp_context_last_kernel_ms(self.h, C.byref(ms)))
        return ms.value

    # -- device memory -----------------------------------------------------------------------------
    def alloc(self, nbytes):
        p = C.c_void_p()
        self._chk(self.lib.L.sp_device_alloc(self.h, nbytes, C.byref(p)))
        return p.value

    def free(self, ptr):
        self._chk(self.lib.L.sp_device_free(self.h, C.c_void_p(ptr)))

    def upload(self, d_ptr, array):
        a = np.ascontiguousarray(array)
        self._chk(self.lib.L.sp_device_upload(self.h, C.c_void_p(d_ptr), a.ctypes.data_as(C.c_void_p), a.nbytes))

    def download(self, d_ptr, nbytes, dtype=np.uint8):
        out = np.empty(nbytes // np.dtype(dtype).itemsize, dtype=dtype)
        self._chk(self.lib.L.sp_device_download(self.h, out.ctypes.data_as(C.c_void_p), C.c_void_p(d_ptr), out.nbytes))
        return out

    def memset(self, d_ptr, value, nbytes):
        self._chk(self.lib.L.sp_device_memset(self.h, C.c_void_p(d_ptr), value, nbytes))

    def synth_trinoise(self, d_ptr, fmt, t0, count, seed, step, gshift, amp, namp):
        fid = parse_format(fmt)[0]
        self._chk(self.lib.L.sp_synth_trinoise(self.h, C.c_void_p(d_ptr), fid, t0, count, seed, step, gshift, amp, namp))

    # -- renderFft on host buffers -------------------------------------------------------------------
    def render(self, fmt, data, n, windowc, block_norm, gain, rng, lut, width, channel_mode=False, waterfall=False):
        """Same argument meaning as the reference message; returns the reply fields as numpy arrays."""
        fid, _ = parse_format(fmt)
        data = np.ascontiguousarray(data, dtype=np.uint8)
        req, keep = _make_request(fid, n, windowc, block_norm, gain, rng, lut, channel_mode, waterfall)
        W = int(width)
        L = len(keep[1])
        out = {"rgba": np.zeros(4 * max(W, 0) * n, np.uint8), "gauge_mins": np.zeros(max(W, 0), np.uint8),
               "gauge_maxs": np.zeros(max(W, 0), np.uint8), "gauge_amps": np.zeros(max(W, 0), np.uint8),
               "c_hist": np.zeros(L, np.uint64), "cB_hist": np.zeros(SP_CB_HIST_SIZE, np.uint64)}
        mm = np.array([0.0, -200.0])
        p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        rep = _Reply(p(out["rgba"]), p(out["gauge_mins"]), p(out["gauge_maxs"]), p(out["gauge_amps"]), p(out["c_hist"]),
                     p(out["cB_hist"]), p(mm))
        self._chk(self.lib.L.sp_render(self.h, C.byref(req), p(data), data.size, W, C.byref(rep)))
        out["dBfs_min"], out["dBfs_max"] = float(mm[0]), float(mm[1])
        return out

    def render_named(self, fmt, data, n, window, cmap, gain, rng, width, channel_mode=False, waterfall=False):
        """The request by option names, as the reference's caller assembles its message (lib/spectroplot.js:1113-1146): the library
        evaluates taper, block_norm and colour map (ends forced) itself and keeps the plan while names and numbers repeat."""
        data = np.ascontiguousarray(data, dtype=np.uint8)
        W = int(width)
        L = named_resolve(window, cmap)[2]
        req = _NamedRequest(str(fmt).encode(), str(window).encode(), str(cmap).encode(), int(n), int(bool(channel_mode)),
                            int(bool(waterfall)), float(gain), float(rng))
        out = {"rgba": np.zeros(4 * max(W, 0) * n, np.uint8), "gauge_mins": np.zeros(max(W, 0), np.uint8),
               "gauge_maxs": np.zeros(max(W, 0), np.uint8), "gauge_amps": np.zeros(max(W, 0), np.uint8),
               "c_hist": np.zeros(L, np.uint64), "cB_hist": np.zeros(SP_CB_HIST_SIZE, np.uint64)}
        mm = np.array([0.0, -200.0])
        p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        rep = _Reply(p(out["rgba"]), p(out["gauge_mins"]), p(out["gauge_maxs"]), p(out["gauge_amps"]), p(out["c_hist"]),
                     p(out["cB_hist"]), p(mm))
        self._chk(self.lib.L.sp_render_named(self.h, C.byref(req), p(data), data.size, W, C.byref(rep)))
        out["dBfs_min"], out["dBfs_max"] = float(mm[0]), float(mm[1])
        return out

    def plan_creations(self):
        n = C.c_int64()
        self._chk(self.lib.L.sp_context_plan_creations(self.h, C.byref(n)))
        return n.value

    def plan(self, fmt, n, windowc, block_norm, gain, rng, lut, channel_mode=False, waterfall=False):
        return Plan(self, fmt, n, windowc, block_norm, gain, rng, lut, channel_mode, waterfall)


class Plan:
    """Request constants resident on the device; execute() runs the frame loop on device-resident operands."""

    def __init__(self, ctx, fmt, n, windowc, block_norm, gain, rng, lut, channel_mode=False, waterfall=False):
        self.ctx = ctx
        self.n = int(n)
        self.fid, self.sample_width = parse_format(fmt)
        req, keep = _make_request(self.fid, n, windowc, block_norm, gain, rng, lut, channel_mode, waterfall)
        self.lut_len = len(keep[1])
        h = C.c_void_p()
        ctx._chk(ctx.lib.L.sp_plan_create(ctx.h, C.byref(req), C.byref(h)))
        self.h = h
        ctx._plans.add(self)

    def close(self):
        if self.h:
            self.ctx.lib.L.sp_plan_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def kernel_name(self):
        return self.ctx.lib.L.sp_plan_kernel_name(self.h).decode()

    def force_kernel(self, which):
        self.ctx._chk(self.ctx.lib.L.sp_plan_force_kernel(self.h, {"auto": 0, "scratch": 1, "frames": 3}[which]))

    def execute(self, d_bytes, nbytes, width, rgba=0, gauge_mins=0, gauge_maxs=0, gauge_amps=0, c_hist=0, cb_hist=0, dbfs_minmax=0):
        """All pointer arguments are device addresses (ints); 0 skips that output. Asynchronous on the context's stream."""
        rep = _Reply(rgba or None, gauge_mins or None, gauge_maxs or None, gauge_amps or None, c_hist or None, cb_hist or None,
                     dbfs_minmax or None)
        self.ctx._chk(self.ctx.lib.L.sp_plan_execute(self.h, C.c_void_p(d_bytes), nbytes, int(width), C.byref(rep)))

    def execute_from_host(self, data, width, rgba=0, gauge_mins=0, gauge_maxs=0, gauge_amps=0, c_hist=0, cb_hist=0, dbfs_minmax=0):
        """sp_plan_execute_from_host: `data` is the capture in HOST memory (numpy uint8; it must stay alive until the context has been
        synchronised), the outputs are device addresses as for execute().  The samples travel in chunks under the renders; a sparse
        request uploads only what its frames read (ctx.last_upload_bytes())."""
        data = np.ascontiguousarray(data, dtype=np.uint8)
        rep = _Reply(rgba or None, gauge_mins or None, gauge_maxs or None, gauge_amps or None, c_hist or None, cb_hist or None,
                     dbfs_minmax or None)
        self.ctx._chk(self.ctx.lib.L.sp_plan_execute_from_host(self.h, data.ctypes.data_as(C.c_void_p), data.size, int(width), C.byref(rep)))
        return data


class Group:
    """The caller's sliced render from one process (sp_group_*): one member context per listed device, slice r rendered on member r,
    strips gathered device to device (RCCL or peer copies) and merged on the root (lib/spectroplot.js:1206-1244)."""

    def __init__(self, devices):
        self.lib = Library.get()
        arr = (C.c_int32 * len(devices))(*devices)
        h = C.c_void_p()
        self.lib.check(self.lib.L.sp_group_create(arr, len(devices), C.byref(h)))
        self.h = h
        self.size = len(devices)

    def close(self):
        if self.h:
            self.lib.L.sp_group_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def transport(self):
        return self.lib.L.sp_group_transport(self.h).decode()

    def transport_note(self):
        """Why the last transport was chosen when it was not the first choice (RCCL failures, peer access that could not be enabled)."""
        return self.lib.L.sp_group_transport_note(self.h).decode()

    def rccl_info(self):
        """Which RCCL the group has loaded (library path, version code, communicators); '' while none is."""
        buf = C.create_string_buffer(512)
        self.lib.L.sp_group_rccl_info(self.h, buf, len(buf))
        return buf.value.decode()

    def timings(self):
        """Milliseconds of the last render's phases: (upload + render of the slowest member, gather on the root, download)."""
        a, b, c = C.c_double(), C.c_double(), C.c_double()
        self.lib.L.sp_group_last_timings(self.h, C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value

    def root_bytes(self):
        """(image bytes, staging bytes) the root member holds on its device for the gather."""
        a, b = C.c_size_t(), C.c_size_t()
        self.lib.L.sp_group_root_bytes(self.h, C.byref(a), C.byref(b))
        return a.value, b.value

    def render(self, fmt, data, n, windowc, block_norm, gain, rng, lut, width, channel_mode=False, waterfall=False, gather="device",
               dirty=None):
        """Same argument meaning as Context.render; the reply is the caller's MERGED result (`rgba` the whole image, histograms and
        dBfs range over all slices, gauges with slice r's at [r * slice_width, (r + 1) * slice_width)).  gather: "device" (strips meet
        in the root's HBM: RCCL / peer copies) or "host" (every member writes its band of the host image over its own link).
        dirty: a byte value the output buffers are pre-filled with (tests: what no slice draws must come back cleared)."""
        fid, _ = parse_format(fmt)
        data = np.ascontiguousarray(data, dtype=np.uint8)
        req, keep = _make_request(fid, n, windowc, block_norm, gain, rng, lut, channel_mode, waterfall)
        W = int(width)
        L = len(keep[1])
        out = {"rgba": np.zeros(4 * max(W, 0) * n, np.uint8), "gauge_mins": np.zeros(max(W, 0), np.uint8),
               "gauge_maxs": np.zeros(max(W, 0), np.uint8), "gauge_amps": np.zeros(max(W, 0), np.uint8),
               "c_hist": np.zeros(L, np.uint64), "cB_hist": np.zeros(SP_CB_HIST_SIZE, np.uint64)}
        mm = np.array([0.0, -200.0])
        p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        rep = _Reply(p(out["rgba"]), p(out["gauge_mins"]), p(out["gauge_maxs"]), p(out["gauge_amps"]), p(out["c_hist"]),
                     p(out["cB_hist"]), p(mm))
        if dirty is not None:
            for k in ("rgba", "gauge_mins", "gauge_maxs", "gauge_amps"):
                out[k][:] = dirty
        mode = {"device": 0, "host": 1}[gather]
        status = self.lib.L.sp_group_render_ex(self.h, C.byref(req), p(data), data.size, W, C.byref(rep), mode)
        if status:
            raise SpectroplotError(status, self.lib.L.sp_group_last_error(self.h).decode() or self.lib.L.sp_status_string(status).decode())
        out["dBfs_min"], out["dBfs_max"] = float(mm[0]), float(mm[1])
        out["slice_width"] = W // self.size
        return out
