"""ctypes front end of the C oracle (oracle/sp_oracle.c).  TEST INFRASTRUCTURE — never imported by the product.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "libsp_oracle.so")

ERR_NOT_POW2 = -1
ERR_BYTE_LENGTH = -2


class OracleError(Exception):
    def __init__(self, code):
        self.code = code
        msg = {ERR_NOT_POW2: "Length is not a power of 2", ERR_BYTE_LENGTH: "byte length is not a multiple of the element size"}
        super().__init__(msg.get(code, "oracle error %d" % code))


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("sp_oracle.c", "v8math.h", "Makefile")]
    if force or not os.path.exists(_LIB) or any(os.path.getmtime(s) > os.path.getmtime(_LIB) for s in src):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.spo_render.restype = C.c_int
        _lib.spo_render.argtypes = [C.c_char_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_double, C.c_double, C.c_double,
                                    C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p, C.c_void_p]
        _lib.spo_window.argtypes = [C.c_char_p, C.c_int, C.c_void_p, C.POINTER(C.c_double)]
        _lib.spo_twiddles.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
        _lib.spo_fft.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        _lib.spo_decode.argtypes = [C.c_char_p, C.c_void_p, C.c_size_t, C.c_int64, C.c_int64, C.c_void_p,
                                    C.POINTER(C.c_double), C.POINTER(C.c_int)]
        for f in ("spo_log10", "spo_cos", "spo_sin"):
            getattr(_lib, f).restype = C.c_double
            getattr(_lib, f).argtypes = [C.c_double]
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def window(name, n):
    out = np.empty(n, dtype=np.float64)
    w = C.c_double()
    rc = lib().spo_window(name.encode(), n, _ptr(out), C.byref(w))
    if rc:
        raise OracleError(rc)
    return out, w.value


def twiddles(n):
    c = np.empty(max(n // 2, 1), dtype=np.float64)
    s = np.empty(max(n // 2, 1), dtype=np.float64)
    rc = lib().spo_twiddles(n, _ptr(c), _ptr(s))
    if rc:
        raise OracleError(rc)
    return c[:n // 2], s[:n // 2]


def fft(re, im, split=False):
    re = np.array(re, dtype=np.float64)
    im = np.array(im, dtype=np.float64)
    rc = lib().spo_fft(len(re), _ptr(re), _ptr(im), int(split))
    if rc:
        raise OracleError(rc)
    return re, im


def decode(fmt, data, pos_lo, count):
    data = np.ascontiguousarray(data, dtype=np.uint8)
    out = np.empty(2 * count, dtype=np.float64)
    sc, sw = C.c_double(), C.c_int()
    rc = lib().spo_decode(fmt.encode(), _ptr(data), data.size, pos_lo, count, _ptr(out), C.byref(sc), C.byref(sw))
    if rc:
        raise OracleError(rc)
    return out.reshape(count, 2), sc.value, sw.value


def render(fmt, data, n, windowc, block_norm, gain, rng, lut, width, channel_mode=False, waterfall=False, planes=False):
    """One worker render.  `lut` is an (len, 3) uint8 array (ends already forced by the caller if desired)."""
    data = np.ascontiguousarray(data, dtype=np.uint8)
    windowc = np.ascontiguousarray(windowc, dtype=np.float64)
    lut = np.ascontiguousarray(lut, dtype=np.uint8).reshape(-1, 3)
    W = int(width)
    rgba = np.zeros(4 * W * n, dtype=np.uint8)
    gmin = np.zeros(W, dtype=np.uint8)
    gmax = np.zeros(W, dtype=np.uint8)
    gamp = np.zeros(W, dtype=np.uint8)
    c_hist = np.zeros(len(lut), dtype=np.int64)
    cb_hist = np.zeros(1000, dtype=np.int64)
    dmin, dmax = C.c_double(), C.c_double()
    db = np.zeros(W * n, dtype=np.float64) if planes else None
    a2 = np.zeros(W * n, dtype=np.float64) if planes else None
    rc = lib().spo_render(fmt.encode(), _ptr(data), data.size, n, _ptr(windowc), block_norm, gain, rng, _ptr(lut), len(lut), W,
                          int(bool(channel_mode)), int(bool(waterfall)), _ptr(rgba), _ptr(gmin), _ptr(gmax), _ptr(gamp),
                          _ptr(c_hist), _ptr(cb_hist), C.byref(dmin), C.byref(dmax),
                          _ptr(db) if planes else None, _ptr(a2) if planes else None)
    if rc:
        raise OracleError(rc)
    out = {"rgba": rgba, "gauge_mins": gmin, "gauge_maxs": gmax, "gauge_amps": gamp, "c_hist": c_hist, "cB_hist": cb_hist,
           "dBfs_min": dmin.value, "dBfs_max": dmax.value}
    if planes:
        out["db"] = db.reshape(W, n)
        out["abs2"] = a2.reshape(W, n)
    return out


def slice_bounds(nbytes, sample_width, index, count):
    """Byte range of the caller's slice `index` of `count` (samples.js:253-258)."""
    end_sample = nbytes // sample_width
    slice_len = sample_width * (end_sample // count)
    return slice_len * index, slice_len * (index + 1)
