"""Deterministic synthetic I/Q generator (numpy) — bit-identical to oracle/js/siggen.js and csrc/sp_siggen.hip.

Test/bench infrastructure.  See oracle/js/siggen.js for the definition of the 'bytes', 'trinoise', 'zeros' and
'hexrepeat' kinds; every arithmetic step below is the same exact-integer / single-rounding f64 operation.
"""
import numpy as np

SAMPLE_WIDTH = {"CU4": 1, "CS4": 1, "CU8": 2, "CS8": 2, "CU12": 3, "CS12": 3, "CU16": 4, "CS16": 4,
                "CU32": 8, "CS32": 8, "CU64": 16, "CS64": 16, "CF32": 8, "CF64": 16}


def fmix32(h):
    h = np.asarray(h, dtype=np.uint32).copy()
    h ^= h >> np.uint32(16)
    h *= np.uint32(0x85EBCA6B)
    h ^= h >> np.uint32(13)
    h *= np.uint32(0xC2B2AE35)
    h ^= h >> np.uint32(16)
    return h


def hash32(seed, idx):
    return fmix32(np.uint32(seed & 0xFFFFFFFF) ^ np.asarray(idx, dtype=np.uint32))


def _values(gen, t, c):
    """f64 signal value of component c (0 = I, 1 = Q) at sample indices t (uint64/int64 array)."""
    t = np.asarray(t, dtype=np.int64)
    ph = ((t & 0xFFFF) * int(gen["step"])) & 0xFFFF
    if c:
        ph = (ph - 16384) & 0xFFFF
    tri = (np.abs(ph - 32768) - 16384).astype(np.float64) / 16384.0
    gate = np.where(((t & 0xFFFFFFFF) >> int(gen["gshift"])) & 1, 1.0, 0.25)
    u = hash32(gen["seed"], ((2 * t + c) & 0xFFFFFFFF).astype(np.uint32)).astype(np.float64) / 4294967296.0 - 0.5
    return (np.float64(gen["amp"]) * gate) * tri + np.float64(gen["namp"]) * u


def _q(v, mul, add, lo, hi):
    return np.clip(np.floor(v * mul + add), lo, hi)


def generate(fmt, gen, count, t0=0):
    """Returns `count` complex samples of format `fmt` starting at global sample index t0, as a uint8 array."""
    fmt = fmt.upper()
    sw = SAMPLE_WIDTH[fmt]
    nbytes = count * sw
    kind = gen["kind"]
    if kind == "zeros":
        return np.zeros(nbytes, dtype=np.uint8)
    if kind == "hexrepeat":
        pat = np.frombuffer(bytes.fromhex(gen["hex"]), dtype=np.uint8)
        return np.resize(pat, nbytes).copy()
    if kind == "bytes":
        w0 = (t0 * sw) >> 2
        words = (nbytes + 3) // 4
        w = hash32(gen["seed"], (np.arange(words, dtype=np.uint64) + np.uint64(w0)).astype(np.uint32))
        return w.astype("<u4").view(np.uint8)[:nbytes].copy()
    if kind != "trinoise":
        raise ValueError("unknown generator kind %r" % kind)
    t = np.arange(count, dtype=np.int64) + int(t0)
    vi, vq = _values(gen, t, 0), _values(gen, t, 1)
    out = np.zeros((count, sw), dtype=np.uint8)

    def put(col, arr, dt):
        b = np.ascontiguousarray(arr.astype(dt)).view(np.uint8).reshape(count, -1)
        out[:, col:col + b.shape[1]] = b

    if fmt == "CF32":
        put(0, vi, "<f4"); put(4, vq, "<f4")
    elif fmt == "CF64":
        put(0, vi, "<f8"); put(8, vq, "<f8")
    elif fmt == "CS16":
        put(0, _q(vi, 32767, 0.5, -32768, 32767), "<i2"); put(2, _q(vq, 32767, 0.5, -32768, 32767), "<i2")
    elif fmt == "CU16":
        put(0, _q(vi, 32767.5, 32768, 0, 65535), "<u2"); put(2, _q(vq, 32767.5, 32768, 0, 65535), "<u2")
    elif fmt == "CS8":
        put(0, _q(vi, 127, 0.5, -128, 127), "i1"); put(1, _q(vq, 127, 0.5, -128, 127), "i1")
    elif fmt == "CU8":
        put(0, _q(vi, 127.5, 128, 0, 255), "u1"); put(1, _q(vq, 127.5, 128, 0, 255), "u1")
    elif fmt == "CS32":
        put(0, _q(vi, 2147483647, 0.5, -2147483648, 2147483647), "<i4")
        put(4, _q(vq, 2147483647, 0.5, -2147483648, 2147483647), "<i4")
    elif fmt == "CU32":
        put(0, _q(vi, 2147483647.5, 2147483648, 0, 4294967295), "<u4")
        put(4, _q(vq, 2147483647.5, 2147483648, 0, 4294967295), "<u4")
    elif fmt in ("CS64", "CU64"):
        if fmt == "CS64":
            hi_i = _q(vi, 2147483647, 0.5, -2147483648, 2147483647).astype(np.int64).astype("<i4").view("<u4")
            hi_q = _q(vq, 2147483647, 0.5, -2147483648, 2147483647).astype(np.int64).astype("<i4").view("<u4")
        else:
            hi_i = _q(vi, 2147483647.5, 2147483648, 0, 4294967295).astype("<u4")
            hi_q = _q(vq, 2147483647.5, 2147483648, 0, 4294967295).astype("<u4")
        s2 = (gen["seed"] ^ 0x10101010) & 0xFFFFFFFF
        put(0, hash32(s2, ((2 * t) & 0xFFFFFFFF).astype(np.uint32)), "<u4")
        put(4, hi_i, "<u4")
        put(8, hash32(s2, ((2 * t + 1) & 0xFFFFFFFF).astype(np.uint32)), "<u4")
        put(12, hi_q, "<u4")
    elif fmt in ("CS12", "CU12"):
        if fmt == "CS12":
            i12 = _q(vi, 2047, 0.5, -2048, 2047).astype(np.int64) & 0xFFF
            q12 = _q(vq, 2047, 0.5, -2048, 2047).astype(np.int64) & 0xFFF
        else:
            i12 = _q(vi, 2047.5, 2048, 0, 4095).astype(np.int64) & 0xFFF
            q12 = _q(vq, 2047.5, 2048, 0, 4095).astype(np.int64) & 0xFFF
        out[:, 0] = (i12 & 0xFF).astype(np.uint8)
        out[:, 1] = (((i12 >> 8) & 0x0F) | ((q12 & 0x0F) << 4)).astype(np.uint8)
        out[:, 2] = ((q12 >> 4) & 0xFF).astype(np.uint8)
    elif fmt in ("CS4", "CU4"):
        if fmt == "CS4":
            i4 = _q(vi, 7, 0.5, -8, 7).astype(np.int64) & 0xF
            q4 = _q(vq, 7, 0.5, -8, 7).astype(np.int64) & 0xF
        else:
            i4 = _q(vi, 7.5, 8, 0, 15).astype(np.int64) & 0xF
            q4 = _q(vq, 7.5, 8, 0, 15).astype(np.int64) & 0xF
        out[:, 0] = ((i4 << 4) | q4).astype(np.uint8)
    else:
        raise ValueError("unhandled format " + fmt)
    return out.reshape(-1)


def case_input(case):
    """Input bytes of a tests/golden/cases.json worker case."""
    gf = case.get("gen_format") or case["format"].upper()
    sw = SAMPLE_WIDTH.get(gf, 2)
    count = -(-case["bytes"] // sw)
    return generate(gf if gf in SAMPLE_WIDTH else "CU8", case["gen"], count)[:case["bytes"]].copy()
