"""CPU: the upload plan of sp_render (sp_debug_upload_plan: chunks of frames; for sparse requests the packed device layout and its pitched
copies) checked frame by frame against the reference's own frame positions, ~~(0.5 + stride * x) (lib/worker.js:50, 72), for a few thousand
request shapes.  No device: the plan is host arithmetic.  What the GPU tests check through results, this checks through geometry: every
frame's samples are copied from where the reference reads them to where the kernel will look for them, inside both buffers, and no two
rows of a chunk overlap on the device."""
import ctypes as C
import struct

import numpy as np
import pytest

from __graft_entry__ import load_package


@pytest.fixture(scope="module")
def lib():
    L = load_package().Library.get().L
    L.sp_debug_upload_plan.argtypes = [C.c_int32, C.c_int32, C.c_size_t, C.c_int32, C.c_int32, C.POINTER(C.c_int64), C.c_size_t, C.POINTER(C.c_size_t)]
    return L


FORMATS = {"CU4": (0, 1), "CU8": (2, 2), "CS12": (5, 3), "CS16": (7, 4), "CF32": (12, 8), "CF64": (13, 16)}


def to_int32(v):
    """ToInt32 of a non-negative double below 2^31 (every position in these tests)."""
    return int(np.floor(v))


def plan(lib, fmt, n, samples, width, want_image=1):
    fid, sw = FORMATS[fmt]
    buf = (C.c_int64 * (1 << 16))()
    used = C.c_size_t()
    rc = lib.sp_debug_upload_plan(fid, n, samples * sw, width, want_image, buf, len(buf), C.byref(used))
    assert rc == 0, (rc, used.value)
    return list(buf[:used.value]), sw


def check_shape(lib, fmt, n, samples, width):
    v, sw = plan(lib, fmt, n, samples, width)
    packed, chunks, dev_bytes, link_bytes = v[0], v[1], v[2], v[3]
    stride = np.float64(samples - n) / np.float64(width - 1) if width > 1 else np.float64(0)
    pos = lambda x: to_int32(np.float64(0.5) + stride * np.float64(x))   # noqa: E731  worker.js:72
    i, x_prev, moved = 4, 0, 0
    for _ in range(chunks):
        x0, x1 = v[i], v[i + 1]
        i += 2
        assert x0 == x_prev and x1 > x0 and (x1 % 32 == 0 or x1 == width), (x0, x1, width)
        x_prev = x1
        if not packed:
            continue
        first, F, P, dev_off, pos2_x0, pos2_last, bits, nblocks = v[i:i + 8]
        i += 8
        stride2 = struct.unpack("<d", struct.pack("<q", bits))[0]
        assert first == pos(x0) and F == int(np.floor(stride)) and P >= n
        rows = x1 - x0
        covered = np.zeros(rows, bool)
        extents = []                                  # (device start, device end) of every row as copied, in samples from dev_off
        for _b in range(nblocks):
            j0, j1, dmin, dmax = v[i:i + 4]
            i += 4
            assert 0 <= j0 < j1 <= rows and 0 <= dmin <= dmax and dmax - dmin <= max(n // 2, 16)
            for j in range(j0, j1):
                assert not covered[j]
                covered[j] = True
                d = pos(x0 + j) - first - j * F           # the frame's drift against the source pitch
                assert dmin <= d <= dmax, (j, d, dmin, dmax)
                # where the kernel looks for frame x0 + j (its own position formula on the packed layout) ...
                k = to_int32(np.float64(0.5) + np.float64(stride2) * np.float64(x0 + j)) - pos2_x0
                # ... is where the pitched copy puts the frame's first sample: row j starts at j * P + dmin and holds source
                # samples first + j * F + dmin onwards
                assert k == j * P + d, (fmt, n, samples, width, x0, j, k, j * P + d)
                # source side: the row's samples lie inside the capture up to the frame's end (the copy is shortened for the request's
                # last rows, see upload_packed_chunk); device side: inside the chunk's area
                assert first + j * F + d + n <= samples
                extents.append((j * P + dmin, j * P + dmax + n))
            moved += (j1 - j0) * (n + dmax - dmin) * sw
        assert covered.all()
        extents.sort()
        for a, b in zip(extents, extents[1:]):
            assert a[1] <= b[0], ("rows overlap on the device", a, b)
        assert pos2_last - pos2_x0 == (rows - 1) * P + (pos(x1 - 1) - first - (rows - 1) * F)
        assert dev_off + (extents[-1][1]) * sw <= dev_bytes
    assert x_prev == width and i == len(v)
    if packed:
        assert moved == link_bytes and width * n * sw <= link_bytes <= samples * sw * 3 // 4
        assert stride > n
    return bool(packed), chunks


def test_sparse_shapes_are_packed_and_every_frame_lands_where_the_kernel_looks(lib):
    rs = np.random.RandomState(20261004)
    packed_count = chunked = 0
    for _ in range(1500):
        fmt = str(rs.choice(list(FORMATS)))
        n = int(1 << rs.randint(6, 14))
        width = int(rs.choice([2, 3, 17, 100, 1000, 1024, 2048, 4096, 5000]))
        hop = float(rs.choice([1.0, 1.5, 2.0, 3.0, 4.25, 8.0, 8.003, 33.3]))
        samples = int(n + (width - 1) * n * hop + rs.randint(0, 7))
        if samples * FORMATS[fmt][1] > (1 << 33) or width * n > (1 << 27):
            continue
        p, c = check_shape(lib, fmt, n, samples, width)
        packed_count += p
        chunked += p and c > 1
    assert packed_count > 500 and chunked > 50, (packed_count, chunked)


def test_the_reference_interactive_shape(lib):
    """16 MSample cf32 at 2048 frames (BASELINE config 2's capture at a screen-wide image)."""
    p, c = check_shape(lib, "CF32", 1024, 1 << 24, 2048)
    assert p and c == 4
    v, _ = plan(lib, "CF32", 1024, 1 << 24, 2048)
    assert 2048 * 1024 * 8 <= v[3] <= 2048 * 1024 * 8 * 5 // 4          # link bytes: the frames' own samples + the rows' widening


def test_dense_and_degenerate_shapes_are_left_alone(lib):
    for fmt, n, samples, width in (("CF32", 1024, 1 << 24, 16384), ("CU8", 512, 1 << 20, 2048), ("CS16", 2048, 1 << 22, 3001),
                                   ("CF32", 1024, 1 << 14, 1), ("CU8", 64, 100, 300)):
        p, c = check_shape(lib, fmt, n, samples, width)
        assert not p
    # config 2 itself: six chunks shrinking towards the end (the samples are the longer transfer), the last one a few hundred frames
    v, _ = plan(lib, "CF32", 1024, 1 << 24, 16384)
    sizes = [v[5 + 2 * k] - v[4 + 2 * k] for k in range(v[1])]
    assert v[1] == 6 and sizes == sorted(sizes, reverse=True) and 256 <= sizes[-1] <= 1024, sizes
    # config 1's format at a long capture: the image is the longer transfer, the chunks grow
    v, _ = plan(lib, "CU8", 512, 1 << 25, 65536)
    sizes = [v[5 + 2 * k] - v[4 + 2 * k] for k in range(v[1])]
    assert v[1] == 6 and sizes == sorted(sizes), sizes
