"""spectroplot-js_amd — MI355X-native drop-in for the compute worker of triq-org/spectroplot-js.

The product is the C-ABI HIP library (include/spectroplot_hip.h, built into lib/libspectroplot_hip.so) and the
Node.js host layer on top of it (js/hip_worker.js + the N-API addon).  This Python package is the thin ctypes
mirror of the same ABI used by the parity tests, bench.py and the multi-GPU driver; it holds no compute of its own
and there is no CPU fallback: importing works anywhere, but every compute call needs the HIP library and a GPU.

Load it with `importlib` (the directory name has a hyphen), e.g. `from __graft_entry__ import load_package`.
"""
from . import binding  # noqa: F401
from .binding import (Library, Context, Plan, Group, SpectroplotError, FORMATS, lib_path, build_library,  # noqa: F401
                      parse_format, slice_bounds, window, twiddles)
from .worker import HipWorker, render_sliced  # noqa: F401

__all__ = ["Library", "Context", "Plan", "Group", "SpectroplotError", "FORMATS", "HipWorker", "render_sliced", "lib_path",
           "build_library", "parse_format", "slice_bounds", "window", "twiddles"]
