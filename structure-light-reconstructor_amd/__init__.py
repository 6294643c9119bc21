"""structure-light-reconstructor_amd -- MI355X-native structured-light decode + triangulate engine.

The product is libslr_hip.so (hand-written HIP kernels for gfx950 behind the C ABI of include/slr.h).  This
Python package is a thin ctypes mirror of that ABI (capi), a synthetic scene generator for tests/bench (synth)
and the frame-sharding helper for multi-GPU runs (dist).  The directory name contains '-', so import it with

    import importlib; slr = importlib.import_module("structure-light-reconstructor_amd")

(tests/conftest.py and bench.py do exactly that).  There is NO CPU fallback anywhere in this package.
"""
from . import capi  # noqa: F401
from .capi import Context, SlrError, make_calib, make_camera  # noqa: F401

__all__ = ["capi", "Context", "SlrError", "make_calib", "make_camera"]
