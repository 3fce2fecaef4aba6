"""crypto12381_amd — MI355X-native batched BLS12-381 backend for crypto12381's group/pairing API.

The product is the HIP library behind the C ABI of include/c12381_hip.h; this package only
holds its sources (csrc/), the build recipe and a thin ctypes binding used by tests and bench.
"""
from .capi import C12381Error, Context, load_library  # noqa: F401

__all__ = ["Context", "C12381Error", "load_library"]
