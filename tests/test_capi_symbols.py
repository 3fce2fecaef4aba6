"""CPU test: the C-ABI library builds, loads and exports every symbol include/c12381_hip.h declares
(no compute calls without a GPU)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from crypto12381_amd.build import build
    lib_path = build()
    lib = ctypes.CDLL(lib_path)
    names = set()
    for fn in os.listdir(os.path.join(ROOT, "include")):
        if fn.endswith(".h"):
            text = open(os.path.join(ROOT, "include", fn)).read()
            names |= set(re.findall(r"\b(c12381_[a-z0-9_]+)\s*\(", text))
    assert len(names) >= 10
    for n in sorted(names):
        assert hasattr(lib, n), f"symbol {n} declared in include/ but not exported"
    assert lib.c12381_version() >= 1


def test_create_fails_loudly_without_gpu():
    """No CPU fallback: on a box without a HIP device the context constructor raises."""
    import torch
    from crypto12381_amd import C12381Error, Context
    if torch.cuda.is_available():
        return
    try:
        Context(0)
    except C12381Error:
        return
    raise AssertionError("Context() succeeded without a GPU")
