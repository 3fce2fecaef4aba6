"""Build the HIP shared library in-tree (crypto12381_amd/lib/libc12381_hip.so) for gfx950."""
from __future__ import annotations

import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "c12381_hip.hip")
LIB = os.path.join(HERE, "lib", "libc12381_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-shared", "-fPIC", "-std=c++17"]


def _sources():
    d = os.path.join(HERE, "csrc")
    return [os.path.join(d, f) for f in os.listdir(d) if f.endswith((".hip", ".hpp", ".h"))] + [
        os.path.join(os.path.dirname(HERE), "include", "c12381_hip.h")]


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(s) > t for s in _sources())


def build(force: bool = False, verbose: bool = False) -> str:
    if force or needs_build():
        os.makedirs(os.path.dirname(LIB), exist_ok=True)
        cmd = [HIPCC, *FLAGS, "-o", LIB, SRC]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force=True, verbose=True)
