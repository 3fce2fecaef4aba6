"""Build the HIP shared library in-tree (crypto12381_amd/lib/libc12381_hip.so) for gfx950.

One hipcc compile per translation unit (csrc/*.hip, in parallel), then one link.  Objects are cached under lib/obj/ and
reused only when (a) no source or header is newer and (b) the stamp next to the object — a hash of the exact command
line and of `hipcc --version` — matches: an object built with other flags or another compiler is never linked in."""
from __future__ import annotations

import hashlib
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
UNITS = ["c12381_hip.hip", "k_g1.hip", "k_g2gt.hip", "k_g2h.hip", "k_pair3.hip", "k_hash_zp.hip", "k_fixed.hip"]
LIB = os.path.join(HERE, "lib", "libc12381_hip.so")
OBJ = os.path.join(HERE, "lib", "obj")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -fno-optimize-sibling-calls: LLVM drops the "callee saves nothing" treatment of an internal function as soon as one
# call site carries a `tail` marker (which every call passing only non-stack pointers gets); without the marker the
# big out-of-line field routines do not save ~65 callee-saved VGPRs in their prologues (130 scratch instructions a call)
# -fno-optimize-sibling-calls: a `tail` call marker costs an internal routine its "callee saves nothing" treatment.
# max-ilp scheduling: every kernel here runs at a FIXED occupancy (launch bounds: 2 waves per SIMD), so the default strategy's
# effort to raise occupancy buys nothing, while scheduling for ILP shortens the dependent multiply-add chains (A/B on MI355X,
# tools/ab_all.sh: pairing kernel 22.8 -> 22.1 ms).  NOT combined with -amdgpu-use-amdgpu-trackers=1: that (experimental)
# option gained another 1 %, but together with max-ilp an experimental variant of the pairing routines (Fp4 squarings as
# calls) returned wrong values for a few lanes of a full-size batch while passing every small test — not worth the risk.
CFLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-fno-optimize-sibling-calls",
          "-mllvm", "-amdgpu-sched-strategy=max-ilp"]
# Round 2 reproduced the wrong-values event with a rebuilt variant: whole wavefront groups wrong, plain grid as
# well as queue, only with -amdgpu-use-amdgpu-trackers=1; the same source without it is exact on every lane.  The option is
# refused outright, and the compiler the full-batch parity tests were run with is recorded: another one prints a notice
# (run tests/test_gpu_full_batch.py before trusting a build from it).
FORBIDDEN_FLAGS = ("amdgpu-use-amdgpu-trackers",)
VALIDATED_COMPILER = "AMD clang version 22.0.0git"          # ROCm 7.2.0


def _headers():
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".h"))] + [
        os.path.join(os.path.dirname(HERE), "include", "c12381_hip.h")]


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


_HIPCC_VERSION = None


def _stamp(unit: str) -> str:
    """hash of everything besides the sources that decides what the object contains"""
    global _HIPCC_VERSION
    if any(bad in f for f in CFLAGS for bad in FORBIDDEN_FLAGS):
        raise RuntimeError("build flag known to produce wrong pairings at full size: %s (DESIGN.md 5b)" % ", ".join(FORBIDDEN_FLAGS))
    if _HIPCC_VERSION is None:
        _HIPCC_VERSION = subprocess.run([HIPCC, "--version"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
        if VALIDATED_COMPILER not in _HIPCC_VERSION:
            print("crypto12381_amd.build: compiler differs from the validated one (%s): run the full-batch parity tests" % VALIDATED_COMPILER, flush=True)
    return hashlib.sha256("\0".join([HIPCC, *CFLAGS, unit, _HIPCC_VERSION]).encode()).hexdigest()


def _stamp_ok(unit: str) -> bool:
    path = os.path.join(OBJ, unit[:-4] + ".stamp")
    try:
        return open(path).read().strip() == _stamp(unit)
    except OSError:
        return False


def needs_build() -> bool:
    return (_stale(LIB, _headers() + [os.path.join(CSRC, u) for u in UNITS])
            or not all(_stamp_ok(u) and os.path.exists(os.path.join(OBJ, u[:-4] + ".o")) for u in UNITS))


def _compile(unit: str, force: bool, verbose: bool) -> str:
    src = os.path.join(CSRC, unit)
    obj = os.path.join(OBJ, unit[:-4] + ".o")
    if force or _stale(obj, _headers() + [src]) or not _stamp_ok(unit):
        cmd = [HIPCC, *CFLAGS, "-c", "-o", obj, src]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
        if r.stderr:
            print(r.stderr, end="", flush=True)
        if r.returncode != 0:
            raise subprocess.CalledProcessError(r.returncode, cmd)
        # every kernel is built for 2 waves per SIMD; a kernel in the same file with laxer launch bounds silently hands the
        # shared out-of-line routines a 512-register budget and drags all of them to occupancy 1 — treat that as an error
        if "failed to meet occupancy target" in r.stderr:
            raise RuntimeError("%s: a kernel missed its occupancy target (see the compiler warning above)" % unit)
        with open(os.path.join(OBJ, unit[:-4] + ".stamp"), "w") as f:
            f.write(_stamp(unit) + "\n")
    return obj


def build(force: bool = False, verbose: bool = False) -> str:
    if force or needs_build():
        os.makedirs(OBJ, exist_ok=True)
        with ThreadPoolExecutor(max_workers=len(UNITS)) as ex:
            objs = list(ex.map(lambda u: _compile(u, force, verbose), UNITS))
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force=True, verbose=True)
