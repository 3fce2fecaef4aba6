"""Build the HIP shared libraries in-tree for gfx950.

    crypto12381_amd/lib/libc12381_hip.so       the product: one code path, reads no environment variable
    crypto12381_amd/lib/libc12381_probe.so     bench-only clock probe (csrc/microbench/clock_probe.hip), loaded by bench.py alone
    crypto12381_amd/lib/libc12381_hip_exp.so   the same sources with -DC12381_EXPERIMENTS: tuning / diagnostic switches (C12381_*
                                               environment variables) and the superseded one-lane pairing kernels, for tools/, A/B
                                               runs and tests/test_gpu_variants.py (selected with C12381_LIB in the Python binding)

One hipcc compile per translation unit (csrc/*.hip, in parallel), then one link per library; the experiments library shares every
object whose source does not look at the define.  Objects are cached under lib/obj/ and reused only when (a) no source or header is
newer and (b) the stamp next to the object — a hash of the exact command line, of the compiler-affecting environment and of
`hipcc --version` — matches: an object built with other flags or another compiler is never linked in."""
from __future__ import annotations

import hashlib
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
UNITS = ["c12381_hip.hip", "k_g1.hip", "k_g2gt.hip", "k_g2h.hip", "k_pair3.hip", "k_hash_zp.hip", "k_fixed.hip"]
EXP_UNITS = ["c12381_hip.hip", "k_g2gt.hip"]                # the units that test C12381_EXPERIMENTS
LIB = os.path.join(HERE, "lib", "libc12381_hip.so")
LIB_EXP = os.path.join(HERE, "lib", "libc12381_hip_exp.so")
# bench-only: the one-lane clock probe bench.py runs beside each dominant kernel (csrc/microbench/clock_probe.hip); not linked into the product
LIB_PROBE = os.path.join(HERE, "lib", "libc12381_probe.so")
PROBE_SRC = os.path.join(CSRC, "microbench", "clock_probe.hip")
OBJ = os.path.join(HERE, "lib", "obj")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -fno-optimize-sibling-calls: LLVM drops the "callee saves nothing" treatment of an internal function as soon as one
# call site carries a `tail` marker (which every call passing only non-stack pointers gets); without the marker the
# big out-of-line field routines do not save ~65 callee-saved VGPRs in their prologues (130 scratch instructions a call)
# max-ilp scheduling: every kernel here runs at a FIXED occupancy (launch bounds: 2 waves per SIMD), so the default strategy's
# effort to raise occupancy buys nothing, while scheduling for ILP shortens the dependent multiply-add chains (A/B on MI355X,
# round 1: pairing kernel 22.8 -> 22.1 ms).  NOT combined with -amdgpu-use-amdgpu-trackers=1: that (experimental)
# option gained another 1 %, but together with max-ilp an experimental variant of the pairing routines (Fp4 squarings as
# calls) returned wrong values for a few lanes of a full-size batch while passing every small test — not worth the risk.
# -opt-disable=reassociate (round 4): LLVM's Reassociate pass orders the operands of every column sum by rank and therefore adds the
# carry of the previous column LAST — each column of a Montgomery product is summed from zero and joined to the carry by one
# v_lshl_add_u64 (26-30 per reduction, 3.4-5 % of every kernel's vector instructions), and the m * p terms become a chain of their
# own.  With the pass off the source order survives: ONE linear v_mad_i64_i32 chain per column sequence through one accumulator pair —
# fp_mul 488 -> 463 vector instructions (the hand count is 460) and 84 -> 44 registers.  A/B in one session, every output digest equal
# (profiles/r04_ab_noreassoc.txt): G1 -3.0 %, G2 -4.3 %, pairing -2.2 %, MSM -2.2 %, BBS+ -2.5 %.  (Inline-asm multiply-adds, the other
# way to pin the chain, cost one s_nop per instruction: the hazard recognizer pads every asm -> dependent-instruction edge.)
CFLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-fno-optimize-sibling-calls",
          "-mllvm", "-amdgpu-sched-strategy=max-ilp", "-mllvm", "-opt-disable=reassociate"]
# With the linear chains of -opt-disable=reassociate the G1 scalar-multiplication kernels no longer gain from max-ilp: the default
# strategy is 1.2 % faster there (three interleaved rounds, digests equal, profiles/r04_ab_sched_per_unit.txt: G1 2^20 24.63 -> 24.33 ms
# mean, MSM unchanged), while G2 (+1.4 %), the Miller loop and the final exponentiation (+1.6 %) still lose without it.
DEFAULT_SCHED_UNITS = ("k_g1.hip",)


def unit_cflags(unit: str):
    if unit in DEFAULT_SCHED_UNITS:
        return [f for i, f in enumerate(CFLAGS) if "amdgpu-sched-strategy" not in f and not (f == "-mllvm" and "amdgpu-sched-strategy" in CFLAGS[i + 1])]
    return list(CFLAGS)
# Round 2 reproduced the wrong-values event with a rebuilt variant (docs/lab_notes.md 5b): whole wavefront groups wrong, plain grid as
# well as queue, only with -amdgpu-use-amdgpu-trackers=1; the same source without it is exact on every lane.  The option is
# refused outright — in CFLAGS and in every environment variable through which hipcc / clang accept extra flags — and the
# compiler the full-batch parity tests were run with is recorded: another one STOPS the build (3-4 % of every kernel hang on an LLVM-internal
# pass gate of exactly this compiler, and the wrong-lanes event was a build-variant effect) unless C12381_ALLOW_UNVALIDATED_COMPILER=1 says
# that the caller will run tests/test_gpu_full_batch.py on the result.
FORBIDDEN_FLAGS = ("amdgpu-use-amdgpu-trackers",)
FLAG_ENV = ("HIPCC_COMPILE_FLAGS_APPEND", "HIPCC_LINK_FLAGS_APPEND", "HIP_CLANG_FLAGS", "CCC_OVERRIDE_OPTIONS", "HIPCC")
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


def _flag_env() -> str:
    return "\0".join("%s=%s" % (k, os.environ.get(k, "")) for k in FLAG_ENV)


def _check_flags() -> None:
    seen = " ".join(CFLAGS) + " " + " ".join(os.environ.get(k, "") for k in FLAG_ENV)
    for bad in FORBIDDEN_FLAGS:
        if bad in seen:
            raise RuntimeError("build flag known to produce wrong pairings at full size: %s (docs/lab_notes.md 5b) — found in CFLAGS or in one of %s"
                               % (bad, ", ".join(FLAG_ENV)))


def _hipcc_version():
    """`hipcc --version`, or None when there is no compiler on this machine (a GPU box that received the prebuilt library)"""
    global _HIPCC_VERSION
    if _HIPCC_VERSION is None:
        try:
            _HIPCC_VERSION = subprocess.run([HIPCC, "--version"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
        except OSError:
            _HIPCC_VERSION = ""
        if _HIPCC_VERSION and VALIDATED_COMPILER not in _HIPCC_VERSION:
            if os.environ.get("C12381_ALLOW_UNVALIDATED_COMPILER") != "1":
                raise RuntimeError("crypto12381_amd.build: %s is not the compiler this library was validated with (%s).  Set "
                                   "C12381_ALLOW_UNVALIDATED_COMPILER=1 to build anyway, then run tests/test_gpu_full_batch.py (every lane of the "
                                   "full-size batches against the reference) before trusting the result.\n%s" % (HIPCC, VALIDATED_COMPILER, _HIPCC_VERSION))
            print("crypto12381_amd.build: compiler differs from the validated one (%s) — allowed by C12381_ALLOW_UNVALIDATED_COMPILER: run "
                  "tests/test_gpu_full_batch.py" % VALIDATED_COMPILER, flush=True)
    return _HIPCC_VERSION or None


def _obj_name(unit: str, exp: bool) -> str:
    return unit[:-4] + ("_exp" if exp else "")


def _stamp(unit: str, exp: bool) -> str:
    """hash of everything besides the sources that decides what the object contains"""
    _check_flags()
    return hashlib.sha256("\0".join([HIPCC, *unit_cflags(unit), "exp" if exp else "", unit, _flag_env(), _hipcc_version() or ""]).encode()).hexdigest()


def _stamp_ok(unit: str, exp: bool) -> bool:
    path = os.path.join(OBJ, _obj_name(unit, exp) + ".stamp")
    try:
        return open(path).read().strip() == _stamp(unit, exp)
    except OSError:
        return False


def _jobs():
    return [(u, False) for u in UNITS] + [(u, True) for u in EXP_UNITS]


def needs_build() -> bool:
    if _hipcc_version() is None:
        # no compiler here: prebuilt libraries are taken as they are (they were stamped where they were built); a missing product or experiments
        # library is an error, the bench-only clock probe is optional (bench.py runs without it: roofline.issue then has no clock)
        missing = [p for p in (LIB, LIB_EXP) if not os.path.exists(p)]
        if not missing:
            if not os.path.exists(LIB_PROBE):
                print("crypto12381_amd.build: no compiler and no prebuilt %s (bench-only clock probe): continuing without it" % LIB_PROBE, flush=True)
            return False
        raise RuntimeError("crypto12381_amd.build: %s not found and no prebuilt %s — build the library where hipcc is available" % (HIPCC, ", ".join(missing)))
    srcs = _headers() + [os.path.join(CSRC, u) for u in UNITS]
    return (_stale(LIB, srcs) or _stale(LIB_EXP, srcs) or _stale(LIB_PROBE, [PROBE_SRC])
            or not all(_stamp_ok(u, e) and os.path.exists(os.path.join(OBJ, _obj_name(u, e) + ".o")) for u, e in _jobs()))


_PASS_GATE_OK = None


def _check_pass_gate() -> None:
    """`-mllvm -opt-disable=<pass>` is a debugging option of LLVM's pass manager, not a stable interface: probe it once on an empty translation unit
    so that a toolchain without it fails HERE with a clear message, not with 'Unknown command line argument' in the middle of seven compiles.
    There is no silent fallback to the plain flags: the shipped instruction counts, the issue figures in profiles/ and the every-lane validation
    all belong to the build WITH the gate (profiles/r04_ab_noreassoc.txt is the digest-equal A/B against the build without it)."""
    global _PASS_GATE_OK
    if _PASS_GATE_OK is None:
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-x", "hip", "-c", "-o", os.devnull, "-mllvm", "-opt-disable=reassociate", "--cuda-device-only", "-"],
                           input="", stderr=subprocess.PIPE, stdout=subprocess.PIPE, text=True)
        _PASS_GATE_OK = r.returncode == 0
        if not _PASS_GATE_OK:
            raise RuntimeError("crypto12381_amd.build: %s does not accept `-mllvm -opt-disable=reassociate` (an LLVM pass gate of AMD clang 22, ROCm 7.2): "
                               "this library is built and validated with it — use that compiler.\n%s" % (HIPCC, r.stderr[-800:]))


def _compile(unit: str, exp: bool, force: bool, verbose: bool) -> str:
    src = os.path.join(CSRC, unit)
    obj = os.path.join(OBJ, _obj_name(unit, exp) + ".o")
    if force or _stale(obj, _headers() + [src]) or not _stamp_ok(unit, exp):
        cmd = [HIPCC, *unit_cflags(unit), *(["-DC12381_EXPERIMENTS"] if exp else []), "-c", "-o", obj, src]
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
        with open(os.path.join(OBJ, _obj_name(unit, exp) + ".stamp"), "w") as f:
            f.write(_stamp(unit, exp) + "\n")
    return obj


def build(force: bool = False, verbose: bool = False) -> str:
    if force or needs_build():
        _check_flags()
        _check_pass_gate()
        os.makedirs(OBJ, exist_ok=True)
        jobs = _jobs()
        with ThreadPoolExecutor(max_workers=8) as ex:
            objs = dict(zip(jobs, ex.map(lambda j: _compile(j[0], j[1], force, verbose), jobs)))
        for lib, exp in ((LIB, False), (LIB_EXP, True)):
            link = [objs[(u, exp and u in EXP_UNITS)] for u in UNITS]
            cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *link]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
        if force or _stale(LIB_PROBE, [PROBE_SRC]):
            cmd = [HIPCC, "-O2", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PROBE, PROBE_SRC]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force=True, verbose=True)
