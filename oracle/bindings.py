"""TEST INFRASTRUCTURE ONLY — ctypes bindings for the two CPU checkers.

* ``Oracle("port")``      -> oracle/liboracle.so       (our plain-C restatement, c12381_oracle.c)
* ``Oracle("reference")`` -> oracle/_ref/libc12381_ref.so (the real reference boundary, built by
  oracle/Makefile from /root/reference; travels to the GPU box as a prebuilt artefact)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Both libraries expose the same batch functions (prefix ``orc_`` / ``ref_``) over canonical
big-endian bytes, so a parity check is a ``bytes ==``.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_PORT = os.path.join(_HERE, "liboracle.so")
_REF = os.path.join(_HERE, "_ref", "libc12381_ref.so")
_SHIM = os.path.join(_HERE, "_ref", "libc12381_shimtest.so")     # same wrapper, boundary overridden by our HIP shim

_sz = ctypes.c_size_t


def build(force: bool = False) -> None:
    """Compile the checkers (make -C oracle). Building the checker is not using it."""
    if force or not os.path.exists(_PORT) or (os.path.isdir("/root/reference") and not os.path.exists(_REF)):
        subprocess.run(["make", "-C", _HERE, "-j8"], check=True, stdout=subprocess.DEVNULL)


def have_reference() -> bool:
    return os.path.exists(_REF)


def have_shim() -> bool:
    return os.path.exists(_SHIM)


class Oracle:
    def __init__(self, kind: str = "port"):
        if kind not in ("port", "reference", "shim"):
            raise ValueError(kind)
        path = {"port": _PORT, "reference": _REF, "shim": _SHIM}[kind]
        if not os.path.exists(path):
            build()
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.kind = kind
        self.lib = ctypes.CDLL(path)
        self.pfx = "orc_" if kind == "port" else "ref_"

    def _f(self, name):
        return getattr(self.lib, self.pfx + name)

    @staticmethod
    def _buf(n):
        return ctypes.create_string_buffer(max(n, 1))

    def _ck(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{self.pfx}{what} failed rc={rc}")

    # ---- constants
    def g1_generator(self) -> bytes:
        o = self._buf(96); self._ck(self._f("g1_generator")(o), "g1_generator"); return o.raw[:96]

    def g2_generator(self) -> bytes:
        o = self._buf(192); self._ck(self._f("g2_generator")(o), "g2_generator"); return o.raw[:192]

    # ---- Fp
    FP_OPS = {"mul": 0, "add": 1, "sub": 2, "sqr": 3, "neg": 4, "inv": 5, "sqrt": 6}

    def fp_op(self, op: str, a: bytes, b: bytes | None = None):
        n = len(a) // 48
        o, ok = self._buf(48 * n), self._buf(n)
        self._ck(self._f("fp_op_batch")(self.FP_OPS[op], _sz(n), a, b, o, ok), "fp_op_batch")
        return o.raw[:48 * n], ok.raw[:n]

    # ---- G1
    def g1_mul(self, pts: bytes, scalars: bytes, fmt: int = 49, nthreads: int = 1) -> bytes:
        n = len(pts) // 96
        o = self._buf(fmt * n)
        self._ck(self._f("g1_mul_batch")(_sz(n), pts, scalars, o, fmt, nthreads), "g1_mul_batch")
        return o.raw[:fmt * n]

    def g1_add(self, a: bytes, b: bytes, fmt: int = 96) -> bytes:
        n = len(a) // 96
        o = self._buf(fmt * n)
        self._ck(self._f("g1_add_batch")(_sz(n), a, b, o, fmt), "g1_add_batch")
        return o.raw[:fmt * n]

    def g1_decompress(self, c: bytes):
        n = len(c) // 49
        o, st = self._buf(96 * n), self._buf(n)
        self._ck(self._f("g1_decompress_batch")(_sz(n), c, o, st), "g1_decompress_batch")
        return o.raw[:96 * n], st.raw[:n]

    def g1_compress(self, a: bytes) -> bytes:
        n = len(a) // 96
        o = self._buf(49 * n)
        self._ck(self._f("g1_compress_batch")(_sz(n), a, o), "g1_compress_batch")
        return o.raw[:49 * n]

    def g1_msm(self, pts: bytes, scalars: bytes, fmt: int = 49, nthreads: int = 1) -> bytes:
        n = len(pts) // 96
        o = self._buf(fmt)
        self._ck(self._f("g1_msm")(_sz(n), pts, scalars, o, fmt, nthreads), "g1_msm")
        return o.raw[:fmt]

    def g1_sum_of_products(self, pts: bytes, scalars: bytes, fmt: int = 49) -> bytes:
        """sum_of_products -> ECP_muln: the true multiples sum [k_i mod r] P_i on any curve points"""
        n = len(pts) // 96
        o = self._buf(fmt)
        self._ck(self._f("g1_sum_of_products")(n, pts, scalars, o, fmt), "g1_sum_of_products")
        return o.raw[:fmt]

    # ---- G2
    def g2_mul(self, pts: bytes, scalars: bytes, fmt: int = 97, nthreads: int = 1) -> bytes:
        n = len(pts) // 192
        o = self._buf(fmt * n)
        self._ck(self._f("g2_mul_batch")(_sz(n), pts, scalars, o, fmt, nthreads), "g2_mul_batch")
        return o.raw[:fmt * n]

    def g2_add(self, a: bytes, b: bytes, fmt: int = 192) -> bytes:
        n = len(a) // 192
        o = self._buf(fmt * n)
        self._ck(self._f("g2_add_batch")(_sz(n), a, b, o, fmt), "g2_add_batch")
        return o.raw[:fmt * n]

    def g2_decompress(self, c: bytes):
        n = len(c) // 97
        o, st = self._buf(192 * n), self._buf(n)
        self._ck(self._f("g2_decompress_batch")(_sz(n), c, o, st), "g2_decompress_batch")
        return o.raw[:192 * n], st.raw[:n]

    def g2_compress(self, a: bytes) -> bytes:
        n = len(a) // 192
        o = self._buf(97 * n)
        self._ck(self._f("g2_compress_batch")(_sz(n), a, o), "g2_compress_batch")
        return o.raw[:97 * n]

    # ---- pairing / GT
    def pair(self, g1: bytes, g2: bytes, nthreads: int = 1) -> bytes:
        n = len(g1) // 96
        o = self._buf(576 * n)
        self._ck(self._f("pair_batch")(_sz(n), g1, g2, o, nthreads), "pair_batch")
        return o.raw[:576 * n]

    def miller_t(self, g1: bytes, g2: bytes, nthreads: int = 1) -> bytes:
        n = len(g1) // 96
        o = self._buf(576 * n)
        self._ck(self._f("miller_batch_t")(_sz(n), g1, g2, o, nthreads), "miller_batch_t"); return o.raw[:576 * n]

    def fexp_t(self, f: bytes, nthreads: int = 1) -> bytes:
        n = len(f) // 576
        o = self._buf(576 * n)
        self._ck(self._f("fexp_batch_t")(_sz(n), f, o, nthreads), "fexp_batch_t"); return o.raw[:576 * n]

    def bbs_plus_verify(self, g1: bytes, g2: bytes, h0: bytes, h: bytes, w: bytes, A: bytes, x: bytes, r: bytes, m: bytes, nthreads: int = 1) -> bytes:
        """examples/bbs-plus/src/bbs+.cpp:57-73 per signature; m is message-major (block i of signature j at (i*n + j)*32)"""
        n, nmsg = len(A) // 96, len(h) // 96
        o = self._buf(n)
        self._ck(self._f("bbs_plus_verify_batch")(_sz(n), _sz(nmsg), g1, g2, h0, h, w, A, x, r, m, o, nthreads), "bbs_plus_verify_batch")
        return o.raw[:n]

    def bbs_plus_verify_wire(self, g1_g2_h0: bytes, h49: bytes, pk97: bytes, sigs145: bytes, msgs: bytes, msg_len: int, nthreads: int = 1) -> bytes:
        """the same from the serialized forms: pp.g1_g2_h0 (195 B), pp.h (49 B each), pk (97 B), signatures (145 B), raw messages"""
        n, nh = len(sigs145) // 145, len(h49) // 49
        o = self._buf(n)
        self._ck(self._f("bbs_plus_verify_wire_batch")(_sz(n), _sz(nh), _sz(msg_len), g1_g2_h0, h49, pk97, sigs145, msgs, o, nthreads),
                 "bbs_plus_verify_wire_batch")
        return o.raw[:n]

    def encode_to_zp(self, msg: bytes) -> bytes:
        """encode_to<Zp> (zp_number.hpp:1011-1037): 31-byte units -> 32-byte scalars"""
        nblk = (len(msg) + 30) // 31
        o = self._buf(32 * nblk)
        self._ck(self._f("encode_to_zp")(_sz(len(msg)), msg, o), "encode_to_zp"); return o.raw[:32 * nblk]

    def miller(self, g1: bytes, g2: bytes) -> bytes:
        n = len(g1) // 96
        o = self._buf(576 * n)
        self._ck(self._f("miller_batch")(_sz(n), g1, g2, o), "miller_batch")
        return o.raw[:576 * n]

    def fexp(self, f: bytes) -> bytes:
        n = len(f) // 576
        o = self._buf(576 * n)
        self._ck(self._f("fexp_batch")(_sz(n), f, o), "fexp_batch")
        return o.raw[:576 * n]

    def pair_eq(self, a1: bytes, a2: bytes, b1: bytes, b2: bytes, nthreads: int = 1) -> bytes:
        n = len(a1) // 96
        o = self._buf(n)
        self._ck(self._f("pair_eq_batch")(_sz(n), a1, a2, b1, b2, o, nthreads), "pair_eq_batch")
        return o.raw[:n]

    def pair2(self, a1: bytes, a2: bytes, b1: bytes, b2: bytes) -> bytes:
        n = len(a1) // 96
        o = self._buf(576 * n)
        self._ck(self._f("pair2_batch")(_sz(n), a1, a2, b1, b2, o), "pair2_batch")
        return o.raw[:576 * n]

    GT_OPS = {"mul": 0, "conj": 1, "pow": 2}

    def gt_op(self, op: str, a: bytes, b: bytes | None = None) -> bytes:
        n = len(a) // 576
        o = self._buf(576 * n)
        self._ck(self._f("gt_op_batch")(self.GT_OPS[op], _sz(n), a, b, o), "gt_op_batch")
        return o.raw[:576 * n]

    # ---- hash-to-G1 and Zp helpers
    def g1_from_hash(self, digests: bytes, fmt: int = 96) -> bytes:
        n = len(digests) // 64
        o = self._buf(fmt * n)
        self._ck(self._f("g1_from_hash_batch")(_sz(n), digests, o, fmt), "g1_from_hash_batch")
        return o.raw[:fmt * n]

    def g1_map_to_point(self, u48: bytes) -> bytes:
        if self.kind == "port":
            raise RuntimeError("map_to_point alone is exposed by the reference build only")
        n = len(u48) // 48
        o = self._buf(96 * n)
        self._ck(self.lib.ref_g1_map_to_point_batch(_sz(n), u48, o), "g1_map_to_point_batch")
        return o.raw[:96 * n]

    ZP_OPS = {"mul": 0, "add": 1, "sub": 2, "neg": 3, "inv": 4}

    def zp_op(self, op: str, a: bytes, b: bytes | None = None) -> bytes:
        n = len(a) // 32
        o = self._buf(32 * n)
        self._ck(self._f("zp_op_batch")(self.ZP_OPS[op], _sz(n), a, b, o), "zp_op_batch")
        return o.raw[:32 * n]

    def zp_from_hash(self, digests: bytes) -> bytes:
        n = len(digests) // 64
        o = self._buf(32 * n)
        self._ck(self._f("zp_from_hash_batch")(_sz(n), digests, o), "zp_from_hash_batch")
        return o.raw[:32 * n]

    # ---- reference-only helpers
    def random_scalars(self, seed: bytes, n: int) -> bytes:
        if self.kind != "reference":
            raise RuntimeError("the seeded MIRACL CSPRNG stream exists only in the reference build")
        o = self._buf(32 * n)
        self._ck(self.lib.ref_random_scalars(seed, len(seed), _sz(n), o), "random_scalars")
        return o.raw[:32 * n]
