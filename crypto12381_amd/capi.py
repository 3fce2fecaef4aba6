"""ctypes binding of the C ABI in include/c12381_hip.h.

This module is plumbing: it loads the in-tree HIP library and moves bytes.  There is NO CPU
fallback — if the library is missing or no HIP device is usable, construction fails loudly.
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libc12381_hip.so")
# The binding reads NO environment variable.  Tools, A/B runs and the variant tests that need another build of the same ABI (the
# experiments library, tools/build_variant.sh outputs) say so explicitly with use_library(path) before the first Context — see
# tools/libsel.py, which is where their C12381_LIB convention lives now.

E_ARG, E_HIP, E_POINT, E_NOMEM, E_INTERNAL = -1, -2, -3, -4, -5
F_IN_SUBGROUP = 1
F_MILLER_ONLY = 2
F_COMPRESSED_IN = 4


class C12381Error(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"c12381 error {code}: {msg}")
        self.code = code


_lib = None
_lib_path = LIB_PATH


def use_library(path: str) -> None:
    """Select another build of the same C ABI (experiments / A-B library) for this process; must precede the first load."""
    global _lib_path
    if _lib is not None and os.path.abspath(path) != os.path.abspath(_lib_path):
        raise RuntimeError("use_library(%s): %s is already loaded in this process" % (path, _lib_path))
    _lib_path = path


def load_library() -> ctypes.CDLL:
    """Load libc12381_hip.so (built by crypto12381_amd.build / __graft_entry__.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(_lib_path):
            raise FileNotFoundError(
                f"{_lib_path} not found: the HIP extension is not built (run `python -m crypto12381_amd.build`). "
                "There is no CPU fallback.")
        lib = ctypes.CDLL(_lib_path)
        vp, sz, ci = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
        lib.c12381_create.argtypes = [ci, ctypes.POINTER(vp)]
        lib.c12381_destroy.argtypes = [vp]
        lib.c12381_destroy.restype = None
        lib.c12381_last_error.argtypes = [vp]
        lib.c12381_last_error.restype = ctypes.c_char_p
        lib.c12381_set_stream.argtypes = [vp, vp]
        lib.c12381_sync.argtypes = [vp]
        lib.c12381_wait_event.argtypes = [vp, vp]
        lib.c12381_record_event.argtypes = [vp, vp]
        lib.c12381_trim.argtypes = [vp]
        lib.c12381_profile.argtypes = [vp, ci]
        lib.c12381_profile_read.argtypes = [vp, ci, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_uint64)]
        for name in ("c12381_fp_op_batch", "c12381_fp_op_batch_dev"):
            getattr(lib, name).argtypes = [vp, ci, sz, vp, vp, vp]
        lib.c12381_fp_mulchain_dev.argtypes = [vp, sz, ci, vp, vp, vp]
        for name in ("c12381_g1_mul_batch", "c12381_g1_mul_batch_dev", "c12381_g1_add_batch", "c12381_g1_msm",
                     "c12381_g1_msm_dev"):
            getattr(lib, name).argtypes = [vp, sz, vp, vp, vp, ci]
        for name in ("c12381_g2_mul_batch", "c12381_g2_mul_batch_dev", "c12381_g2_add_batch"):
            getattr(lib, name).argtypes = [vp, sz, vp, vp, vp, ci]
        for name in ("c12381_g1_sum", "c12381_g1_sum_dev"):
            getattr(lib, name).argtypes = [vp, sz, vp, vp, ci]
        for name in ("c12381_g1_sum_of_products", "c12381_g1_sum_of_products_dev"):
            getattr(lib, name).argtypes = [vp, sz, vp, vp, vp, ci]
        for name in ("c12381_g2_msm", "c12381_g2_msm_dev"):
            getattr(lib, name).argtypes = [vp, sz, vp, vp, vp, ci]
        for name in ("c12381_g1_mul_batch_flags", "c12381_g1_mul_batch_flags_dev", "c12381_g2_mul_batch_flags", "c12381_g2_mul_batch_flags_dev",
                     "c12381_g1_msm_flags", "c12381_g1_msm_flags_dev"):
            getattr(lib, name).argtypes = [vp, sz, vp, vp, vp, ci, ctypes.c_uint]
        for name in ("c12381_pair_batch_flags", "c12381_pair_batch_flags_dev"):
            getattr(lib, name).argtypes = [vp, sz, vp, vp, vp, ctypes.c_uint]
        for name in ("c12381_pair_batch", "c12381_pair_batch_dev"):
            getattr(lib, name).argtypes = [vp, sz, vp, vp, vp]
        for name in ("c12381_pair_product_batch", "c12381_pair_product_batch_dev"):
            getattr(lib, name).argtypes = [vp, sz, ci, vp, vp, vp, ctypes.c_uint]
        for name in ("c12381_pair_eq_batch", "c12381_pair_eq_batch_dev"):
            getattr(lib, name).argtypes = [vp, sz, vp, vp, vp, vp, vp]
        for name in ("c12381_g1_decompress_batch", "c12381_g2_decompress_batch"):
            getattr(lib, name).argtypes = [vp, sz, vp, vp, vp]
        for name in ("c12381_miller_batch", "c12381_miller_batch_dev"):
            getattr(lib, name).argtypes = [vp, sz, vp, vp, vp]
        for name in ("c12381_fexp_batch", "c12381_fexp_batch_dev", "c12381_gt_is_unity_batch", "c12381_gt_is_unity_batch_dev"):
            getattr(lib, name).argtypes = [vp, sz, vp, vp]
        for name in ("c12381_gt_op_batch", "c12381_gt_op_batch_dev"):
            getattr(lib, name).argtypes = [vp, ci, sz, vp, vp, vp]
        for name in ("c12381_g1_from_hash_batch", "c12381_g1_from_hash_batch_dev"):
            getattr(lib, name).argtypes = [vp, sz, vp, vp, ci]
        for name in ("c12381_g1_mul_fixed_batch", "c12381_g1_mul_fixed_batch_dev", "c12381_g2_mul_fixed_batch", "c12381_g2_mul_fixed_batch_dev"):
            getattr(lib, name).argtypes = [vp, sz, vp, vp, vp, ci]
        for name in ("c12381_pair_fixed_g2_batch", "c12381_pair_fixed_g2_batch_dev"):
            getattr(lib, name).argtypes = [vp, sz, vp, vp, vp]
        for name in ("c12381_bbs_plus_sign_batch", "c12381_bbs_plus_sign_batch_dev"):
            getattr(lib, name).argtypes = [vp, sz, sz, vp, vp, vp, vp, vp, vp, vp, vp]
        for name in ("c12381_g1_decompress_batch_dev", "c12381_g2_decompress_batch_dev"):
            getattr(lib, name).argtypes = [vp, sz, vp, vp, vp]
        lib.c12381_g1_msm_multi.argtypes = [ctypes.POINTER(vp), ci, sz, vp, vp, vp, ci]
        lib.c12381_g1_map_to_point_batch.argtypes = [vp, sz, vp, vp]
        lib.c12381_g1_clear_cofactor_batch.argtypes = [vp, sz, vp, vp]
        for name in ("c12381_zp_op_batch", "c12381_zp_op_batch_dev"):
            getattr(lib, name).argtypes = [vp, ci, sz, vp, vp, vp]
        lib.c12381_zp_from_hash_batch.argtypes = [vp, sz, vp, vp]
        for name in ("c12381_zp_inner_product", "c12381_zp_inner_product_dev"):
            getattr(lib, name).argtypes = [vp, sz, vp, vp, vp]
        for name in ("c12381_bbs_plus_verify_batch", "c12381_bbs_plus_verify_batch_dev"):
            getattr(lib, name).argtypes = [vp, sz, sz] + [vp] * 10
        for name in ("c12381_bbs_plus_verify_wire_batch", "c12381_bbs_plus_verify_wire_batch_dev"):
            getattr(lib, name).argtypes = [vp, sz, sz, sz] + [vp] * 6
        for name in ("c12381_bbs_plus_verify_aggregate", "c12381_bbs_plus_verify_aggregate_dev"):
            getattr(lib, name).argtypes = [vp, sz, sz] + [vp] * 11
        _lib = lib
    return _lib


def _p(b):
    """bytes / ctypes buffer / int device address -> c_void_p"""
    if b is None:
        return None
    if isinstance(b, int):
        return ctypes.c_void_p(b)
    if isinstance(b, (bytes, bytearray)):
        return ctypes.cast(ctypes.c_char_p(bytes(b)), ctypes.c_void_p)
    return ctypes.cast(b, ctypes.c_void_p)


class Context:
    """One context per process/GPU (mirrors c12381_ctx)."""

    def __init__(self, device: int = 0):
        self.lib = load_library()
        self.h = ctypes.c_void_p()
        rc = self.lib.c12381_create(device, ctypes.byref(self.h))
        if rc != 0:
            self.h = None
            raise C12381Error(rc, "c12381_create failed (no usable HIP device?)")

    def close(self):
        if getattr(self, "h", None):
            self.lib.c12381_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc, allow_point=False):
        if rc == 0 or (allow_point and rc == E_POINT):
            return rc
        raise C12381Error(rc, (self.lib.c12381_last_error(self.h) or b"").decode())

    def set_stream(self, hip_stream):
        self._ck(self.lib.c12381_set_stream(self.h, _p(hip_stream)))

    def sync(self) -> int:
        return self._ck(self.lib.c12381_sync(self.h), allow_point=True)

    def trim(self):
        """free the context's device workspaces (they grow to the largest call served; the next call allocates what it needs)"""
        self._ck(self.lib.c12381_trim(self.h))

    def wait_event(self, hip_event):
        """everything this context launches from now on waits for `hip_event` (a hipEvent_t as an integer, e.g. torch.cuda.Event.cuda_event
        after record()): the device-side edge from the stream that produced a _dev call's inputs to the context's stream"""
        self._ck(self.lib.c12381_wait_event(self.h, _p(hip_event)))

    def record_event(self, hip_event):
        """record `hip_event` on the context's stream behind everything launched so far: the edge to a stream that consumes the outputs"""
        self._ck(self.lib.c12381_record_event(self.h, _p(hip_event)))

    def profile(self, enable: bool):
        self._ck(self.lib.c12381_profile(self.h, 1 if enable else 0))

    def profile_read(self, kind: int):
        ms, cnt = ctypes.c_double(), ctypes.c_uint64()
        self._ck(self.lib.c12381_profile_read(self.h, kind, ctypes.byref(ms), ctypes.byref(cnt)))
        return ms.value, cnt.value

    # ---- host-buffer entry points (bytes in, bytes out)
    FP_OPS = {"mul": 0, "add": 1, "sub": 2, "sqr": 3, "neg": 4, "inv": 5}

    def fp_op(self, op: str, a: bytes, b: bytes | None = None) -> bytes:
        n = len(a) // 48
        out = ctypes.create_string_buffer(max(48 * n, 1))
        self._ck(self.lib.c12381_fp_op_batch(self.h, self.FP_OPS[op], n, _p(a), _p(b), _p(out)))
        return out.raw[:48 * n]

    def g1_mul(self, pts: bytes, scalars: bytes, fmt: int = 49, strict: bool = True) -> bytes:
        n = len(pts) // 96
        out = ctypes.create_string_buffer(max(fmt * n, 1))
        self._ck(self.lib.c12381_g1_mul_batch(self.h, n, _p(pts), _p(scalars), _p(out), fmt), allow_point=not strict)
        return out.raw[:fmt * n]

    def g1_mul_flags(self, pts: bytes, scalars: bytes, fmt: int = 49, flags: int = 0, strict: bool = True) -> bytes:
        n = len(scalars) // 32
        out = ctypes.create_string_buffer(max(fmt * n, 1))
        self._ck(self.lib.c12381_g1_mul_batch_flags(self.h, n, _p(pts), _p(scalars), _p(out), fmt, flags), allow_point=not strict)
        return out.raw[:fmt * n]

    def g2_mul_flags(self, pts: bytes, scalars: bytes, fmt: int = 97, flags: int = 0, strict: bool = True) -> bytes:
        n = len(scalars) // 32
        out = ctypes.create_string_buffer(max(fmt * n, 1))
        self._ck(self.lib.c12381_g2_mul_batch_flags(self.h, n, _p(pts), _p(scalars), _p(out), fmt, flags), allow_point=not strict)
        return out.raw[:fmt * n]

    def g1_sum(self, pts: bytes, fmt: int = 49) -> bytes:
        n = len(pts) // 96
        out = ctypes.create_string_buffer(fmt)
        self._ck(self.lib.c12381_g1_sum(self.h, n, _p(pts), _p(out), fmt))
        return out.raw[:fmt]

    def g1_sum_dev(self, n, pts_ptr, out_ptr, fmt=49):
        self._ck(self.lib.c12381_g1_sum_dev(self.h, n, _p(pts_ptr), _p(out_ptr), fmt))

    def g1_msm_flags(self, pts: bytes, scalars: bytes, fmt: int = 49, flags: int = 0, strict: bool = True) -> bytes:
        n = len(scalars) // 32
        out = ctypes.create_string_buffer(fmt)
        self._ck(self.lib.c12381_g1_msm_flags(self.h, n, _p(pts), _p(scalars), _p(out), fmt, flags), allow_point=not strict)
        return out.raw[:fmt]

    def pair_flags(self, g1: bytes, g2: bytes, flags: int = 0, strict: bool = True) -> bytes:
        n = len(g1) // (49 if flags & F_COMPRESSED_IN else 96)
        out = ctypes.create_string_buffer(max(576 * n, 1))
        self._ck(self.lib.c12381_pair_batch_flags(self.h, n, _p(g1), _p(g2), _p(out), flags), allow_point=not strict)
        return out.raw[:576 * n]

    def g1_mul_flags_dev(self, n, pts_ptr, sc_ptr, out_ptr, fmt=49, flags=0):
        self._ck(self.lib.c12381_g1_mul_batch_flags_dev(self.h, n, _p(pts_ptr), _p(sc_ptr), _p(out_ptr), fmt, flags))

    def g2_mul_flags_dev(self, n, pts_ptr, sc_ptr, out_ptr, fmt=97, flags=0):
        self._ck(self.lib.c12381_g2_mul_batch_flags_dev(self.h, n, _p(pts_ptr), _p(sc_ptr), _p(out_ptr), fmt, flags))

    def g1_add(self, a: bytes, b: bytes, fmt: int = 96, strict: bool = True) -> bytes:
        n = len(a) // 96
        out = ctypes.create_string_buffer(max(fmt * n, 1))
        self._ck(self.lib.c12381_g1_add_batch(self.h, n, _p(a), _p(b), _p(out), fmt), allow_point=not strict)
        return out.raw[:fmt * n]

    def g1_msm(self, pts: bytes, scalars: bytes, fmt: int = 49) -> bytes:
        n = len(pts) // 96
        out = ctypes.create_string_buffer(fmt)
        self._ck(self.lib.c12381_g1_msm(self.h, n, _p(pts), _p(scalars), _p(out), fmt))
        return out.raw[:fmt]

    def g2_mul(self, pts: bytes, scalars: bytes, fmt: int = 97, strict: bool = True) -> bytes:
        n = len(pts) // 192
        out = ctypes.create_string_buffer(max(fmt * n, 1))
        self._ck(self.lib.c12381_g2_mul_batch(self.h, n, _p(pts), _p(scalars), _p(out), fmt), allow_point=not strict)
        return out.raw[:fmt * n]

    def g1_sum_of_products(self, pts: bytes, scalars: bytes, fmt: int = 49) -> bytes:
        n = len(pts) // 96
        out = ctypes.create_string_buffer(fmt)
        self._ck(self.lib.c12381_g1_sum_of_products(self.h, n, _p(pts) if n else None, _p(scalars) if n else None, _p(out), fmt))
        return out.raw[:fmt]

    def g2_msm(self, pts: bytes, scalars: bytes | None, fmt: int = 97) -> bytes:
        n = len(pts) // 192
        out = ctypes.create_string_buffer(fmt)
        self._ck(self.lib.c12381_g2_msm(self.h, n, _p(pts) if n else None, _p(scalars) if scalars else None, _p(out), fmt))
        return out.raw[:fmt]

    def g2_add(self, a: bytes, b: bytes, fmt: int = 192, strict: bool = True) -> bytes:
        n = len(a) // 192
        out = ctypes.create_string_buffer(max(fmt * n, 1))
        self._ck(self.lib.c12381_g2_add_batch(self.h, n, _p(a), _p(b), _p(out), fmt), allow_point=not strict)
        return out.raw[:fmt * n]

    def pair(self, g1: bytes, g2: bytes, strict: bool = True) -> bytes:
        n = len(g1) // 96
        out = ctypes.create_string_buffer(max(576 * n, 1))
        self._ck(self.lib.c12381_pair_batch(self.h, n, _p(g1), _p(g2), _p(out)), allow_point=not strict)
        return out.raw[:576 * n]

    def pair_product(self, g1s: bytes, g2s: bytes, k: int, flags: int = 0) -> bytes:
        """g1s / g2s: k argument-major arrays of n points; returns n GT (or Miller) values"""
        n = len(g1s) // (96 * k)
        out = ctypes.create_string_buffer(max(576 * n, 1))
        self._ck(self.lib.c12381_pair_product_batch(self.h, n, k, _p(g1s), _p(g2s), _p(out), flags))
        return out.raw[:576 * n]

    def pair_fixed_g2(self, g1: bytes, g2_one: bytes, strict: bool = True) -> bytes:
        n = len(g1) // 96
        out = ctypes.create_string_buffer(max(576 * n, 1))
        self._ck(self.lib.c12381_pair_fixed_g2_batch(self.h, n, _p(g1), _p(g2_one), _p(out)), allow_point=not strict)
        return out.raw[:576 * n]

    def pair_fixed_g2_dev(self, n, g1_ptr, g2_ptr, gt_ptr):
        self._ck(self.lib.c12381_pair_fixed_g2_batch_dev(self.h, n, _p(g1_ptr), _p(g2_ptr), _p(gt_ptr)))

    def pair_eq(self, a1: bytes, a2: bytes, b1: bytes, b2: bytes, strict: bool = True) -> bytes:
        n = len(a1) // 96
        out = ctypes.create_string_buffer(max(n, 1))
        self._ck(self.lib.c12381_pair_eq_batch(self.h, n, _p(a1), _p(a2), _p(b1), _p(b2), _p(out)), allow_point=not strict)
        return out.raw[:n]

    def g1_decompress(self, c: bytes):
        n = len(c) // 49
        out, st = ctypes.create_string_buffer(max(96 * n, 1)), ctypes.create_string_buffer(max(n, 1))
        self._ck(self.lib.c12381_g1_decompress_batch(self.h, n, _p(c), _p(out), _p(st)))
        return out.raw[:96 * n], st.raw[:n]

    def g2_decompress(self, c: bytes):
        n = len(c) // 97
        out, st = ctypes.create_string_buffer(max(192 * n, 1)), ctypes.create_string_buffer(max(n, 1))
        self._ck(self.lib.c12381_g2_decompress_batch(self.h, n, _p(c), _p(out), _p(st)))
        return out.raw[:192 * n], st.raw[:n]

    def g1_from_hash(self, digests: bytes, fmt: int = 96) -> bytes:
        n = len(digests) // 64
        out = ctypes.create_string_buffer(max(fmt * n, 1))
        self._ck(self.lib.c12381_g1_from_hash_batch(self.h, n, _p(digests), _p(out), fmt))
        return out.raw[:fmt * n]

    def g1_mul_fixed(self, base: bytes, scalars: bytes, fmt: int = 49, strict: bool = True) -> bytes:
        n = len(scalars) // 32
        out = ctypes.create_string_buffer(max(fmt * n, 1))
        self._ck(self.lib.c12381_g1_mul_fixed_batch(self.h, n, _p(base), _p(scalars), _p(out), fmt), allow_point=not strict)
        return out.raw[:fmt * n]

    def g2_mul_fixed(self, base: bytes, scalars: bytes, fmt: int = 97, strict: bool = True) -> bytes:
        n = len(scalars) // 32
        out = ctypes.create_string_buffer(max(fmt * n, 1))
        self._ck(self.lib.c12381_g2_mul_fixed_batch(self.h, n, _p(base), _p(scalars), _p(out), fmt), allow_point=not strict)
        return out.raw[:fmt * n]

    def g1_mul_fixed_dev(self, n, base_ptr, sc_ptr, out_ptr, fmt=49):
        self._ck(self.lib.c12381_g1_mul_fixed_batch_dev(self.h, n, _p(base_ptr), _p(sc_ptr), _p(out_ptr), fmt))

    def g2_mul_fixed_dev(self, n, base_ptr, sc_ptr, out_ptr, fmt=97):
        self._ck(self.lib.c12381_g2_mul_fixed_batch_dev(self.h, n, _p(base_ptr), _p(sc_ptr), _p(out_ptr), fmt))

    def g1_map_to_point(self, u48: bytes) -> bytes:
        n = len(u48) // 48
        out = ctypes.create_string_buffer(max(96 * n, 1))
        self._ck(self.lib.c12381_g1_map_to_point_batch(self.h, n, _p(u48), _p(out)))
        return out.raw[:96 * n]

    def g1_clear_cofactor(self, pts: bytes, strict: bool = True) -> bytes:
        n = len(pts) // 96
        out = ctypes.create_string_buffer(max(96 * n, 1))
        self._ck(self.lib.c12381_g1_clear_cofactor_batch(self.h, n, _p(pts), _p(out)), allow_point=not strict)
        return out.raw[:96 * n]

    ZP_OPS = {"mul": 0, "add": 1, "sub": 2, "neg": 3, "inv": 4}

    def zp_op(self, op: str, a: bytes, b: bytes | None = None) -> bytes:
        n = len(a) // 32
        out = ctypes.create_string_buffer(max(32 * n, 1))
        self._ck(self.lib.c12381_zp_op_batch(self.h, self.ZP_OPS[op], n, _p(a), _p(b), _p(out)))
        return out.raw[:32 * n]

    def zp_from_hash(self, digests: bytes) -> bytes:
        n = len(digests) // 64
        out = ctypes.create_string_buffer(max(32 * n, 1))
        self._ck(self.lib.c12381_zp_from_hash_batch(self.h, n, _p(digests), _p(out)))
        return out.raw[:32 * n]

    def zp_inner_product(self, a: bytes, b: bytes | None = None) -> bytes:
        n = len(a) // 32
        out = ctypes.create_string_buffer(32)
        self._ck(self.lib.c12381_zp_inner_product(self.h, n, _p(a), _p(b), _p(out)))
        return out.raw

    def miller(self, g1: bytes, g2: bytes) -> bytes:
        n = len(g1) // 96
        out = ctypes.create_string_buffer(max(576 * n, 1))
        self._ck(self.lib.c12381_miller_batch(self.h, n, _p(g1), _p(g2), _p(out)))
        return out.raw[:576 * n]

    def fexp(self, f: bytes) -> bytes:
        n = len(f) // 576
        out = ctypes.create_string_buffer(max(576 * n, 1))
        self._ck(self.lib.c12381_fexp_batch(self.h, n, _p(f), _p(out)))
        return out.raw[:576 * n]

    GT_OPS = {"mul": 0, "conj": 1, "pow": 2, "fexp": 3}

    def gt_op(self, op: str, a: bytes, b: bytes | None = None) -> bytes:
        n = len(a) // 576
        out = ctypes.create_string_buffer(max(576 * n, 1))
        self._ck(self.lib.c12381_gt_op_batch(self.h, self.GT_OPS[op], n, _p(a), _p(b), _p(out)))
        return out.raw[:576 * n]

    def gt_is_unity(self, a: bytes) -> bytes:
        n = len(a) // 576
        out = ctypes.create_string_buffer(max(n, 1))
        self._ck(self.lib.c12381_gt_is_unity_batch(self.h, n, _p(a), _p(out)))
        return out.raw[:n]

    def bbs_plus_verify(self, g1: bytes, g2: bytes, h0: bytes, h: bytes, w: bytes, A: bytes, x: bytes, r: bytes, m: bytes,
                        strict: bool = True) -> bytes:
        """m is message-major: block i of signature j at m[32*(i*n + j)]."""
        n = len(A) // 96
        nmsg = len(h) // 96
        out = ctypes.create_string_buffer(max(n, 1))
        self._ck(self.lib.c12381_bbs_plus_verify_batch(self.h, n, nmsg, _p(g1), _p(g2), _p(h0), _p(h) if nmsg else None, _p(w), _p(A),
                                                       _p(x), _p(r), _p(m) if nmsg else None, _p(out)), allow_point=not strict)
        return out.raw[:n]

    def bbs_plus_verify_wire(self, g1_g2_h0: bytes, h49: bytes, pk97: bytes, sigs145: bytes, msgs: bytes, msg_len: int, strict: bool = True) -> bytes:
        """serialized public parameters / key / signatures (49 + 48 + 48 B) and raw messages in, one byte per signature out"""
        n, nh = len(sigs145) // 145, len(h49) // 49
        out = ctypes.create_string_buffer(max(n, 1))
        self._ck(self.lib.c12381_bbs_plus_verify_wire_batch(self.h, n, nh, msg_len, _p(g1_g2_h0), _p(h49) if nh else None, _p(pk97), _p(sigs145),
                                                            _p(msgs) if msg_len else None, _p(out)), allow_point=not strict)
        return out.raw[:n]

    def bbs_plus_verify_aggregate(self, g1: bytes, g2: bytes, h0: bytes, h: bytes, w: bytes, A: bytes, x: bytes, r: bytes, m: bytes,
                                  rho: bytes) -> bool:
        """ONE verdict for the batch by a random linear combination (rho: n x 32 B caller-drawn scalars).  True: every
        signature verifies (up to 2^-k for k-bit rho); False settles nothing — run bbs_plus_verify."""
        n = len(A) // 96
        nmsg = len(h) // 96
        out = ctypes.c_int(0)
        self._ck(self.lib.c12381_bbs_plus_verify_aggregate(self.h, n, nmsg, _p(g1), _p(g2), _p(h0), _p(h) if nmsg else None, _p(w),
                                                           _p(A) if n else None, _p(x) if n else None, _p(r) if n else None,
                                                           _p(m) if nmsg and n else None, _p(rho) if n else None, ctypes.byref(out)))
        return out.value == 1

    def bbs_plus_verify_aggregate_dev(self, n, nmsg, g1, g2, h0, h, w, A, x, r, m, rho, all_ok):
        self._ck(self.lib.c12381_bbs_plus_verify_aggregate_dev(self.h, n, nmsg, _p(g1), _p(g2), _p(h0), _p(h), _p(w), _p(A), _p(x), _p(r),
                                                               _p(m), _p(rho), _p(all_ok)))

    def bbs_plus_verify_dev(self, n, nmsg, g1, g2, h0, h, w, A, x, r, m, ok):
        self._ck(self.lib.c12381_bbs_plus_verify_batch_dev(self.h, n, nmsg, _p(g1), _p(g2), _p(h0), _p(h), _p(w), _p(A), _p(x), _p(r),
                                                           _p(m), _p(ok)))

    # ---- device-pointer entry points (ints = device addresses, e.g. torch tensor.data_ptr())
    def bbs_plus_sign(self, g1, h0, h, gamma32, x, r, m) -> bytes:
        n = len(x) // 32
        nmsg = len(h) // 96
        out = ctypes.create_string_buffer(max(96 * n, 1))
        self._ck(self.lib.c12381_bbs_plus_sign_batch(self.h, n, nmsg, _p(g1), _p(h0), _p(h), _p(gamma32), _p(x), _p(r), _p(m), _p(out)))
        return out.raw[:96 * n]

    def miller_dev(self, n, g1_ptr, g2_ptr, out_ptr):
        self._ck(self.lib.c12381_miller_batch_dev(self.h, n, _p(g1_ptr), _p(g2_ptr), _p(out_ptr)))

    def gt_op_dev(self, op, n, a_ptr, b_ptr, out_ptr):
        self._ck(self.lib.c12381_gt_op_batch_dev(self.h, self.GT_OPS[op], n, _p(a_ptr), _p(b_ptr), _p(out_ptr)))

    def gt_is_unity_dev(self, n, a_ptr, out_ptr):
        self._ck(self.lib.c12381_gt_is_unity_batch_dev(self.h, n, _p(a_ptr), _p(out_ptr)))

    def g1_decompress_dev(self, n, in_ptr, out_ptr, status_ptr):
        self._ck(self.lib.c12381_g1_decompress_batch_dev(self.h, n, _p(in_ptr), _p(out_ptr), _p(status_ptr)))

    def g2_decompress_dev(self, n, in_ptr, out_ptr, status_ptr):
        self._ck(self.lib.c12381_g2_decompress_batch_dev(self.h, n, _p(in_ptr), _p(out_ptr), _p(status_ptr)))

    def bbs_plus_verify_wire_dev(self, n, nh, msg_len, pub_ptr, h_ptr, pk_ptr, sig_ptr, msg_ptr, ok_ptr):
        self._ck(self.lib.c12381_bbs_plus_verify_wire_batch_dev(self.h, n, nh, msg_len, _p(pub_ptr), _p(h_ptr), _p(pk_ptr), _p(sig_ptr), _p(msg_ptr), _p(ok_ptr)))

    def g1_from_hash_dev(self, n, digests_ptr, out_ptr, fmt=96):
        self._ck(self.lib.c12381_g1_from_hash_batch_dev(self.h, n, _p(digests_ptr), _p(out_ptr), fmt))

    def zp_op_dev(self, op, n, a_ptr, b_ptr, out_ptr):
        self._ck(self.lib.c12381_zp_op_batch_dev(self.h, self.ZP_OPS[op], n, _p(a_ptr), _p(b_ptr), _p(out_ptr)))

    def zp_inner_product_dev(self, n, a_ptr, b_ptr, out_ptr):
        self._ck(self.lib.c12381_zp_inner_product_dev(self.h, n, _p(a_ptr), _p(b_ptr), _p(out_ptr)))

    def g2_mul_dev(self, n, pts_ptr, sc_ptr, out_ptr, fmt=97):
        self._ck(self.lib.c12381_g2_mul_batch_dev(self.h, n, _p(pts_ptr), _p(sc_ptr), _p(out_ptr), fmt))

    def pair_dev(self, n, g1_ptr, g2_ptr, gt_ptr):
        self._ck(self.lib.c12381_pair_batch_dev(self.h, n, _p(g1_ptr), _p(g2_ptr), _p(gt_ptr)))

    def pair_eq_dev(self, n, a1, a2, b1, b2, ok_ptr):
        self._ck(self.lib.c12381_pair_eq_batch_dev(self.h, n, _p(a1), _p(a2), _p(b1), _p(b2), _p(ok_ptr)))

    def fp_mulchain_dev(self, n, iters, a_ptr, b_ptr, out_ptr):
        self._ck(self.lib.c12381_fp_mulchain_dev(self.h, n, iters, _p(a_ptr), _p(b_ptr), _p(out_ptr)))

    def fp_op_dev(self, op, n, a_ptr, b_ptr, out_ptr):
        self._ck(self.lib.c12381_fp_op_batch_dev(self.h, self.FP_OPS[op], n, _p(a_ptr), _p(b_ptr), _p(out_ptr)))

    def g1_mul_dev(self, n, pts_ptr, sc_ptr, out_ptr, fmt=49):
        self._ck(self.lib.c12381_g1_mul_batch_dev(self.h, n, _p(pts_ptr), _p(sc_ptr), _p(out_ptr), fmt))

    def g1_msm_dev(self, n, pts_ptr, sc_ptr, out_ptr, fmt=49):
        self._ck(self.lib.c12381_g1_msm_dev(self.h, n, _p(pts_ptr), _p(sc_ptr), _p(out_ptr), fmt))


def g1_msm_multi(contexts, pts: bytes, scalars: bytes, fmt: int = 49) -> bytes:
    """c12381_g1_msm_multi: one host process, one Context per GPU."""
    lib = contexts[0].lib
    n = len(pts) // 96
    arr = (ctypes.c_void_p * len(contexts))(*[c.h for c in contexts])
    out = ctypes.create_string_buffer(fmt)
    rc = lib.c12381_g1_msm_multi(arr, len(contexts), n, _p(pts), _p(scalars), _p(out), fmt)
    if rc != 0:
        raise C12381Error(rc, "c12381_g1_msm_multi")
    return out.raw[:fmt]
