"""TEST INFRASTRUCTURE — never the product path (only tests/ and bench.py's cpu_baseline leg import this).

The CPU baseline over worker PROCESSES instead of threads.

Why: MIRACL's constant-time moves keep a function-level `static chunk R` that every call advances
(/root/reference/3rd-party/miracl-core/big_B384_58.cpp:99-124 BIG_cmove, :126-148 BIG_cswap, :150-170 BIG_dcmove).
Threads of one address space therefore write one cache line from every core on every table look-up:
ECP_select / ECP2_select / FP12_select and the field inversions sit on that line, the Miller loop does not.
Measured in the 8-vCPU build container (4096 units): 8 threads deliver 2.7x one thread on PAIR_G1mul and
2.6x on PAIR_fexp but 6.3x on PAIR_ate; 8 processes deliver 4.9x / 5.4x / 5.4x.  A deployment that wants the
reference's best rate per host runs it one process per core, so that is what the baseline times.

The pool is forked BEFORE the caller initialises the GPU (bench.py creates it ahead of torch.cuda.is_available()),
so no worker ever holds a HIP context; workers only call into the oracle library on one thread each.
Methods mirror oracle.bindings.Oracle for the batched calls bench.py times; each splits the units into
contiguous chunks, one per worker, and joins the rows (the MSM adds the per-worker partial sums, as the
threaded wrapper ref_g1_msm does with its per-thread partial sums — oracle/ref_wrap.cpp:239-261).
"""
from __future__ import annotations

import multiprocessing as mp

_ORACLES: dict = {}


def _call(kind: str, method: str, args: tuple):
    o = _ORACLES.get(kind)
    if o is None:
        from oracle.bindings import Oracle
        o = _ORACLES[kind] = Oracle(kind)
    return getattr(o, method)(*args)


def _ready(_):
    return True


class OraclePool:
    def __init__(self, kind: str, procs: int):
        self.kind, self.procs = kind, max(1, int(procs))
        self._pool = mp.get_context("fork").Pool(self.procs)
        self._pool.map(_ready, range(self.procs))                  # workers exist before the first timed call

    def close(self):
        if self._pool is not None:
            self._pool.terminate()
            self._pool.join()
            self._pool = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def bounds(self, n: int):
        p = min(self.procs, max(n, 1))
        return [(n * t // p, n * (t + 1) // p) for t in range(p)]

    def _rows(self, method: str, n: int, build) -> bytes:
        """build(lo, hi) -> the argument tuple of Oracle.<method> for units [lo, hi) on ONE thread"""
        return b"".join(self._pool.starmap(_call, [(self.kind, method, build(lo, hi)) for lo, hi in self.bounds(n)]))

    # ---- the calls bench.py times (argument meaning as in oracle.bindings.Oracle; the thread count is the pool's size)
    def g1_mul(self, pts: bytes, scalars: bytes, fmt: int = 49) -> bytes:
        return self._rows("g1_mul", len(pts) // 96, lambda lo, hi: (pts[96 * lo:96 * hi], scalars[32 * lo:32 * hi], fmt, 1))

    def g2_mul(self, pts: bytes, scalars: bytes, fmt: int = 97) -> bytes:
        return self._rows("g2_mul", len(pts) // 192, lambda lo, hi: (pts[192 * lo:192 * hi], scalars[32 * lo:32 * hi], fmt, 1))

    def pair(self, g1: bytes, g2: bytes) -> bytes:
        return self._rows("pair", len(g1) // 96, lambda lo, hi: (g1[96 * lo:96 * hi], g2[192 * lo:192 * hi], 1))

    def miller_t(self, g1: bytes, g2: bytes) -> bytes:
        return self._rows("miller_t", len(g1) // 96, lambda lo, hi: (g1[96 * lo:96 * hi], g2[192 * lo:192 * hi], 1))

    def fexp_t(self, f: bytes) -> bytes:
        return self._rows("fexp_t", len(f) // 576, lambda lo, hi: (f[576 * lo:576 * hi], 1))

    def g1_msm(self, pts: bytes, scalars: bytes, fmt: int = 49) -> bytes:
        n = len(pts) // 96
        parts = self._pool.starmap(_call, [(self.kind, "g1_msm", (pts[96 * lo:96 * hi], scalars[32 * lo:32 * hi], 96, 1))
                                           for lo, hi in self.bounds(n)])
        acc = parts[0]
        for p in parts[1:]:
            acc = _call(self.kind, "g1_add", (acc, p, 96))
        return acc if fmt == 96 else _call(self.kind, "g1_compress", (acc,))

    def bbs_plus_verify(self, g1: bytes, g2: bytes, h0: bytes, h: bytes, w: bytes, A: bytes, x: bytes, r: bytes, m: bytes) -> bytes:
        n, nmsg = len(A) // 96, len(h) // 96

        def build(lo, hi):                                         # m is message-major: block i of signature j at (i * n + j) * 32
            mm = b"".join(m[(i * n + lo) * 32:(i * n + hi) * 32] for i in range(nmsg))
            return (g1, g2, h0, h, w, A[96 * lo:96 * hi], x[32 * lo:32 * hi], r[32 * lo:32 * hi], mm, 1)
        return self._rows("bbs_plus_verify", n, build)

    def bbs_plus_verify_wire(self, g1_g2_h0: bytes, h49: bytes, pk97: bytes, sigs145: bytes, msgs: bytes, msg_len: int) -> bytes:
        n = len(sigs145) // 145
        return self._rows("bbs_plus_verify_wire", n,                # msg_len raw bytes per signature, signature-major
                          lambda lo, hi: (g1_g2_h0, h49, pk97, sigs145[145 * lo:145 * hi], msgs[msg_len * lo:msg_len * hi], msg_len, 1))


class OracleThreads:
    """the same call shapes over the wrapper's own threads (parity-only runs, where no pool was forked)"""

    def __init__(self, oracle, threads: int):
        self.o, self.t = oracle, max(1, int(threads))

    def g1_mul(self, pts, scalars, fmt=49): return self.o.g1_mul(pts, scalars, fmt, self.t)
    def g2_mul(self, pts, scalars, fmt=97): return self.o.g2_mul(pts, scalars, fmt, self.t)
    def pair(self, g1, g2): return self.o.pair(g1, g2, self.t)
    def miller_t(self, g1, g2): return self.o.miller_t(g1, g2, self.t)
    def fexp_t(self, f): return self.o.fexp_t(f, self.t)
    def g1_msm(self, pts, scalars, fmt=49): return self.o.g1_msm(pts, scalars, fmt, self.t)
    def bbs_plus_verify(self, *a): return self.o.bbs_plus_verify(*a, self.t)
    def bbs_plus_verify_wire(self, *a): return self.o.bbs_plus_verify_wire(*a, self.t)
