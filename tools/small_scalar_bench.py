#!/usr/bin/env python3
"""Scalar multiplication with SHORT scalars (k mod r < x^2, i.e. any scalar of at most 127 bits): the reference's
multiply() then owes the [r]phi(P) / [r]psi^i(Q) terms, which the device evaluates on a side
path.  Times 2^20 G1 and 2^17 G2 multiplications with uniform 255-bit, 128-bit and 64-bit scalars.
usage: python tools/small_scalar_bench.py [flags]   (flags: integer passed to the *_flags entry points when present)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools.libsel  # noqa: E402,F401  (C12381_LIB -> capi.use_library)
from crypto12381_amd import Context  # noqa: E402
from tools.prof_driver import G1, G2, sc  # noqa: E402

R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001


def scalars(seed, n, bits):
    raw = sc(seed, n)
    if bits >= 255:
        return b"".join((int.from_bytes(raw[32 * i:32 * i + 32], "big") % R).to_bytes(32, "big") for i in range(n))
    keep = bits // 8
    out = bytearray(32 * n)
    for i in range(n):
        out[32 * i + 32 - keep:32 * i + 32] = raw[32 * i:32 * i + keep]
    return bytes(out)


def main():
    flags = int(sys.argv[1]) if len(sys.argv) > 1 else None
    c = Context(0)
    dev = torch.device("cuda", 0)
    s = torch.cuda.Stream(device=dev)
    c.set_stream(s.cuda_stream)
    n1, n2 = 1 << 20, 1 << 17
    p = c.g1_mul(G1 * 4096, sc(1, 4096), 96) * (n1 // 4096)
    q = c.g2_mul(G2 * 1024, sc(4, 1024), 192) * (n2 // 1024)
    dp = torch.frombuffer(bytearray(p), dtype=torch.uint8).to(dev)
    dq = torch.frombuffer(bytearray(q), dtype=torch.uint8).to(dev)
    o1 = torch.empty(96 * n1, dtype=torch.uint8, device=dev)
    o2 = torch.empty(192 * n2, dtype=torch.uint8, device=dev)
    for bits in (255, 128, 64):
        k1 = torch.frombuffer(bytearray(scalars(7, n1, bits)), dtype=torch.uint8).to(dev)
        k2 = k1[:32 * n2].clone()
        for name, fn, n in (("g1", lambda: c.g1_mul_dev(n1, dp.data_ptr(), k1.data_ptr(), o1.data_ptr(), 96) if flags is None else
                             c.g1_mul_flags_dev(n1, dp.data_ptr(), k1.data_ptr(), o1.data_ptr(), 96, flags), n1),
                            ("g2", lambda: c.g2_mul_dev(n2, dq.data_ptr(), k2.data_ptr(), o2.data_ptr(), 192) if flags is None else
                             c.g2_mul_flags_dev(n2, dq.data_ptr(), k2.data_ptr(), o2.data_ptr(), 192, flags), n2)):
            fn(); c.sync()
            t0 = time.perf_counter()
            for _ in range(3):
                fn()
            c.sync()
            dt = (time.perf_counter() - t0) / 3
            print("%s %3d-bit scalars flags=%s: %8.2f ms per 2^%d  %.3e /s" % (name, bits, flags, dt * 1e3, n.bit_length() - 1, n / dt), flush=True)
    c.close()


if __name__ == "__main__":
    main()
