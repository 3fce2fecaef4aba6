#!/usr/bin/env python3
"""PCIe-inclusive rates of the host-pointer entry points (DESIGN.md note; never the bench value)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools.libsel  # noqa: E402,F401  (C12381_LIB -> capi.use_library)
from crypto12381_amd import Context  # noqa: E402
from tools.prof_driver import G1, G2, sc  # noqa: E402

c = Context(0)
n = 1 << 20
base = c.g1_mul(G1 * 4096, sc(1, 4096), 96) * (n // 4096)
k = sc(2, n)
import ctypes  # noqa: E402
c.g1_mul(base[:96 * 1024], k[:32 * 1024], 96)
outb = ctypes.create_string_buffer(96 * n)          # the C call alone: no Python-side copies of the 100 MB result
for _ in range(2):
    t0 = time.perf_counter()
    rc = c.lib.c12381_g1_mul_batch(c.h, n, base, k, outb, 96)
    dt = time.perf_counter() - t0
assert rc == 0
print("g1_mul_batch host pointers, 2^20: %.1f ms  %.3e /s  (%.0f MB over PCIe)" % (dt * 1e3, n / dt, n * (96 + 32 + 96) / 1e6))
m = 1 << 16
p = base[:96 * m]
q = c.g2_mul(G2 * 1024, sc(4, 1024), 192) * (m // 1024)
c.pair(p[:96 * 64], q[:192 * 64])
gtb = ctypes.create_string_buffer(576 * m)
for _ in range(2):
    t0 = time.perf_counter()
    rc = c.lib.c12381_pair_batch(c.h, m, p, q, gtb)
    dt = time.perf_counter() - t0
assert rc == 0
print("pair_batch host pointers, 2^16: %.1f ms  %.3e /s  (%.0f MB over PCIe)" % (dt * 1e3, m / dt, m * (96 + 192 + 576) / 1e6))
c.close()
