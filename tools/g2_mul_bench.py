#!/usr/bin/env python3
"""Generic G1 / G2 scalar multiplication and the MSM at sizes that fill whole machine rounds at 2, 3 and 4 waves per SIMD
(A/B of the occupancy variants: C12381_LIB=crypto12381_amd/lib/exp/lib<name>.so python tools/g2_mul_bench.py)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools.libsel  # noqa: E402,F401  (C12381_LIB -> capi.use_library)
from crypto12381_amd import Context  # noqa: E402
from tools.prof_driver import G1, G2, sc  # noqa: E402

c = Context(0)
dev = torch.device("cuda", 0)
s = torch.cuda.Stream(device=dev)
c.set_stream(s.cuda_stream)


def dev_b(b):
    return torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)


def timeit(fn, reps=3):
    fn(); c.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    c.sync()
    return (time.perf_counter() - t0) / reps


n2 = 3 << 17                                          # 6 x 2^16: whole rounds at 2 and 3 waves per SIMD
q = c.g2_mul(G2 * 1024, sc(4, 1024), 192) * (n2 // 1024)
dq, dk = dev_b(q), dev_b(sc(5, n2))
out = torch.empty(192 * n2, dtype=torch.uint8, device=dev)
dt = timeit(lambda: c.g2_mul_dev(n2, dq.data_ptr(), dk.data_ptr(), out.data_ptr(), 192))
print("g2_mul %d: %.2f ms  %.3e /s  (%.2f ms per 2^17)" % (n2, dt * 1e3, n2 / dt, dt * 1e3 * (1 << 17) / n2))
n1 = 3 << 19
p = c.g1_mul(G1 * 1024, sc(6, 1024), 96) * (n1 // 1024)
dp, dk1 = dev_b(p), dev_b(sc(7, n1))
out1 = torch.empty(96 * n1, dtype=torch.uint8, device=dev)
dt = timeit(lambda: c.g1_mul_dev(n1, dp.data_ptr(), dk1.data_ptr(), out1.data_ptr(), 96))
print("g1_mul %d: %.2f ms  %.3e /s  (%.2f ms per 2^20)" % (n1, dt * 1e3, n1 / dt, dt * 1e3 * (1 << 20) / n1))
nm = 1 << 22
pm = (p * ((nm * 96 + len(p) - 1) // len(p)))[: nm * 96]
dpm, dkm = dev_b(pm), dev_b(sc(8, nm))
om = torch.empty(96, dtype=torch.uint8, device=dev)
dt = timeit(lambda: c.g1_msm_dev(nm, dpm.data_ptr(), dkm.data_ptr(), om.data_ptr(), 96))
print("msm 2^22: %.2f ms" % (dt * 1e3))
npair = 1 << 16
pp = (p * ((npair * 96 + len(p) - 1) // len(p)))[: npair * 96]
qq = (q * ((npair * 192 + len(q) - 1) // len(q)))[: npair * 192]
dpp, dqq = dev_b(pp), dev_b(qq)
gt = torch.empty(576 * npair, dtype=torch.uint8, device=dev)
dt = timeit(lambda: c.pair_dev(npair, dpp.data_ptr(), dqq.data_ptr(), gt.data_ptr()))
print("pair 2^16: %.2f ms  %.3e /s" % (dt * 1e3, npair / dt))
