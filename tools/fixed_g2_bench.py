#!/usr/bin/env python3
"""2^16 pairings against ONE G2 argument (table-driven lines) vs the general entry point."""
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
n = 1 << 16
p = c.g1_mul(G1 * 1024, sc(3, 1024), 96) * (n // 1024)
q = c.g2_mul(G2, sc(4, 1), 192)
dp = torch.frombuffer(bytearray(p), dtype=torch.uint8).to(dev)
dq1 = torch.frombuffer(bytearray(q), dtype=torch.uint8).to(dev)
dqn = torch.frombuffer(bytearray(q * n), dtype=torch.uint8).to(dev)
out = torch.empty(576 * n, dtype=torch.uint8, device=dev)
for name, fn in (("general", lambda: c.pair_dev(n, dp.data_ptr(), dqn.data_ptr(), out.data_ptr())),
                 ("fixed G2", lambda: c.pair_fixed_g2_dev(n, dp.data_ptr(), dq1.data_ptr(), out.data_ptr()))):
    fn(); c.sync()
    t0 = time.perf_counter()
    for _ in range(3):
        fn()
    c.sync()
    dt = (time.perf_counter() - t0) / 3
    print("%-9s 2^16 pairings: %.2f ms  %.3e /s" % (name, dt * 1e3, n / dt))
