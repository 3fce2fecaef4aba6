#!/usr/bin/env python3
"""Generic G2 scalar multiplication, 2^17 distinct points (one launch)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crypto12381_amd import Context  # noqa: E402
from tools.prof_driver import G2, sc  # noqa: E402

c = Context(0)
dev = torch.device("cuda", 0)
s = torch.cuda.Stream(device=dev)
c.set_stream(s.cuda_stream)
n = 1 << 17
q = c.g2_mul(G2 * 1024, sc(4, 1024), 192) * (n // 1024)
dq = torch.frombuffer(bytearray(q), dtype=torch.uint8).to(dev)
dk = torch.frombuffer(bytearray(sc(5, n)), dtype=torch.uint8).to(dev)
out = torch.empty(192 * n, dtype=torch.uint8, device=dev)
c.g2_mul_dev(n, dq.data_ptr(), dk.data_ptr(), out.data_ptr(), 192); c.sync()
t0 = time.perf_counter()
for _ in range(3):
    c.g2_mul_dev(n, dq.data_ptr(), dk.data_ptr(), out.data_ptr(), 192)
c.sync()
dt = (time.perf_counter() - t0) / 3
print("g2_mul 2^17: %.2f ms  %.3e /s" % (dt * 1e3, n / dt))
