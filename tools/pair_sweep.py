#!/usr/bin/env python3
"""Pairing kernel time against the number of resident wavefronts (21 pairings per wavefront, 2048 wavefront
slots at 2 waves/SIMD): tells a latency-bound kernel (t(1024 waves) ~ t(2048 waves)) from an issue-bound one."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools.libsel  # noqa: E402,F401  (C12381_LIB -> capi.use_library)
from crypto12381_amd import Context  # noqa: E402
from tools.prof_driver import G1, G2, sc  # noqa: E402


def main():
    c = Context(0)
    dev = torch.device("cuda", 0)
    s = torch.cuda.Stream(device=dev)
    c.set_stream(s.cuda_stream)
    base = 1024
    p = c.g1_mul(G1 * base, sc(3, base), 96)
    q = c.g2_mul(G2 * base, sc(4, base), 192)
    waves = [int(a) for a in sys.argv[1:]] or [256, 512, 1024, 1536, 2048, 3072, 3121, 4096, 6144]
    for w in waves:
        n = w * 21
        rep = (n + base - 1) // base
        dp = torch.frombuffer(bytearray((p * rep)[:96 * n]), dtype=torch.uint8).to(dev)
        dq = torch.frombuffer(bytearray((q * rep)[:192 * n]), dtype=torch.uint8).to(dev)
        out = torch.empty(576 * n, dtype=torch.uint8, device=dev)
        c.pair_dev(n, dp.data_ptr(), dq.data_ptr(), out.data_ptr()); c.sync()
        t0 = time.perf_counter()
        for _ in range(3):
            c.pair_dev(n, dp.data_ptr(), dq.data_ptr(), out.data_ptr())
        c.sync()
        dt = (time.perf_counter() - t0) / 3
        print("waves=%5d n=%7d  %.2f ms  %.3e pairings/s  ms per 2048-wave round=%.2f" % (w, n, dt * 1e3, n / dt, dt * 1e3 / (w / 2048)), flush=True)
    c.close()


if __name__ == "__main__":
    main()
