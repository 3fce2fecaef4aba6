#!/usr/bin/env python3
"""Fixed workload for the round-3 rocprofv3 passes: ONE launch sequence of every dominant kernel at its BASELINE size through the
device-pointer entry points — G1 2^18 (one g1_mul_kernel launch of two machine rounds), G2 2^17 (one g2_mul2_kernel launch), 2^16 pairings
(pair3_queue_kernel), Miller loops and final exponentiations alone, MSM 2^22 (msm_bucket_kernel), 2^18 BBS+ verifications
(pair3_prod_fixed_queue_kernel).  Usage: python3 tools/prof_driver3.py [all|g1|g2|pair|msm|bbs]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools.libsel  # noqa: E402,F401  (C12381_LIB -> capi.use_library)
from crypto12381_amd import Context  # noqa: E402
from tools.prof_driver import G1, G2, sc  # noqa: E402


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    c = Context(0)
    dev = torch.device("cuda", 0)
    s = torch.cuda.Stream(device=dev)
    c.set_stream(s.cuda_stream)

    def d(b):
        return torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    p1k = c.g1_mul(G1 * 1024, sc(3, 1024), 96)
    q1k = c.g2_mul(G2 * 1024, sc(4, 1024), 192)
    if what in ("all", "g1"):
        n = 1 << 18
        dp, dk, o = d(p1k * (n // 1024)), d(sc(2, n)), torch.empty(96 * n, dtype=torch.uint8, device=dev)
        c.g1_mul_dev(n, dp.data_ptr(), dk.data_ptr(), o.data_ptr(), 96); c.sync()
    if what in ("all", "g2"):
        n = 1 << 17
        dq, dk, o = d(q1k * (n // 1024)), d(sc(5, n)), torch.empty(192 * n, dtype=torch.uint8, device=dev)
        c.g2_mul_dev(n, dq.data_ptr(), dk.data_ptr(), o.data_ptr(), 192); c.sync()
    if what in ("all", "pair"):
        n = 1 << 16
        dp, dq = d(p1k * (n // 1024)), d(q1k * (n // 1024))
        gt, mil = torch.empty(576 * n, dtype=torch.uint8, device=dev), torch.empty(576 * n, dtype=torch.uint8, device=dev)
        c.pair_dev(n, dp.data_ptr(), dq.data_ptr(), gt.data_ptr()); c.sync()
        c.miller_dev(n, dp.data_ptr(), dq.data_ptr(), mil.data_ptr()); c.sync()
        c.gt_op_dev("fexp", n, mil.data_ptr(), None, gt.data_ptr()); c.sync()
    if what in ("all", "msm"):
        n = 1 << 22
        dp, dk, o = d(p1k * (n // 1024)), d(sc(6, n)), torch.empty(96, dtype=torch.uint8, device=dev)
        c.g1_msm_dev(n, dp.data_ptr(), dk.data_ptr(), o.data_ptr(), 96); c.sync()
    if what in ("all", "bbs"):
        nb = 1 << 18

        def red(seed, k):
            a = np.frombuffer(sc(seed, k), dtype=np.uint8).reshape(k, 32).copy()
            a[:, 0] &= 0x3f
            return a
        pub = c.g1_mul_fixed(G1, red(51, 3).tobytes(), 96)
        g1p, h0, h = pub[:96], pub[96:192], pub[192:288]
        g2p = c.g2_mul_fixed(G2, red(52, 1).tobytes(), 192)
        gamma = red(53, 1).tobytes()
        w = c.g2_mul_fixed(g2p, gamma, 192)
        xs, rs, mm = red(54, nb), red(55, nb), red(56, nb)
        A = c.bbs_plus_sign(g1p, h0, h, gamma, xs.tobytes(), rs.tobytes(), mm.tobytes())
        dA, dx, dr, dm = d(A), d(xs.tobytes()), d(rs.tobytes()), d(mm.tobytes())
        dpub = [d(b) for b in (g1p, g2p, h0, h, w)]
        ok = torch.empty(nb, dtype=torch.uint8, device=dev)
        for _ in range(2):                      # the first call builds the tables
            c.bbs_plus_verify_dev(nb, 1, dpub[0].data_ptr(), dpub[1].data_ptr(), dpub[2].data_ptr(), dpub[3].data_ptr(), dpub[4].data_ptr(),
                                  dA.data_ptr(), dx.data_ptr(), dr.data_ptr(), dm.data_ptr(), ok.data_ptr())
            c.sync()
        assert int(ok.sum().item()) == nb
    c.close()


if __name__ == "__main__":
    main()
