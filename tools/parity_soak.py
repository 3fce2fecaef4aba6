#!/usr/bin/env python3
"""Parity soak: fresh random inputs every iteration, EVERY lane of every batch against the compiled reference (oracle/_ref, all host
threads), for a time budget.  Not a rerun of one input set: each iteration draws new points and scalars from its own seed, so the run
widens the set of inputs the GPU path has been compared on (G1 / G2 scalar multiplications, pairings incl. the split forms, GT powers on both routes, bucket MSM).

    python tools/parity_soak.py [--minutes 6] [--log2-g1 17] [--log2-g2 15] [--log2-pair 14] [--log2-msm 15]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools.libsel  # noqa: E402,F401  (C12381_LIB -> capi.use_library)
from crypto12381_amd import Context  # noqa: E402
from oracle.bindings import Oracle  # noqa: E402  (the checker; tools/ and tests/ only)
from tools.prof_driver import G1, G2  # noqa: E402

R_ORDER = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001


def scalars(rng, n, edges=True):
    s = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    if edges:
        for j, k in enumerate([0, 1, 2, R_ORDER - 1, R_ORDER, R_ORDER + 1, (1 << 256) - 1, (1 << 128) - 1, 1 << 128, 0xd201000000010000 ** 2]):
            if j < n:
                s[j] = np.frombuffer(int(k).to_bytes(32, "big"), dtype=np.uint8)
    return s.tobytes()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--minutes", type=float, default=6.0)
    ap.add_argument("--log2-g1", type=int, default=17)
    ap.add_argument("--log2-g2", type=int, default=15)
    ap.add_argument("--log2-pair", type=int, default=14)
    ap.add_argument("--log2-msm", type=int, default=15)
    ap.add_argument("--seed", type=int, default=20261004)
    a = ap.parse_args()
    cores = os.cpu_count() or 1
    c = Context(0)
    ref = Oracle("reference")
    n1, n2, n3, n4 = 1 << a.log2_g1, 1 << a.log2_g2, 1 << a.log2_pair, 1 << a.log2_msm
    t0 = time.time()
    it = 0
    lanes = {"g1_mul": 0, "g2_mul": 0, "pair": 0, "miller": 0, "fexp": 0, "gt_pow": 0, "msm_terms": 0}
    while time.time() - t0 < a.minutes * 60:
        rng = np.random.Generator(np.random.PCG64(a.seed + it))
        # inputs: random multiples of the generators made on the GPU; the reference decodes them itself (a point off the curve fails there)
        p1 = c.g1_mul(G1 * n1, scalars(rng, n1, edges=False), 96)
        p2 = c.g2_mul(G2 * n2, scalars(rng, n2, edges=False), 192)
        k1, k2 = scalars(rng, n1), scalars(rng, n2)
        for fmt in (96, 49):
            assert c.g1_mul(p1, k1, fmt) == ref.g1_mul(p1, k1, fmt, cores), ("g1_mul", it, fmt)
        for fmt in (192, 97):
            assert c.g2_mul(p2, k2, fmt) == ref.g2_mul(p2, k2, fmt, cores), ("g2_mul", it, fmt)
        lanes["g1_mul"] += 2 * n1
        lanes["g2_mul"] += 2 * n2
        q1, q2 = p1[:96 * n3], (p2 * ((n3 + n2 - 1) // n2))[:192 * n3]
        gt = c.pair(q1, q2)
        assert gt == ref.pair(q1, q2, cores), ("pair", it)
        mil = c.miller(q1, q2)
        assert mil == ref.miller_t(q1, q2, cores), ("miller", it)
        assert c.gt_op("fexp", mil) == gt, ("fexp", it)
        lanes["pair"] += n3
        lanes["miller"] += n3
        lanes["fexp"] += n3
        # GT powers: pairing values (windowed ladder) with one Miller value per 64 elements spliced in (its wavefront takes the reference's digit
        # sequence); exponents = the G1 scalars, edge values in front
        n5 = min(n3, 2048)
        base = bytearray(gt[:576 * n5])
        for j in range(31, n5, 64):
            base[576 * j:576 * j + 576] = mil[576 * j:576 * j + 576]
        assert c.gt_op("pow", bytes(base), k1[:32 * n5]) == ref.gt_op("pow", bytes(base), k1[:32 * n5]), ("gt_pow", it)
        lanes["gt_pow"] += n5
        m1, mk = p1[:96 * n4], k1[:32 * n4]
        assert c.g1_msm(m1, mk, 96) == ref.g1_msm(m1, mk, 96, cores), ("msm", it)
        lanes["msm_terms"] += n4
        it += 1
        print("iteration %3d ok  %.0f s  %s" % (it, time.time() - t0, " ".join("%s=%d" % kv for kv in lanes.items())), flush=True)
    print("parity soak ok: %d iterations, every lane equal to the compiled reference: %s" % (it, ", ".join("%s %d" % kv for kv in lanes.items())))


if __name__ == "__main__":
    main()
