#!/usr/bin/env python3
"""What "drop-in" costs per call: latency of the reference's scalar seam functions when they are served by the HIP library through
crypto12381_amd/csrc/miracl_core_interface_hip.cpp (oracle/_ref/libc12381_shimtest.so), next to the same calls on the reference's CPU
path (oracle/_ref/libc12381_ref.so).  Every shim call is a size-1 batch: two pageable host<->device copies, one to three kernel
launches at single-wavefront latency and a stream synchronisation.  Usage (GPU box): python tools/shim_latency.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from oracle.bindings import Oracle  # noqa: E402
from util import scalars  # noqa: E402


def main():
    shim, ref = Oracle("shim"), Oracle("reference")
    g1, g2 = ref.g1_generator(), ref.g2_generator()
    k = scalars(1, 2)
    P = ref.g1_mul(g1, k[:32], 96)
    Q = ref.g2_mul(g2, k[32:], 192)
    gt = ref.pair(P, Q, 1)
    c49 = ref.g1_mul(g1, k[:32], 49)
    calls = [
        ("multiply(point1&, big)  [PAIR_G1mul]", lambda o: o.g1_mul(P, k[:32], 96)),
        ("add(point1&, point1&)  [ECP_add]", lambda o: o.g1_add(P, g1, 96)),
        ("from_bytes(point1&) compressed  [ECP_fromOctet]", lambda o: o.g1_decompress(c49)),
        ("multiply(point2&, big)  [PAIR_G2mul]", lambda o: o.g2_mul(Q, k[:32], 192)),
        ("pair_ate + pair_final_exponentiation", lambda o: o.pair(P, Q, 1)),
        ("pair_ate alone", lambda o: o.miller(P, Q)),
        ("pow(fp12&, fp12&, big)  [FP12_pow]", lambda o: o.gt_op("pow", gt, k[:32])),
        ("multiply(fp12&, fp12&)  [FP12_mul]", lambda o: o.gt_op("mul", gt, gt)),
    ]
    print("%-52s %12s %12s" % ("seam function (one call)", "shim -> GPU", "reference CPU"))
    for name, fn in calls:
        res = []
        for o in (shim, ref):
            fn(o)
            reps = 50
            t0 = time.perf_counter()
            for _ in range(reps):
                fn(o)
            res.append((time.perf_counter() - t0) / reps * 1e6)
        print("%-52s %9.0f us %9.0f us" % (name, res[0], res[1]), flush=True)


if __name__ == "__main__":
    main()
