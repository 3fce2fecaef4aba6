#!/usr/bin/env python3
"""Experiment: do consecutive batches overlap when they come from TWO contexts on two HIP streams?

The work-queue kernels end with a tail in which the older wavefront of every SIMD has left (3-5 % of a launch);
a caller that streams batches can fill that tail with the head of the next batch by alternating two contexts (each with its
own stream and workspaces).  This script times K batches of 2^16 pairings (or Miller loops / final exponentiations / 2^20 G1
multiplications) issued (a) from one context, back to back, and (b) alternately from two contexts into two output buffers, and
checks that both outputs of (b) equal the output of (a).

    python tools/pipelined_batches.py [--steps 20] [--log2-pairings 16] [--log2-g1 20]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B                                              # generators and input helpers only
from crypto12381_amd import Context


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--log2-pairings", type=int, default=16)
    ap.add_argument("--log2-g1", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=3)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    ctxs, streams = [], []
    for _ in range(2):
        c = Context(0)
        s = torch.cuda.Stream(device=dev)
        c.set_stream(s.cuda_stream)
        ctxs.append(c); streams.append(s)
    n1, npair = 1 << args.log2_g1, 1 << args.log2_pairings
    gen1, gen2 = B.dev_bytes(B.G1_GEN, dev), B.dev_bytes(B.G2_GEN, dev)
    s1 = torch.from_numpy(B.reduced_scalars(11, n1)).to(dev)
    s2 = torch.from_numpy(B.reduced_scalars(12, npair)).to(dev)
    k1 = torch.from_numpy(B.make_scalars(13, n1)).to(dev)
    pts = torch.empty(n1 * 96, dtype=torch.uint8, device=dev)
    q2 = torch.empty(npair * 192, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize(dev)
    c0 = ctxs[0]
    c0.g1_mul_fixed_dev(n1, gen1.data_ptr(), s1.data_ptr(), pts.data_ptr(), 96)
    c0.g2_mul_fixed_dev(npair, gen2.data_ptr(), s2.data_ptr(), q2.data_ptr(), 192)
    assert c0.sync() == 0
    p1 = pts[: npair * 96]
    mil = torch.empty(npair * 576, dtype=torch.uint8, device=dev)
    c0.miller_dev(npair, p1.data_ptr(), q2.data_ptr(), mil.data_ptr())
    assert c0.sync() == 0

    def outs(nbytes):
        return [torch.empty(nbytes, dtype=torch.uint8, device=dev) for _ in range(3)]

    legs = {
        "pairing": (outs(npair * 576), lambda c, o: c.pair_dev(npair, p1.data_ptr(), q2.data_ptr(), o.data_ptr()), npair),
        "miller": (outs(npair * 576), lambda c, o: c.miller_dev(npair, p1.data_ptr(), q2.data_ptr(), o.data_ptr()), npair),
        "fexp": (outs(npair * 576), lambda c, o: c.gt_op_dev("fexp", npair, mil.data_ptr(), None, o.data_ptr()), npair),
        "g1_mul": (outs(n1 * 96), lambda c, o: c.g1_mul_dev(n1, pts.data_ptr(), k1.data_ptr(), o.data_ptr(), 96), n1),
    }

    def timed(fn):
        torch.cuda.synchronize(dev)
        t = time.perf_counter()
        fn()
        torch.cuda.synchronize(dev)
        return time.perf_counter() - t

    print("batches issued back to back from ONE context against alternating TWO contexts (two streams), ms per batch, %d batches per run" % args.steps)
    for name, (o, call, units) in legs.items():
        for c in ctxs:                                         # warm both contexts (workspaces, tables)
            call(c, o[0])
        torch.cuda.synchronize(dev)
        one, two = [], []
        for _ in range(args.rounds):
            one.append(timed(lambda: [call(ctxs[0], o[0]) for _ in range(args.steps)]) / args.steps * 1e3)
            two.append(timed(lambda: [call(ctxs[i & 1], o[1 + (i & 1)]) for i in range(args.steps)]) / args.steps * 1e3)
        assert all(c.sync() == 0 for c in ctxs)
        same = bool(torch.equal(o[0], o[1]) and torch.equal(o[0], o[2]))
        a, b = float(np.median(one)), float(np.median(two))
        print("%-8s one context %8.3f   two contexts %8.3f   (%+.1f %%)   outputs equal: %s   runs one %s two %s"
              % (name, a, b, (b / a - 1) * 100, same, ["%.3f" % x for x in one], ["%.3f" % x for x in two]), flush=True)
        if not same:
            raise SystemExit("pipelined_batches: outputs differ")
    for c in ctxs:
        c.close()


if __name__ == "__main__":
    main()
