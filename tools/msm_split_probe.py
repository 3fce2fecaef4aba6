#!/usr/bin/env python3
"""Experiment: what would a pipelined bucket product gain?  One 2^22-term product (a) in one call, (b) as 2 / 4 / 8 sub-products issued
alternately on two contexts (the partial sums added at the end: one point addition each, not timed here), (c) whole products streamed over
two contexts.  The small kernels around the bucket kernel (sorts, window reductions, Horner) are latency chains: beside another product's
bucket kernel they cost nothing.

    python tools/msm_split_probe.py [--log2-terms 22] [--steps 10]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B
from crypto12381_amd import Context


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2-terms", type=int, default=22)
    ap.add_argument("--steps", type=int, default=10)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    ctxs = []
    for _ in range(2):
        c = Context(0)
        s = torch.cuda.Stream(device=dev)
        c.set_stream(s.cuda_stream)
        ctxs.append((c, s))
    n = 1 << args.log2_terms
    gen1 = B.dev_bytes(B.G1_GEN, dev)
    s1 = torch.from_numpy(B.reduced_scalars(21, n)).to(dev)
    k1 = torch.from_numpy(B.make_scalars(22, n)).to(dev)
    pts = torch.empty(n * 96, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize(dev)
    ctxs[0][0].g1_mul_fixed_dev(n, gen1.data_ptr(), s1.data_ptr(), pts.data_ptr(), 96)
    assert ctxs[0][0].sync() == 0
    outs = torch.zeros(16 * 96, dtype=torch.uint8, device=dev)

    def part(c, lo, hi, slot):
        c.g1_msm_dev(hi - lo, pts.data_ptr() + 96 * lo, k1.data_ptr() + 32 * lo, outs.data_ptr() + 96 * slot, 96)

    def timed(fn, steps):
        fn()
        torch.cuda.synchronize(dev)
        t = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t) / steps * 1e3

    def split(parts):
        def go():
            for p in range(parts):
                part(ctxs[p & 1][0], n * p // parts, n * (p + 1) // parts, p)
            for c, _ in ctxs:                                  # the join a library-internal pipeline would express with events
                c.sync()
        return go
    whole = timed(lambda: (part(ctxs[0][0], 0, n, 0), ctxs[0][0].sync()), args.steps)
    print("one product of 2^%d terms, one call:                     %7.3f ms" % (args.log2_terms, whole))
    ref = bytes(outs[:96].cpu().numpy())
    for parts in (2, 4, 8):
        t = timed(split(parts), args.steps)
        # the partial sums must add up to the whole product
        acc = bytes(outs[:96].cpu().numpy())
        for p in range(1, parts):
            acc = ctxs[0][0].g1_add(acc, bytes(outs[96 * p:96 * p + 96].cpu().numpy()), 96)
        print("the same as %d sub-products alternating over two contexts:  %7.3f ms  (%+.1f %%)  sum of the parts equal: %s"
              % (parts, t, (t / whole - 1) * 100, acc == ref), flush=True)
    k = [0]

    def streamed():
        part(ctxs[k[0] & 1][0], 0, n, k[0] & 1)
        k[0] += 1
    t = timed(streamed, 2 * args.steps)
    for c, _ in ctxs:
        c.sync()
    print("whole products streamed over two contexts:                %7.3f ms per product  (%+.1f %%)" % (t, (t / whole - 1) * 100))
    for c, _ in ctxs:
        c.close()


if __name__ == "__main__":
    main()
