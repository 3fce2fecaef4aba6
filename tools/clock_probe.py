#!/usr/bin/env python3
"""The clock the chip holds inside each dominant kernel: a one-lane probe kernel of the experiments library samples
(s_memtime, s_memrealtime) on a side stream while the kernel under study runs back to back on the main stream for ~0.4 s; the clock
over an interval is d(memtime) / d(memrealtime) x 100 MHz (MI355X_MICROARCH.md, DVFS give-back item 6).  Printed: the median over the
samples of the second half of the run (the first half lets the clock settle).

    C12381_LIB=crypto12381_amd/lib/libc12381_hip_exp.so python tools/clock_probe.py"""
import ctypes
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("C12381_LIB", os.path.join(ROOT, "crypto12381_amd", "lib", "libc12381_hip_exp.so"))
import tools.libsel  # noqa: E402,F401  (C12381_LIB -> capi.use_library)
from crypto12381_amd import Context  # noqa: E402
from crypto12381_amd.capi import _p  # noqa: E402
from tools.prof_driver import G1, G2, sc  # noqa: E402


def main():
    c = Context(0)
    dev = torch.device("cuda", 0)
    s = torch.cuda.Stream(device=dev)
    c.set_stream(s.cuda_stream)
    c.lib.c12381_exp_clock_probe.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]

    def d(b):
        return torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    p1k = c.g1_mul(G1 * 1024, sc(3, 1024), 96)
    q1k = c.g2_mul(G2 * 1024, sc(4, 1024), 192)
    n1, n2, npair, nm = 1 << 20, 1 << 18, 1 << 16, 1 << 22
    dp1, dk1, o1 = d(p1k * (n1 // 1024)), d(sc(2, n1)), torch.empty(96 * n1, dtype=torch.uint8, device=dev)
    dq2, dk2, o2 = d(q1k * (n2 // 1024)), d(sc(5, n2)), torch.empty(192 * n2, dtype=torch.uint8, device=dev)
    dpp, dqq = d(p1k * (npair // 1024)), d(q1k * (npair // 1024))
    gt, mil = torch.empty(576 * npair, dtype=torch.uint8, device=dev), torch.empty(576 * npair, dtype=torch.uint8, device=dev)
    dpm, dkm, om = d(p1k * (nm // 1024)), d(sc(6, nm)), torch.empty(96, dtype=torch.uint8, device=dev)
    work = {
        "g1_mul (2^20)": lambda: c.g1_mul_dev(n1, dp1.data_ptr(), dk1.data_ptr(), o1.data_ptr(), 96),
        "g2_mul (2^18)": lambda: c.g2_mul_dev(n2, dq2.data_ptr(), dk2.data_ptr(), o2.data_ptr(), 192),
        "pairing (2^16)": lambda: c.pair_dev(npair, dpp.data_ptr(), dqq.data_ptr(), gt.data_ptr()),
        "miller (2^16)": lambda: c.miller_dev(npair, dpp.data_ptr(), dqq.data_ptr(), mil.data_ptr()),
        "fexp (2^16)": lambda: c.gt_op_dev("fexp", npair, mil.data_ptr(), None, gt.data_ptr()),
        "msm (2^22)": lambda: c.g1_msm_dev(nm, dpm.data_ptr(), dkm.data_ptr(), om.data_ptr(), 96),
        "idle": lambda: time.sleep(0.02),
    }
    nsamp, gap = 4000, 30                      # ~100 us per sample
    buf = torch.zeros(2 * nsamp, dtype=torch.int64, device=dev)
    for name, fn in work.items():
        fn(); c.sync()
        buf.zero_()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        c.lib.c12381_exp_clock_probe(c.h, _p(buf.data_ptr()), nsamp, gap)
        reps = 0
        while time.perf_counter() - t0 < 0.40:
            fn(); reps += 1
            if reps % 4 == 0:
                c.sync()
        c.sync()
        el = time.perf_counter() - t0
        torch.cuda.synchronize(dev)
        h = buf.cpu().numpy().reshape(nsamp, 2)
        h = h[h[:, 1] != 0]
        span = h[:, 1] - h[0, 1]                                    # 100 MHz ticks since the probe started
        inrun = h[(span > 0.5 * el * 1e8) & (span < 0.95 * el * 1e8)]
        if len(inrun) < 8:
            print("%-16s too few samples (%d)" % (name, len(inrun)))
            continue
        ghz = np.diff(inrun[:, 0]) / np.diff(inrun[:, 1]) * 0.1
        print("%-16s launches %3d in %.2f s   in-kernel clock: median %.3f GHz  (p10 %.3f, p90 %.3f, %d samples)" % (
            name, reps, el, np.median(ghz), np.percentile(ghz, 10), np.percentile(ghz, 90), len(ghz)), flush=True)
    c.close()


if __name__ == "__main__":
    main()
