#!/usr/bin/env python3
"""A few bucket products of 2^22 terms and nothing else (for a kernel trace of the MSM pipeline: rocprofv3 --kernel-trace -- python3 tools/msm_only.py)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools.libsel  # noqa: E402,F401  (C12381_LIB -> capi.use_library)
from crypto12381_amd import Context  # noqa: E402
from tools.prof_driver import G1, sc  # noqa: E402

c = Context(0)
dev = torch.device("cuda", 0)
s = torch.cuda.Stream(device=dev)
c.set_stream(s.cuda_stream)
nm = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 22)
p = c.g1_mul(G1 * 1024, sc(6, 1024), 96)
pm = (p * ((nm * 96 + len(p) - 1) // len(p)))[: nm * 96]
dpm = torch.frombuffer(bytearray(pm), dtype=torch.uint8).to(dev)
ks = bytearray(sc(8, nm))
if len(sys.argv) > 2 and sys.argv[2] == "edge":          # scalar 1 in lane 1: the small-scalar bucket is in use (as on bench.py's edge lanes)
    ks[32:64] = (1).to_bytes(32, "big")
dkm = torch.frombuffer(ks, dtype=torch.uint8).to(dev)
om = torch.empty(96, dtype=torch.uint8, device=dev)
for rep in range(4):
    t0 = time.perf_counter()
    c.g1_msm_dev(nm, dpm.data_ptr(), dkm.data_ptr(), om.data_ptr(), 96)
    t1 = time.perf_counter()
    c.sync()
    t2 = time.perf_counter()
    print("msm 2^%d: enqueue %.2f ms, total %.2f ms" % (nm.bit_length() - 1, (t1 - t0) * 1e3, (t2 - t0) * 1e3), flush=True)
