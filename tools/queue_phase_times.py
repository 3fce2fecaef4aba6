#!/usr/bin/env python3
"""Per-phase task durations and hand-over waits of pair3_queue_kernel from the diagnostic stamps (C12381_PAIR_STAMPS).
usage (GPU box): C12381_PAIR_STAMPS=/tmp/st.bin python tools/queue_phase_times.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
path = os.environ.setdefault("C12381_PAIR_STAMPS", "/tmp/c12381_stamps.bin")
import tools.libsel  # noqa: E402,F401  (C12381_LIB -> capi.use_library)
from crypto12381_amd import Context  # noqa: E402
from tools.prof_driver import G1, G2, sc  # noqa: E402

c = Context(0)
n = 1 << 16
p = c.g1_mul(G1 * 1024, sc(3, 1024), 96) * (n // 1024)
q = c.g2_mul(G2 * 1024, sc(4, 1024), 192) * (n // 1024)
c.pair(p, q)
c.pair(p, q)
c.sync()
a = np.fromfile(path, dtype=np.uint64).reshape(-1, 4)
groups_all = (n + 20) // 21
nwaves = 2048
# the kernel's split (k_pair3.hip queue_direct_groups): whole groups first, the last third (between half a grid and a grid) queued
queued = groups_all if groups_all <= nwaves else min(max(groups_all // 3, nwaves // 2), 2 * nwaves)
if os.environ.get("C12381_QUEUE_GROUPS"):                # experiments build: the override the library reads
    queued = min(int(os.environ["C12381_QUEUE_GROUPS"]), groups_all)
groups = queued
print("groups %d: %d claimed whole, %d through the queue (stamps cover the queued ones)" % (groups_all, groups_all - queued, queued))
a = a[: groups * 10]
claim, start, end = a[:, 0].astype(np.float64), a[:, 1].astype(np.float64), a[:, 2].astype(np.float64)
hw = a[:, 3]
hwid, xcc = (hw & np.uint64(0xffffffff)).astype(np.int64), (hw >> np.uint64(32)).astype(np.int64) & 0xf
simd, cu, sh, se = (hwid >> 4) & 3, (hwid >> 8) & 0xf, (hwid >> 12) & 1, (hwid >> 13) & 7
ph = np.arange(groups * 10) // groups
run, wait = end - start, start - claim
t_first = claim[claim > 0].min()
print("first claim .. last end: %.0f Kcyc;  last claim at %.0f Kcyc" % ((end.max() - t_first) / 1e3, (claim.max() - t_first) / 1e3))
print("phase  tasks  run Kcyc: p5     p25     p50     p75     p95    mean | wait mean    max | claimed at Kcyc: p5 p50 p95")
for k in range(10):
    m = ph == k
    q = np.percentile(run[m], [5, 25, 50, 75, 95]) / 1e3
    cq = np.percentile(claim[m] - t_first, [5, 50, 95]) / 1e3
    print("%5d  %5d  %15.0f %7.0f %7.0f %7.0f %7.0f %7.0f | %9.1f %7.0f | %7.0f %7.0f %7.0f" % (k, m.sum(), q[0], q[1], q[2], q[3], q[4], run[m].mean() / 1e3, wait[m].mean() / 1e3, wait[m].max() / 1e3, cq[0], cq[1], cq[2]))
slot = ((xcc * 8 + se) * 2 + sh) * 16 + cu
key = slot * 4 + simd
print("distinct (xcc, se, sh, cu): %d   distinct SIMDs: %d   tasks per SIMD: min %d max %d" % (len(np.unique(slot)), len(np.unique(key)), np.bincount(key).min() if len(key) else 0, np.bincount(key).max()))
# per-SIMD total run time: a SIMD that hosts two wavefronts accumulates twice its wall time
tot = np.bincount(key, weights=run)
cnt = np.bincount(key)
nz = cnt > 0
print("per-SIMD sum of run cycles: p5 %.1f  p50 %.1f  p95 %.1f  max %.1f (M cycles)" % tuple(np.percentile(tot[nz], [5, 50, 95, 100]) / 1e6))
print("per-XCC task counts:", np.bincount(xcc).tolist())
m9 = ph >= 4
print("final-exp task run time by xcc (Kcyc):", [round(run[m9 & (xcc == x)].mean() / 1e3) for x in range(8)])
np.save(os.path.join(os.path.dirname(path) if os.path.dirname(path) else ".", "stamps_summary.npy"), np.stack([ph, run, wait, key]))
