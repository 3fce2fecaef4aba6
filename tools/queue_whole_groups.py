#!/usr/bin/env python3
"""Whole-group timing of pair3_queue_kernel from the diagnostic stamps (experiments build, C12381_PAIR_STAMPS): how the two wavefronts of a SIMD
share it.  Per whole group: start and end relative to its wavefront's entry into the kernel, by hardware wave slot; then the queued tasks the same
wavefront slot ran afterwards.   usage (GPU box): C12381_LIB=<libc12381_hip_exp.so> python tools/queue_whole_groups.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
path = os.environ.setdefault("C12381_PAIR_STAMPS", "/tmp/c12381_stamps.bin")
import tools.libsel  # noqa: E402,F401
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
ngroups, nwaves = (n + 20) // 21, 2048
queued = min(max(ngroups // 3, nwaves // 2), 2 * nwaves)
if os.environ.get("C12381_QUEUE_GROUPS"):
    queued = min(int(os.environ["C12381_QUEUE_GROUPS"]), ngroups)
ndirect = ngroups - queued
w = a[queued * 10: queued * 10 + ndirect]
w = w[w[:, 2] > 0]
entry, g0, g1_ = (w[:, k].astype(np.float64) for k in range(3))
hw = w[:, 3]
hwid, xcc = (hw & np.uint64(0xffffffff)).astype(np.int64), (hw >> np.uint64(32)).astype(np.int64) & 0xf
wave_id, simd, cu, sh, se = hwid & 0xf, (hwid >> 4) & 3, (hwid >> 8) & 0xf, (hwid >> 12) & 1, (hwid >> 13) & 7
print("whole groups with stamps: %d of %d; hardware wave slots in use: %s" % (len(w), ndirect, np.bincount(wave_id).tolist()))
simd_key = ((((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd)
first = (g0 - entry) < 50e3                       # the group a wavefront took first
print("run time of a wavefront's FIRST whole group by hardware wave slot (M cycles): p5 / p50 / p95")
for s in np.unique(wave_id):
    m = first & (wave_id == s)
    if m.sum():
        r = (g1_[m] - g0[m]) / 1e6
        print("  slot %d: %5d groups   %.2f / %.2f / %.2f" % (s, m.sum(), *np.percentile(r, [5, 50, 95])))
# pairs of wavefronts on one SIMD: ratio of their first-group run times
d = {}
for k, r, s in zip(simd_key[first], (g1_[first] - g0[first]) / 1e6, wave_id[first]):
    d.setdefault(int(k), []).append((int(s), float(r)))
pairs = [sorted(v) for v in d.values() if len(v) == 2]
if pairs:
    lo = np.array([min(v[0][1], v[1][1]) for v in pairs]); hi = np.array([max(v[0][1], v[1][1]) for v in pairs])
    print("SIMDs with two first groups: %d;  shorter / longer run time (M cycles) p50: %.2f / %.2f;  ratio p5 / p50 / p95: %.2f / %.2f / %.2f"
          % (len(pairs), np.median(lo), np.median(hi), *np.percentile(lo / hi, [5, 50, 95])))
    lower_slot_shorter = np.mean([v[0][1] < v[1][1] for v in pairs])
    print("the wavefront in the LOWER slot has the shorter run time on %.0f %% of the SIMDs" % (100 * lower_slot_shorter))
second = ~first
if second.sum():
    print("later whole groups: %d, start %.2f .. %.2f M cycles after entry, run p50 %.2f" % (second.sum(), (g0[second] - entry[second]).min() / 1e6,
          (g0[second] - entry[second]).max() / 1e6, np.median(g1_[second] - g0[second]) / 1e6))
t = a[: queued * 10]
t = t[t[:, 2] > 0]
print("queued tasks with stamps: %d; run time p50 by phase (K cycles): %s" % (len(t), [int(np.median((t[k * queued:(k + 1) * queued, 2] - t[k * queued:(k + 1) * queued, 1]).astype(np.float64)) / 1e3) for k in range(10)]))
