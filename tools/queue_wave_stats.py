#!/usr/bin/env python3
"""Where the wavefronts of the work-queue kernels spend a launch of 2^16 elements (experiments build, C12381_PAIR_STAMPS): per wavefront the
cycles in whole groups, in queue tasks (state loads / stores included), in hand-over waits and idle before / after (k_pair3.hip
queue_wave_stats), split by the wavefront's age on its SIMD (the older one is served first).  One table for pairings, Miller loops and final
exponentiations.   usage (GPU box): C12381_LIB=crypto12381_amd/lib/libc12381_hip_exp.so python tools/queue_wave_stats.py [n]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
path = os.environ.setdefault("C12381_PAIR_STAMPS", "/tmp/c12381_stamps.bin")
import tools.libsel  # noqa: E402,F401
from crypto12381_amd import Context  # noqa: E402
from tools.prof_driver import G1, G2, sc  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 16
c = Context(0)
p = c.g1_mul(G1 * 1024, sc(3, 1024), 96) * (n // 1024)
q = c.g2_mul(G2 * 1024, sc(4, 1024), 192) * (n // 1024)
STAMP_WAVES = 4096


def stats(label, run):
    run(); run()                                   # the second launch's stamps are the ones read (warm caches, workspace in place)
    c.sync()
    a = np.fromfile(path, dtype=np.uint64)
    w = a[-STAMP_WAVES * 12:].reshape(-1, 12)
    w = w[w[:, 1] > 0]
    entry, exit_, whole, nwhole, task, ntask, wait = (w[:, k].astype(np.float64) for k in range(7))
    load, store, reread = w[:, 8].astype(np.float64), w[:, 9].astype(np.float64), w[:, 10].astype(np.float64)
    hwid = (w[:, 7] & np.uint64(0xffffffff)).astype(np.int64)
    xcc = (w[:, 7] >> np.uint64(32)).astype(np.int64) & 0xf
    wave_id, simd, cu, sh, se = hwid & 0xf, (hwid >> 4) & 3, (hwid >> 8) & 0xf, (hwid >> 12) & 1, (hwid >> 13) & 7
    simd_key = ((((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd)
    # s_memtime is not one clock for the chip (the stamps of different CUs lie up to tens of ms apart): every time is taken relative to the first
    # entry seen on the SAME CU (its eight wavefronts start within microseconds of the launch), the span is the longest CU's
    cu_key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    for k in np.unique(cu_key):
        m = cu_key == k
        base = entry[m].min()
        entry[m] -= base; exit_[m] -= base
    t0, t1 = 0.0, exit_.max()
    span = t1 - t0
    # age on the SIMD: the wavefront that entered first
    older = np.zeros(len(w), bool)
    for k in np.unique(simd_key):
        idx = np.nonzero(simd_key == k)[0]
        older[idx[np.argmin(entry[idx])]] = True
    print("== %s: %d elements, %d wavefronts on %d SIMDs; launch span %.2f M cycles (first entry .. last exit)" % (label, n, len(w), len(np.unique(simd_key)), span / 1e6))
    print("   %-8s %6s | %9s %9s %9s %9s %9s %9s | whole groups, tasks per wavefront" % ("", "waves", "whole", "tasks", "wait", "claim&c", "start", "tail"))
    for name, m in (("older", older), ("younger", ~older), ("all", np.ones(len(w), bool))):
        k = m.sum()
        if not k:
            continue
        other = (exit_[m] - entry[m]) - whole[m] - task[m] - wait[m]
        f = lambda x: 100.0 * x.sum() / (k * span)
        print("   %-8s %6d | %8.2f%% %8.2f%% %8.2f%% %8.2f%% %8.2f%% %8.2f%% | %.2f  %.2f" % (
            name, k, f(whole[m]), f(task[m]), f(wait[m]), f(other), f(entry[m] - t0), f(t1 - exit_[m]), nwhole[m].mean(), ntask[m].mean()))
    print("   cycles per whole group: older p50 %.2f M, younger p50 %.2f M;  per task: older %.3f M, younger %.3f M" % (
        np.median((whole / np.maximum(nwhole, 1))[older & (nwhole > 0)]) / 1e6 if (older & (nwhole > 0)).any() else 0,
        np.median((whole / np.maximum(nwhole, 1))[~older & (nwhole > 0)]) / 1e6 if (~older & (nwhole > 0)).any() else 0,
        (task[older].sum() / max(ntask[older].sum(), 1)) / 1e6, (task[~older].sum() / max(ntask[~older].sum(), 1)) / 1e6))
    nt = max(ntask.sum(), 1)
    print("   inside a task (mean K cycles): state arrives %.1f, arithmetic %.1f, state leaves + publish %.1f" % (
        load.sum() / nt / 1e3, (task.sum() - load.sum() - store.sum()) / nt / 1e3, store.sum() / nt / 1e3) + (";  re-reads of the first value: %d in %d tasks" % (reread.sum(), nt) if reread.sum() else ""))
    ex = np.sort(exit_ - t0)
    print("   exits (M cycles after the first entry): p5 %.2f  p25 %.2f  p50 %.2f  p75 %.2f  p95 %.2f  last %.2f" % tuple(np.percentile(ex, [5, 25, 50, 75, 95, 100]) / 1e6))
    # SIMD-level: cycles in which 0 / 1 / 2 of the SIMD's wavefronts were inside the kernel and not in a hand-over wait cannot be told from totals;
    # what can: the time after the SIMD's first wavefront left (one wavefront alone) and after both left
    alone, empty = [], []
    for k in np.unique(simd_key):
        e = np.sort(exit_[simd_key == k])
        if len(e) == 2:
            alone.append(e[1] - e[0]); empty.append(t1 - e[1])
    if alone:
        print("   per SIMD: one wavefront alone at the end %.2f%% of the span (p50 %.2f M cycles), both gone %.2f%%" % (
            100 * np.mean(alone) / span, np.median(alone) / 1e6, 100 * np.mean(empty) / span))


stats("pairings (pair3_queue_kernel)", lambda: c.pair(p, q))
m = c.miller(p, q)
stats("Miller loops (miller3_queue_kernel)", lambda: c.miller(p, q))
stats("final exponentiations (fexp3_queue_kernel)", lambda: c.fexp(m))
c.close()
