#!/usr/bin/env python3
"""A/B timing of builds of the same C ABI in ONE GPU session: every dominant entry point at its BASELINE size, device-resident inputs,
wall time per call over a few repetitions, and a SHA-256 of every output so that a variant that is faster AND different is caught at once.
The variants run in child processes (one library per process), interleaved round by round (A B A B ...) so that clock drift of the box
hits all of them alike.

    python tools/ab_bench.py [--legs g1,g2,pair,miller,fexp,msm,bbs] [--reps 3] [--rounds 2] default crypto12381_amd/lib/exp/libX.so ...

`default` = the product library.  Prints one table; exit status 1 if any digest differs from the first variant's or any child failed
(crash, time-out, no result line): the run stops at the first such child."""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ALL_LEGS = ["g1", "g2", "pair", "miller", "fexp", "msm", "bbs"]


def child(lib, legs, reps):
    import numpy as np
    import torch
    from crypto12381_amd import capi
    if lib != "default":
        capi.use_library(lib)
    from crypto12381_amd import Context
    from tools.prof_driver import G1, G2, sc
    c = Context(0)
    dev = torch.device("cuda", 0)
    s = torch.cuda.Stream(device=dev)
    c.set_stream(s.cuda_stream)

    def d(b):
        return torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)

    def digest(t):
        return hashlib.sha256(t.cpu().numpy().tobytes()).hexdigest()[:16]

    def timeit(fn):
        fn(); c.sync()
        best, tot = 1e9, 0.0
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            c.sync()
            dt = time.perf_counter() - t0
            best, tot = min(best, dt), tot + dt
        return best * 1e3, tot / reps * 1e3

    res = {}
    p1k = c.g1_mul(G1 * 1024, sc(3, 1024), 96)
    q1k = c.g2_mul(G2 * 1024, sc(4, 1024), 192)
    if "g1" in legs:
        n = 1 << 20
        dp, dk, o = d(p1k * (n // 1024)), d(sc(2, n)), torch.empty(96 * n, dtype=torch.uint8, device=dev)
        res["g1"] = (*timeit(lambda: c.g1_mul_dev(n, dp.data_ptr(), dk.data_ptr(), o.data_ptr(), 96)), digest(o))
        del dp, dk, o
    if "g2" in legs:
        n = 1 << 18
        dq, dk, o = d(q1k * (n // 1024)), d(sc(5, n)), torch.empty(192 * n, dtype=torch.uint8, device=dev)
        res["g2"] = (*timeit(lambda: c.g2_mul_dev(n, dq.data_ptr(), dk.data_ptr(), o.data_ptr(), 192)), digest(o))
        del dq, dk, o
    if {"pair", "miller", "fexp"} & set(legs):
        n = 1 << 16
        # distinct (P_i, Q_j) combinations: P index i mod 1024, Q index (5 i + i div 1024) mod 1024
        qq = b"".join(q1k[192 * ((5 * i + i // 1024) % 1024):192 * ((5 * i + i // 1024) % 1024) + 192] for i in range(n))
        dp, dq = d(p1k * (n // 1024)), d(qq)
        gt, mil = torch.empty(576 * n, dtype=torch.uint8, device=dev), torch.empty(576 * n, dtype=torch.uint8, device=dev)
        if "pair" in legs:
            res["pair"] = (*timeit(lambda: c.pair_dev(n, dp.data_ptr(), dq.data_ptr(), gt.data_ptr())), digest(gt))
        if "miller" in legs or "fexp" in legs:
            res["miller"] = (*timeit(lambda: c.miller_dev(n, dp.data_ptr(), dq.data_ptr(), mil.data_ptr())), digest(mil))
        if "fexp" in legs:
            res["fexp"] = (*timeit(lambda: c.gt_op_dev("fexp", n, mil.data_ptr(), None, gt.data_ptr())), digest(gt))
        del dp, dq, gt, mil
    if "msm" in legs:
        n = 1 << 22
        dp, dk, o = d(p1k * (n // 1024)), d(sc(6, n)), torch.empty(96, dtype=torch.uint8, device=dev)
        res["msm"] = (*timeit(lambda: c.g1_msm_dev(n, dp.data_ptr(), dk.data_ptr(), o.data_ptr(), 96)), digest(o))
        del dp, dk, o
    if "bbs" in legs:
        n = 1 << 18
        R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001

        def red(seed, m):
            a = np.frombuffer(sc(seed, m), dtype=np.uint8).reshape(m, 32).copy()
            a[:, 0] &= 0x3f
            return a
        pub = c.g1_mul_fixed(G1, red(51, 3).tobytes(), 96)
        g1p, h0, h = pub[:96], pub[96:192], pub[192:288]
        g2p = c.g2_mul_fixed(G2, red(52, 1).tobytes(), 192)
        gamma = red(53, 1).tobytes()
        w = c.g2_mul_fixed(g2p, gamma, 192)
        xs, rs, mm = red(54, n), red(55, n), red(56, n)
        A = c.bbs_plus_sign(g1p, h0, h, gamma, xs.tobytes(), rs.tobytes(), mm.tobytes())
        mm[7::1009, 31] ^= 1
        dA, dx, dr, dm = d(A), d(xs.tobytes()), d(rs.tobytes()), d(mm.tobytes())
        dpub = [d(b) for b in (g1p, g2p, h0, h, w)]
        ok = torch.empty(n, dtype=torch.uint8, device=dev)
        res["bbs"] = (*timeit(lambda: c.bbs_plus_verify_dev(n, 1, dpub[0].data_ptr(), dpub[1].data_ptr(), dpub[2].data_ptr(), dpub[3].data_ptr(),
                                                            dpub[4].data_ptr(), dA.data_ptr(), dx.data_ptr(), dr.data_ptr(), dm.data_ptr(), ok.data_ptr())), digest(ok))
        assert int(ok.sum().item()) == n - len(range(7, n, 1009)), "BBS+ verdicts wrong"
    c.close()
    print("ABJSON " + json.dumps(res), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--legs", default=",".join(ALL_LEGS))
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--child", default=None)
    ap.add_argument("libs", nargs="*", default=["default"])
    a = ap.parse_args()
    legs = a.legs.split(",")
    if a.child:
        child(a.child, legs, a.reps)
        return
    runs = {lib: [] for lib in a.libs}
    bad = False
    # A child that crashes, times out or prints no result stops the WHOLE comparison with a non-zero status: a variant that faulted the GPU must
    # not be followed by further launches on the same box (tools/gpu_pass.sh stops its pass on this status), and a table with a silent hole
    # in it is not a record.
    for rnd in range(a.rounds):
        for lib in a.libs:
            cmd = [sys.executable, os.path.abspath(__file__), "--child", lib, "--legs", a.legs, "--reps", str(a.reps)]
            try:
                r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
                rc, out, err = r.returncode, r.stdout, r.stderr
            except subprocess.TimeoutExpired as e:
                dec = lambda b: b.decode(errors="replace") if isinstance(b, bytes) else (b or "")
                rc, out, err = -9, dec(e.stdout), dec(e.stderr) + "\n(timed out after 900 s)"
            line = [l for l in out.splitlines() if l.startswith("ABJSON ")]
            if rc != 0 or not line:
                print("%s: FAILED rc=%d in round %d — stopping, no further variant is started\n--- stdout tail\n%s\n--- stderr tail\n%s"
                      % (lib, rc, rnd, out[-1500:], err[-2500:]), flush=True)
                bad = True
                break
            runs[lib].append(json.loads(line[0][7:]))
            print("round %d %-40s %s" % (rnd, os.path.basename(lib), "  ".join("%s %.2f" % (k, v[0]) for k, v in runs[lib][-1].items())), flush=True)
        if bad:
            break
    first = a.libs[0]
    print("\n%-8s" % "leg" + "".join("%28s" % os.path.basename(l)[:26] for l in a.libs))
    for leg in legs + (["miller"] if "fexp" in legs and "miller" not in legs else []):
        row = "%-8s" % leg
        for lib in a.libs:
            rs = [r[leg] for r in runs[lib] if leg in r]
            if not rs:
                row += "%28s" % "-"
                continue
            best, mean = min(x[0] for x in rs), sum(x[1] for x in rs) / len(rs)
            same = runs[first] and leg in runs[first][0] and all(x[2] == runs[first][0][leg][2] for x in rs)
            bad |= not same
            row += "%28s" % ("%.2f / %.2f ms %s" % (best, mean, "=" if same else "DIFF"))
        print(row)
    print("(best / mean wall ms per call; '=' : output digest equals the first variant's)")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
