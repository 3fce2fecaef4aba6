#!/usr/bin/env python3
"""Operation counts of the device routines AS BUILT, from the host simulation (tests/host_sim, C12381_CHECK_BOUNDS build):
fp.hpp counts every scanned column of a 14 x 14 limb product (27 per product) and every Montgomery reduction.

For each unit of work the script reports
    products, reductions            field-level work of this library's own operation sequence
    issued_mad                      multiply-adds in the 14 x 28-bit format: 196 per product + 210 per reduction
    algorithmic_mac32               the same sequence priced as SURVEY.md 8(d) prices the reference's: 144 per product + 156 per reduction
next to the reference's count for the unit where it has one.  Counts are per element: the slope between two
batch sizes of identical elements, so one-off work (tables, line coefficients) drops out.

    python tools/count_ops.py [out.json]
"""
import ctypes
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import cat, golden, scalars  # noqa: E402

SIM_DIR = os.path.join(ROOT, "tests", "host_sim")
sz = ctypes.c_size_t
REF = {"g1_mul": 579_456, "g2_mul": 1_163_808, "miller": 2_177_268, "fexp": 2_097_972, "pairing": 4_275_240}


def load_sim():
    so = os.path.join(SIM_DIR, "libsim.so")
    subprocess.run(["g++", "-O1", "-std=c++17", "-DC12381_CHECK_BOUNDS", "-fPIC", "-shared", "-pthread", "-o", so, os.path.join(SIM_DIR, "sim.cpp")], check=True)
    return ctypes.CDLL(so)


def main():
    sim = load_sim()

    def count(fn):
        sim.sim_ops_reset()
        fn()
        c, r = ctypes.c_ulonglong(), ctypes.c_ulonglong()
        sim.sim_ops_read(ctypes.byref(c), ctypes.byref(r))
        return c.value, r.value

    def per_element(run, n_lo=2, n_hi=6):
        c0, r0 = count(lambda: run(n_lo))
        c1, r1 = count(lambda: run(n_hi))
        d = n_hi - n_lo
        return (c1 - c0) / d / 27.0, (r1 - r0) / d

    g1, g2, pr = golden("g1"), golden("g2"), golden("pairing")
    p1, q2 = cat(g1["points"])[:96], cat(g2["points"])[:192]
    k = scalars(77, 1)                                   # one uniform scalar below r
    gt_in = cat(pr["miller"])[:576] if "miller" in pr else None
    out = {}

    def unit(name, run, note):
        prods, reds = per_element(run)
        d = {"products": round(prods, 1), "reductions": round(reds, 1), "issued_mad": round(196 * prods + 210 * reds),
             "algorithmic_mac32": round(144 * prods + 156 * reds), "note": note}
        if name in REF:
            d["reference_mac32"] = REF[name]
            d["ratio_to_reference"] = round(d["algorithmic_mac32"] / REF[name], 3)
        out[name] = d
        print("%-22s products %9.1f reductions %9.1f issued mads %9d algorithmic MAC32 %9d %s" % (
            name, prods, reds, d["issued_mad"], d["algorithmic_mac32"], ("(reference %d)" % REF[name]) if name in REF else ""))

    def buf(n, b):
        return ctypes.create_string_buffer(b * n)

    unit("g1_mul", lambda n: sim.sim_g1_mul_batch(sz(n), p1 * n, k * n, buf(n, 96), 96), "g1_scalar_mul + per-element affine conversion (device: one inversion per 16 elements)")
    unit("g2_mul", lambda n: sim.sim_g2h_mul_batch(sz(n), q2 * n, k * n, buf(n, 192), 192), "two-lane form (k_g2h.hip), both lanes; per-element affine conversion as above")
    unit("g2_mul_one_lane", lambda n: sim.sim_g2_mul_batch(sz(n), q2 * n, k * n, buf(n, 192), 192), "one-lane form (BBS+ internals)")
    unit("miller", lambda n: sim.sim_miller3_batch(sz(n), p1 * n, q2 * n, buf(n, 576)), "three-lane Miller loop, sum over the three lanes")
    unit("pairing", lambda n: sim.sim_pair3_batch(sz(n), p1 * n, q2 * n, buf(n, 576)), "three-lane Miller loop + final exponentiation")
    mil = buf(1, 576)
    sim.sim_miller3_batch(sz(1), p1, q2, mil)
    unit("fexp", lambda n: sim.sim_gt3_op_batch(3, sz(n), mil.raw * n, None, buf(n, 576)), "three-lane final exponentiation")
    gen1, gen2 = bytes.fromhex(g1["generator"]), bytes.fromhex(g2["generator"])
    unit("pair2_fixed", lambda n: sim.sim_pair2_fixed_batch(sz(n), p1 * n, q2, p1 * n, gen2, buf(n, 576)),
         "BBS+ kernel: product of two pairings against two FIXED G2 arguments (line tables) + final exponentiation")
    unit("g1_fixed_mul", lambda n: sim.sim_g1_fixed_mul_batch(sz(n), gen1, k * n, buf(n, 96)), "table-driven g^x (32 mixed additions) + per-element affine conversion")
    # one BBS+ verification as the pipeline runs it (1 message block): x A generic, r h0 and m h1 from tables, three point additions,
    # the two-table product of pairings
    add = 12 * 144 + 9 * 156
    v = out["g1_mul"]["algorithmic_mac32"] + 2 * out["g1_fixed_mul"]["algorithmic_mac32"] + 4 * add + out["pair2_fixed"]["algorithmic_mac32"]
    out["bbs_plus_verify_pipeline"] = {"algorithmic_mac32": v, "note": "g1_mul + 2 g1_fixed_mul + 4 point additions + pair2_fixed (own operation sequence)",
                                       "reference_sequence_mac32": REF["g2_mul"] + 2 * REF["g1_mul"] + 2 * REF["miller"] + REF["fexp"]}
    print("bbs_plus_verify_pipeline algorithmic MAC32 %d (reference sequence %d)" % (v, out["bbs_plus_verify_pipeline"]["reference_sequence_mac32"]))
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
