#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY — generate tests/golden/*.json from the REAL reference.

Runs in the build container (where /root/reference exists): builds
oracle/_ref/libc12381_ref.so from the reference's own sources (oracle/Makefile), drives the
reference boundary (src/miracl_core_interface.cpp of the reference) through oracle/ref_wrap.cpp
and records inputs + expected canonical outputs.  The fixtures are DATA (hex strings of inputs
and outputs); no reference source text is stored.  Re-run:  python3 oracle/gen_golden.py
"""
from __future__ import annotations

import hashlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.bindings import Oracle, build  # noqa: E402

P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def prng(seed: int, i: int, nbytes: int = 64) -> int:
    """Counter-mode SHA-256 stream (the same generator the tests and bench.py use)."""
    out = b""
    ctr = 0
    while len(out) < nbytes:
        out += hashlib.sha256(b"c12381|%d|%d|%d" % (seed, i, ctr)).digest()
        ctr += 1
    return int.from_bytes(out[:nbytes], "big")


def scalars(seed: int, n: int, mod: int = R) -> bytes:
    return b"".join((prng(seed, i) % mod).to_bytes(32, "big") for i in range(n))


def hx(b: bytes, size: int):
    return [b[i:i + size].hex() for i in range(0, len(b), size)]


def main():
    build()
    ref = Oracle("reference")
    os.makedirs(OUT, exist_ok=True)
    g1, g2 = ref.g1_generator(), ref.g2_generator()
    inf1, inf2 = bytes(96), bytes(192)

    # ---------------- Fp
    edge = [0, 1, 2, P - 1, P - 2, (P - 1) // 2, P, P + 5, (1 << 384) - 1, 1 << 380]
    a = [prng(11, i, 48) % P for i in range(24)] + edge
    b = [prng(12, i, 48) % P for i in range(24)] + edge[::-1]
    ab = b"".join(x.to_bytes(48, "big") for x in a)
    bb = b"".join(x.to_bytes(48, "big") for x in b)
    fp = {"a": hx(ab, 48), "b": hx(bb, 48)}
    for op in ("mul", "add", "sub", "sqr", "neg", "inv"):
        o, _ = ref.fp_op(op, ab, bb)
        fp[op] = hx(o, 48)
    o, ok = ref.fp_op("sqrt", ab, bb)
    fp["sqrt_is_qr"] = list(ok)
    json.dump(fp, open(os.path.join(OUT, "fp.json"), "w"), indent=0)

    # ---------------- G1
    n = 24
    base_sc = scalars(21, n)
    pts = ref.g1_mul(g1 * n, base_sc, 96)
    edge_sc = [0, 1, 2, 3, R - 1, R, R + 1, 2 * R + 7, (1 << 256) - 1, 1 << 255, (1 << 128) - 1, 1 << 128,
               0xd201000000010000 ** 2, 0xd201000000010000 ** 2 - 1]
    sc = scalars(22, n - len(edge_sc) - 2) + b"".join(k.to_bytes(32, "big") for k in edge_sc) + scalars(23, 2)
    pts_e = pts[:96 * (n - 2)] + inf1 * 2          # last two lanes: point at infinity
    g1j = {"points": hx(pts_e, 96), "scalars": hx(sc, 32),
           "mul49": hx(ref.g1_mul(pts_e, sc, 49), 49), "mul96": hx(ref.g1_mul(pts_e, sc, 96), 96)}
    # additions incl. P+P, P+(-P), inf+P, P+inf, inf+inf
    neg = lambda p96: p96[:48] + ((P - int.from_bytes(p96[48:], "big")) % P).to_bytes(48, "big")
    p0, p1, p2 = pts[:96], pts[96:192], pts[192:288]
    A = p0 + p0 + p0 + inf1 + p1 + inf1 + p1 + p2
    B = p1 + p0 + neg(p0) + p1 + inf1 + inf1 + p2 + neg(p1)
    g1j["add_a"], g1j["add_b"] = hx(A, 96), hx(B, 96)
    g1j["add96"] = hx(ref.g1_add(A, B, 96), 96)
    g1j["add49"] = hx(ref.g1_add(A, B, 49), 49)
    # compression / decompression incl. rejects
    comp = ref.g1_compress(pts_e)
    bad = [b"\xff" * 49,                                # unit-tests/g1_point.cpp:132-138 (all-0xff rejects)
           b"\x02" + (5).to_bytes(48, "big"),           # x with x^3+4 a non-residue? (status recorded from the reference)
           b"\x03" + (1).to_bytes(48, "big"),
           b"\x02" + (P + 1).to_bytes(48, "big"),       # x >= p is taken mod p by the reference
           b"\x05" + pts[:48], b"\x80" + pts[:48], bytes(49)]
    cin = comp + b"".join(bad)
    dec, st = ref.g1_decompress(cin)
    g1j["compressed"] = hx(cin, 49)
    g1j["decompressed"] = hx(dec, 96)
    g1j["decompress_status"] = list(st)
    # MSM
    g1j["msm_n"] = n
    g1j["msm49"] = ref.g1_msm(pts_e, sc, 49, 4).hex()
    # points ON the curve but OUTSIDE the order-r subgroup (the reference performs no subgroup check, and its GLV
    # evaluation [k mod x^2]P - [k div x^2]phi(P) is then NOT [k]P): decompress small x values
    off, xv = [], 1
    while len(off) < 6:
        o, st = ref.g1_decompress(bytes([2 + (xv & 1)]) + xv.to_bytes(48, "big"))
        if st[0] == 1:
            off.append(o)
        xv += 1
    off = b"".join(off)
    off_sc = scalars(24, 6, 1 << 256)
    g1j["offsubgroup_points"], g1j["offsubgroup_scalars"] = hx(off, 96), hx(off_sc, 32)
    g1j["offsubgroup_mul96"] = hx(ref.g1_mul(off, off_sc, 96), 96)
    g1j["offsubgroup_msm49"] = ref.g1_msm(off, off_sc, 49, 2).hex()
    # the boundary's OTHER product: sum_of_products -> ECP_muln (plain Pippenger, true multiples) differs from the chain above off the subgroup
    g1j["offsubgroup_sum_of_products49"] = ref.g1_sum_of_products(off, off_sc, 49).hex()
    g1j["sum_of_products49"] = ref.g1_sum_of_products(pts_e, sc, 49).hex()
    # ... and scalars below x^2: glv() leaves u1 = r there, so the reference adds [r]phi(P) (pair_BLS12381.cpp:793-805, :899-906)
    X2 = 0xd201000000010000 ** 2
    small = [0, 1, 2, 0xd201000000010001, X2 - 1, X2, X2 + 1, R - 1, R, R + 1, 3 * X2, (1 << 127), 12345, R + 7]
    off_small_pts = b"".join(off[96 * (i % 6):96 * (i % 6) + 96] for i in range(len(small)))
    off_small_sc = b"".join(v.to_bytes(32, "big") for v in small)
    g1j["offsubgroup_small_points"], g1j["offsubgroup_small_scalars"] = hx(off_small_pts, 96), hx(off_small_sc, 32)
    g1j["offsubgroup_small_mul96"] = hx(ref.g1_mul(off_small_pts, off_small_sc, 96), 96)
    g1j["generator"] = g1.hex()
    json.dump(g1j, open(os.path.join(OUT, "g1.json"), "w"), indent=0)

    # ---------------- G2
    n2 = 16
    pts2 = ref.g2_mul(g2 * n2, scalars(31, n2), 192)
    edge_sc2 = [0, 1, 2, R - 1, R, R + 1, (1 << 256) - 1, 0xd201000000010000, 0xd201000000010000 ** 3]
    sc2 = scalars(32, n2 - len(edge_sc2) - 1) + b"".join(k.to_bytes(32, "big") for k in edge_sc2) + scalars(33, 1)
    pts2_e = pts2[:192 * (n2 - 1)] + inf2
    g2j = {"points": hx(pts2_e, 192), "scalars": hx(sc2, 32),
           "mul97": hx(ref.g2_mul(pts2_e, sc2, 97), 97), "mul192": hx(ref.g2_mul(pts2_e, sc2, 192), 192)}

    def neg2(q):
        yb, ya = int.from_bytes(q[96:144], "big"), int.from_bytes(q[144:192], "big")
        return q[:96] + ((P - yb) % P).to_bytes(48, "big") + ((P - ya) % P).to_bytes(48, "big")
    q0, q1, q2 = pts2[:192], pts2[192:384], pts2[384:576]
    A2 = q0 + q0 + q0 + inf2 + q1 + inf2 + q1
    B2 = q1 + q0 + neg2(q0) + q1 + inf2 + inf2 + q2
    g2j["add_a"], g2j["add_b"] = hx(A2, 192), hx(B2, 192)
    g2j["add192"] = hx(ref.g2_add(A2, B2, 192), 192)
    comp2 = ref.g2_compress(pts2_e)
    bad2 = [b"\x80" + bytes(96),                         # unit-tests/g2_point.cpp:112-118 (leading 0x80)
            b"\xff" * 97, b"\x02" + (1).to_bytes(48, "big") + (1).to_bytes(48, "big"),
            b"\x03" + (2).to_bytes(48, "big") + (0).to_bytes(48, "big"), bytes(97)]
    cin2 = comp2 + b"".join(bad2)
    dec2, st2 = ref.g2_decompress(cin2)
    g2j["compressed"] = hx(cin2, 97)
    g2j["decompressed"] = hx(dec2, 192)
    g2j["decompress_status"] = list(st2)
    off2, xv = [], 1
    while len(off2) < 6:
        o, st = ref.g2_decompress(bytes([2 + (xv & 1)]) + (7 * xv).to_bytes(48, "big") + xv.to_bytes(48, "big"))
        if st[0] == 1:
            off2.append(o)
        xv += 1
    off2 = b"".join(off2)
    off2_sc = scalars(34, 6, 1 << 256)
    g2j["offsubgroup_points"], g2j["offsubgroup_scalars"] = hx(off2, 192), hx(off2_sc, 32)
    g2j["offsubgroup_mul192"] = hx(ref.g2_mul(off2, off2_sc, 192), 192)   # = u0 Q - u1 psi(Q) + u2 psi^2(Q) - u3 psi^3(Q)
    # zero odd digits in base |x|: gs() turns them into r (BIG_modneg(0) = r, pair_BLS12381.cpp:868-871), adding [r]psi^i(Q)
    XA = 0xd201000000010000
    small2 = [0, 1, XA - 1, XA, XA + 1, XA ** 2, XA ** 2 + 5, XA ** 3, XA ** 3 + XA, XA ** 3 + 9, 7 * XA ** 2 + 3 * XA + 1, R - 1, R + 2, 1 << 190]
    off2_small_pts = b"".join(off2[192 * (i % 6):192 * (i % 6) + 192] for i in range(len(small2)))
    off2_small_sc = b"".join(v.to_bytes(32, "big") for v in small2)
    g2j["offsubgroup_small_points"], g2j["offsubgroup_small_scalars"] = hx(off2_small_pts, 192), hx(off2_small_sc, 32)
    g2j["offsubgroup_small_mul192"] = hx(ref.g2_mul(off2_small_pts, off2_small_sc, 192), 192)
    g2j["generator"] = g2.hex()
    json.dump(g2j, open(os.path.join(OUT, "g2.json"), "w"), indent=0)

    # ---------------- pairing / GT
    npair = 8
    P1 = pts[:96 * 6] + inf1 + g1
    Q2 = pts2[:192 * 5] + g2 + pts2[192 * 5:192 * 6] + inf2
    gt = ref.pair(P1, Q2)
    pj = {"g1": hx(P1, 96), "g2": hx(Q2, 192), "gt": hx(gt, 576)}
    pj["sha256_e_g1_g2"] = hashlib.sha256(ref.pair(g1, g2)).hexdigest()
    # pair2 = product of two pairings through pair_double_ate
    a1, a2, b1, b2 = P1[:96 * 4], Q2[:192 * 4], P1[96 * 4:], Q2[192 * 4:]
    pj["pair2"] = hx(ref.pair2(a1, a2, b1, b2), 576)
    # equality: e(xP, Q) == e(P, xQ) (true) and a corrupted one (false)
    xs = scalars(41, 4)
    xP = ref.g1_mul(pts[:96 * 4], xs, 96)
    xQ = ref.g2_mul(pts2[:192 * 4], xs, 192)
    eq_a1 = xP + xP[:96] + inf1
    eq_a2 = pts2[:192 * 4] + pts2[:192] + q0
    eq_b1 = pts[:96 * 4] + pts[96:192] + inf1
    eq_b2 = xQ + xQ[:192] + q1
    pj["eq_a1"], pj["eq_a2"], pj["eq_b1"], pj["eq_b2"] = hx(eq_a1, 96), hx(eq_a2, 192), hx(eq_b1, 96), hx(eq_b2, 192)
    pj["eq"] = list(ref.pair_eq(eq_a1, eq_a2, eq_b1, eq_b2))
    # equality, degenerate and adversarial rows: infinity on one side only, both sides trivial, negated arguments,
    # identical arguments, points outside the subgroups (the reference checks none of this, it just evaluates)
    def neg1(pt):
        y = int.from_bytes(pt[48:], "big")
        return pt[:48] + ((P - y) % P).to_bytes(48, "big")
    Pa, Pb, Qa, Qb = pts[:96], pts[96:192], pts2[:192], pts2[192:384]
    offp, offq = off[:96], off2[:192]
    rows = [(Pa, inf2, inf1, Qa), (Pa, inf2, Pa, Qa), (inf1, Qa, Pa, Qa), (Pa, Qa, Pa, Qa), (Pa, Qa, neg1(Pa), neg2(Qa)),
            (Pa, Qa, neg1(Pa), Qa), (Pa, Qa, Pb, Qa), (offp, Qa, offp, Qa), (offp, Qa, Pa, Qa), (Pa, offq, Pa, offq),
            (Pa, offq, Pb, offq), (offp, offq, neg1(offp), neg2(offq)), (inf1, inf2, Pa, inf2), (Pa, Qb, Pb, Qa)]
    e1, e2, e3, e4 = (b"".join(r[k] for r in rows) for k in range(4))
    pj["eq2_a1"], pj["eq2_a2"], pj["eq2_b1"], pj["eq2_b2"] = hx(e1, 96), hx(e2, 192), hx(e3, 96), hx(e4, 192)
    pj["eq2"] = list(ref.pair_eq(e1, e2, e3, e4))
    # GT ops
    gta, gtb = gt[:576 * 4], gt[576 * 4:]
    pj["gt_mul"] = hx(ref.gt_op("mul", gta, gtb), 576)
    pj["gt_conj"] = hx(ref.gt_op("conj", gta), 576)
    esc = b"".join(k.to_bytes(32, "big") for k in (0, 1, R - 1, prng(42, 0) % R))
    pj["gt_pow_exp"] = hx(esc, 32)
    pj["gt_pow"] = hx(ref.gt_op("pow", gta, esc), 576)
    json.dump(pj, open(os.path.join(OUT, "pairing.json"), "w"), indent=0)

    # ---------------- the reference's own seeded inputs (unit-tests/liner_pair.cpp:44)
    seed = b"pairing bilinearity seed"
    rs = ref.random_scalars(seed, 4)
    a_, b_, x_, y_ = (rs[32 * i:32 * i + 32] for i in range(4))
    Pp = ref.g1_mul(g1, a_, 96); Qq = ref.g2_mul(g2, b_, 192)
    lhs = ref.pair(ref.g1_mul(Pp, x_, 96), ref.g2_mul(Qq, y_, 192))
    xy = ((int.from_bytes(x_, "big") * int.from_bytes(y_, "big")) % R).to_bytes(32, "big")
    rhs = ref.gt_op("pow", ref.pair(Pp, Qq), xy)
    assert lhs == rhs, "bilinearity (config 1) failed on the reference itself"
    json.dump({"seed": seed.decode(), "scalars": hx(rs, 32), "P": Pp.hex(), "Q": Qq.hex(),
               "pair_Px_Qy": lhs.hex(), "xy": xy.hex()},
              open(os.path.join(OUT, "config1_bilinearity.json"), "w"), indent=0)

    # ---------------- hash-to-G1 (g1_point.hpp:219-234) and Zp helpers (zp_number.hpp) — SURVEY.md 8(f) rows 3, 4
    edge = [0, P, 1, P - 1, P + 1, (1 << 512) - 1, 2 * P, 11, R, 1 << 381]
    dig = b"".join(x.to_bytes(64, "big") for x in edge) + b"".join(hashlib.sha3_512(b"c12381 h2c|%d" % i).digest() for i in range(22))
    zedge = [0, 1, 2, R - 1, R, R + 1, (1 << 256) - 1, R - 2]
    za = b"".join(x.to_bytes(32, "big") for x in zedge) + scalars(31, 24, 1 << 256)
    zb = scalars(32, 8, 1 << 256) + b"".join(x.to_bytes(32, "big") for x in zedge) + scalars(33, 16, 1 << 256)
    json.dump({"digests": hx(dig, 64), "g1_from_hash_96": hx(ref.g1_from_hash(dig, 96), 96), "g1_from_hash_49": hx(ref.g1_from_hash(dig, 49), 49),
               "zp_from_hash": hx(ref.zp_from_hash(dig), 32),
               "zp_a": hx(za, 32), "zp_b": hx(zb, 32),
               **{"zp_" + op: hx(ref.zp_op(op, za, zb if op in ("mul", "add", "sub") else None), 32) for op in ("mul", "add", "sub", "neg", "inv")}},
              open(os.path.join(OUT, "hash_zp.json"), "w"), indent=0)
    print("golden vectors written to", OUT)


if __name__ == "__main__":
    main()
