// Fp4 = Fp2[s]/(s^2 - (1+i)) and Fp12 = Fp4[w]/(w^3 - s): the reference's 2-2-3 tower
// (config_field_BLS12381.h:32,36; fp4_BLS12381.cpp:243-364; fp12_BLS12381.cpp:117-939), kept on the
// device so that GT values need no basis change before their canonical 576-byte encoding.
//
// Bound contract of this layer (asserted by the host simulation): every fp4/fp12 function takes
// NORMALISED components (limb bound ~2^28) and returns normalised components; inside a function
// additions are lazy and one parallel carry round (fp_norm1) is inserted exactly where a sum feeds
// a multiplication.  Fp2 products are lazily reduced (fp2.hpp).
#pragma once
#include "fp2.hpp"

namespace c12381 {

struct fp4 { fp2 a, b; };            // a + b*s
struct fp12 { fp4 a, b, c; };        // a + b*w + c*w^2

// ------------------------------------------------------------------ Fp4
C12381_HD void fp4_add(fp4& r, const fp4& x, const fp4& y) { fp2_add(r.a, x.a, y.a); fp2_add(r.b, x.b, y.b); }
C12381_HD void fp4_sub(fp4& r, const fp4& x, const fp4& y) { fp2_sub(r.a, x.a, y.a); fp2_sub(r.b, x.b, y.b); }
C12381_HD void fp4_neg(fp4& r, const fp4& x) { fp2_neg(r.a, x.a); fp2_neg(r.b, x.b); }
C12381_HD void fp4_norm1(fp4& r, const fp4& x) { fp2_norm1(r.a, x.a); fp2_norm1(r.b, x.b); }
C12381_HD void fp4_conj(fp4& r, const fp4& x) { r.a = x.a; fp2_neg(r.b, x.b); }            // FP4_conj :162
C12381_HD void fp4_nconj(fp4& r, const fp4& x) { fp2_neg(r.a, x.a); r.b = x.b; }           // FP4_nconj :171
C12381_HD void fp4_zero(fp4& r) { fp2_zero(r.a); fp2_zero(r.b); }
C12381_HD void fp4_select(fp4& r, bool c, const fp4& x, const fp4& y) { fp2_select(r.a, c, x.a, y.a); fp2_select(r.b, c, x.b, y.b); }
// multiply by s (FP4_times_i :343): (a + b s) s = (1+i) b + a s      — lazy (limb bound of .a doubles)
C12381_HD void fp4_times_i(fp4& r, const fp4& x) {
    fp2 t;
    fp2_mul_ip(t, x.b);
    r.b = x.a;
    r.a = t;
}
// FP4_mul :274-304.  3 Fp2 products.
C12381_HDN void fp4_mul(fp4& w, const fp4& x, const fp4& y) {
    fp2 t1, t2, t3, t4;
    fp2_mul(t1, x.a, y.a);
    fp2_mul(t2, x.b, y.b);
    fp2_add(t3, y.b, y.a);
    fp2_add(t4, x.b, x.a);
    fp2_mul(t4, t4, t3);
    fp2_sub(t4, t4, t1);
    fp2_sub(t4, t4, t2);
    fp2_mul_ip(t3, t2);
    fp2_add(t3, t3, t1);
    fp2_norm1(w.b, t4);
    fp2_norm1(w.a, t3);
}
// FP4_sqr :243-271.  2 Fp2 products.
C12381_HDN void fp4_sqr(fp4& w, const fp4& x) {
    fp2 t1, t2, t3, wa;
    fp2_mul(t3, x.a, x.b);
    fp2_add(t1, x.a, x.b);
    fp2_mul_ip(t2, x.b);
    fp2_add(t2, x.a, t2);
    fp2_norm1(t2, t2);
    fp2_mul(wa, t1, t2);
    fp2_mul_ip(t2, t3);
    fp2_add(t2, t2, t3);
    fp2_sub(wa, wa, t2);
    fp2_dbl(t3, t3);
    fp2_norm1(w.a, wa);
    fp2_norm1(w.b, t3);
}
// FP4_inv :326-340
C12381_HDN void fp4_inv(fp4& w, const fp4& x) {
    fp2 t1, t2;
    fp2_sqr(t1, x.a);
    fp2_sqr(t2, x.b);
    fp2_mul_ip(t2, t2);
    fp2_sub(t1, t1, t2);
    fp2_norm1(t1, t1);
    fp2_inv(t1, t1);
    fp2_mul(w.a, t1, x.a);
    fp2_neg(t1, t1);
    fp2_mul(w.b, t1, x.b);
}
// FP4_frob :359-364 with f = FROB_F3 supplied by the caller
C12381_HD void fp4_frob(fp4& w, const fp4& x, const fp2& f) {
    fp2_conj(w.a, x.a);
    fp2 t;
    fp2_conj(t, x.b);
    fp2_mul(w.b, f, t);
}
C12381_HD void fp4_pmul(fp4& w, const fp4& x, const fp2& s) { fp2_mul(w.a, x.a, s); fp2_mul(w.b, x.b, s); }   // FP4_pmul :210

// ------------------------------------------------------------------ Fp12
C12381_HD void fp12_one(fp12& w) { fp4_zero(w.a); fp4_zero(w.b); fp4_zero(w.c); fp_one(w.a.a.a); }
C12381_HD void fp12_conj(fp12& w, const fp12& x) { fp4_conj(w.a, x.a); fp4_nconj(w.b, x.b); fp4_conj(w.c, x.c); }   // FP12_conj :117
C12381_HD void fp12_select(fp12& r, bool c, const fp12& x, const fp12& y) {
    fp4_select(r.a, c, x.a, y.a); fp4_select(r.b, c, x.b, y.b); fp4_select(r.c, c, x.c, y.c);
}
C12381_HD void fp4_addn(fp4& r, const fp4& x, const fp4& y) { fp4 t; fp4_add(t, x, y); fp4_norm1(r, t); }
C12381_HD void fp2_weak_reduce(fp2& r, const fp2& x) { fp_weak_reduce(r.a, x.a); fp_weak_reduce(r.b, x.b); }
C12381_HD void fp4_weak_reduce(fp4& r, const fp4& x) { fp2_weak_reduce(r.a, x.a); fp2_weak_reduce(r.b, x.b); }
C12381_HDN void fp12_weak_reduce(fp12& r, const fp12& x) { fp4_weak_reduce(r.a, x.a); fp4_weak_reduce(r.b, x.b); fp4_weak_reduce(r.c, x.c); }

// ---- fused helpers: the Fp12 routines below keep their Fp4 temporaries in private memory, so every pass over
// them is HBM traffic (profiles/r01_pmc_summary_before_table_fix.txt: the first pairing kernel moved ~2.5 MB per
// pairing and was bandwidth-bound).  Sums feeding a product are formed inside the product routine and each
// result is assembled in ONE pass over the products.
// (x1 + x2) * (y1 + y2)
C12381_HDN void fp4_mul_ss(fp4& w, const fp4& x1, const fp4& x2, const fp4& y1, const fp4& y2) {
    fp4 sx, sy;
    fp4_addn(sx, x1, x2);
    fp4_addn(sy, y1, y2);
    fp2 t1, t2, t3, t4;
    fp2_mul(t1, sx.a, sy.a);
    fp2_mul(t2, sx.b, sy.b);
    fp2_add(t3, sy.b, sy.a);
    fp2_add(t4, sx.b, sx.a);
    fp2_mul(t4, t4, t3);
    fp2_sub(t4, t4, t1);
    fp2_sub(t4, t4, t2);
    fp2_mul_ip(t3, t2);
    fp2_add(t3, t3, t1);
    fp2_norm1(w.b, t4);
    fp2_norm1(w.a, t3);
}
// (x1 + x2 + x3)^2
C12381_HDN void fp4_sqr_s3(fp4& w, const fp4& x1, const fp4& x2, const fp4& x3) {
    fp4 t, x;
    fp4_add(t, x1, x2); fp4_add(t, t, x3); fp4_norm1(x, t);
    fp2 t1, t2, t3, wa;
    fp2_mul(t3, x.a, x.b);
    fp2_add(t1, x.a, x.b);
    fp2_mul_ip(t2, x.b);
    fp2_add(t2, x.a, t2);
    fp2_norm1(t2, t2);
    fp2_mul(wa, t1, t2);
    fp2_mul_ip(t2, t3);
    fp2_add(t2, t2, t3);
    fp2_sub(wa, wa, t2);
    fp2_dbl(t3, t3);
    fp2_norm1(w.a, wa);
    fp2_norm1(w.b, t3);
}
// w = z0 + s(z3 - z2 - z4),  z1 - z0 - z2 + s z4,  z5 - z0 - z4 + z2     (one pass over the six products)
C12381_HDN void fp12_mul_combine(fp12& w, const fp4& z0, const fp4& z1, const fp4& z2, const fp4& z3, const fp4& z4, const fp4& z5) {
    fp4 t, u;
    fp4_sub(t, z3, z2); fp4_sub(t, t, z4); fp4_times_i(u, t); fp4_add(u, u, z0);
    fp4_norm1(w.a, u);
    fp4_sub(t, z1, z0); fp4_sub(t, t, z2); fp4_times_i(u, z4); fp4_add(t, t, u);
    fp4_norm1(w.b, t);
    fp4_sub(t, z5, z0); fp4_sub(t, t, z4); fp4_add(t, t, z2);
    fp4_norm1(w.c, t);
}
// FP12_mul :246-299 (Karatsuba over Fp4: 6 Fp4 products = 18 Fp2 products)
C12381_HDN void fp12_mul(fp12& w, const fp12& x, const fp12& y) {
    fp4 z0, z1, z2, z3, z4, z5;
    fp4_mul(z0, x.a, y.a);
    fp4_mul(z2, x.b, y.b);
    fp4_mul(z4, x.c, y.c);
    fp4_mul_ss(z1, x.a, x.b, y.a, y.b);
    fp4_mul_ss(z3, x.b, x.c, y.b, y.c);
    fp4_mul_ss(z5, x.a, x.c, y.a, y.c);
    fp12_mul_combine(w, z0, z1, z2, z3, z4, z5);
}
// wa = A + s 2B,  wb = s C + 2D,  wc = S - A - 2B - C - 2D
C12381_HDN void fp12_sqr_combine(fp12& w, const fp4& A, const fp4& B, const fp4& C, const fp4& D, const fp4& S) {
    fp4 b2, d2, t, u;
    fp4_add(b2, B, B); fp4_add(d2, D, D);
    fp4_times_i(u, b2); fp4_add(u, u, A);
    fp4_norm1(w.a, u);
    fp4_times_i(u, C); fp4_add(u, u, d2);
    fp4_norm1(w.b, u);
    fp4_sub(t, S, A); fp4_sub(t, t, b2); fp4_sub(t, t, C); fp4_sub(t, t, d2);
    fp4_norm1(w.c, t);
}
// FP12_sqr :190-238 (Chung-Hasan SQR2: 3 Fp4 squarings + 2 Fp4 products).  w may alias x.
C12381_HDN void fp12_sqr(fp12& w, const fp12& x) {
    fp4 A, B, C, D, S;
    fp4_sqr(A, x.a);
    fp4_mul(B, x.b, x.c);
    fp4_sqr(C, x.c);
    fp4_mul(D, x.a, x.b);
    fp4_sqr_s3(S, x.a, x.b, x.c);
    fp12_sqr_combine(w, A, B, C, D, S);
}
// FP12_usqr :147-186 (Granger-Scott; equals sqr only for unitary elements), split by output coefficient so
// that every Fp4 operand is read once and every result written once (the squaring ladder of the final
// exponentiation was the largest source of private-memory traffic):
//   wa = 3 xa^2 - 2 conj(xa)                       needs xa only
//   wb = 3 s xc^2 + 2 conj(xb),  wc = 3 xb^2 - 2 conj(xc)   need xb and xc
// `reduce` re-bounds the results (fp_weak_reduce) in the same pass.  Outputs may alias inputs.
// A/B switch (pairing3.hpp says why it is off)
// fp4_sqr body without the final carry round: limbs up to 2^30 (w.a) / 2^29 (w.b) — for callers that add or select
// before they normalise anyway (the unitary squaring)
C12381_HD void fp4_sqr_core_raw(fp4& w, const fp4& x) {
    fp2 t1, t2, t3, wa;
    fp2_mul(t3, x.a, x.b);
    fp2_add(t1, x.a, x.b);
    fp2_mul_ip(t2, x.b);
    fp2_add(t2, x.a, t2);
    fp2_norm1(t2, t2);
    fp2_mul(wa, t1, t2);
    fp2_mul_ip(t2, t3);
    fp2_add(t2, t2, t3);
    fp2_sub(w.a, wa, t2);
    fp2_dbl(w.b, t3);
}
C12381_HD void fp4_sqr_core(fp4& w, const fp4& x) {            // fp4_sqr body, inlined into the callers below
    fp4 r;
    fp4_sqr_core_raw(r, x);
    fp2_norm1(w.a, r.a);
    fp2_norm1(w.b, r.b);
}
C12381_HDN void fp12_usqr_a(fp4& wa, const fp4& xa, bool reduce) {
    fp4 A, t, u;
    fp4_sqr_core(A, xa);
    fp4_add(t, A, A); fp4_add(t, t, A);
    fp4_nconj(u, xa); fp4_add(u, u, u); fp4_add(t, t, u);
    if (reduce) fp4_weak_reduce(wa, t); else fp4_norm1(wa, t);
}
C12381_HDN void fp12_usqr_bc(fp4& wb, fp4& wc, const fp4& xb, const fp4& xc, bool reduce) {
    fp4 B, C, t, u, rb, rc;
    fp4_sqr_core(B, xc);
    fp4_sqr_core(C, xb);
    fp4_times_i(t, B); fp4_norm1(t, t); fp4_add(u, t, t); fp4_add(t, u, t);
    fp4_conj(u, xb); fp4_add(u, u, u); fp4_add(rb, t, u);
    fp4_add(t, C, C); fp4_add(t, t, C);
    fp4_nconj(u, xc); fp4_add(u, u, u); fp4_add(rc, t, u);
    if (reduce) { fp4_weak_reduce(wb, rb); fp4_weak_reduce(wc, rc); } else { fp4_norm1(wb, rb); fp4_norm1(wc, rc); }
}
C12381_HD void fp12_usqr_r(fp12& w, const fp12& x, bool reduce) {
    fp12_usqr_a(w.a, x.a, reduce);
    fp12_usqr_bc(w.b, w.c, x.b, x.c, reduce);
}
C12381_HD void fp12_usqr(fp12& w, const fp12& x) { fp12_usqr_r(w, x, false); }
// FP12_inv :627-664
C12381_HDN void fp12_inv(fp12& w, const fp12& x) {
    fp4 f0, f1, f2, f3, t;
    fp4_sqr(f0, x.a); fp4_mul(f1, x.b, x.c); fp4_times_i(t, f1); fp4_sub(f0, f0, t); fp4_norm1(f0, f0);
    fp4_sqr(f1, x.c); fp4_times_i(t, f1); fp4_mul(f2, x.a, x.b); fp4_sub(f1, t, f2); fp4_norm1(f1, f1);
    fp4_sqr(f2, x.b); fp4_mul(f3, x.a, x.c); fp4_sub(f2, f2, f3); fp4_norm1(f2, f2);
    fp4_mul(f3, x.b, f2); fp4_times_i(f3, f3);
    fp4_mul(t, f0, x.a); fp4_add(f3, t, f3);
    fp4_mul(t, f1, x.c); fp4_times_i(t, t); fp4_add(f3, t, f3);
    fp4_norm1(f3, f3);
    fp4_inv(f3, f3);
    fp4_norm1(f3, f3);
    fp4_mul(w.a, f0, f3); fp4_mul(w.b, f1, f3); fp4_mul(w.c, f2, f3);
}
// FP12_frob :867-880 (f, f^2, f^3 are compile-time constants)
C12381_HDN void fp12_frob(fp12& w, const fp12& x) {
    fp2 f, f2, f3;
    fp2_set_const(f, FROB_F_A, FROB_F_B);
    fp2_set_const(f2, FROB_F2_A, FROB_F2_B);
    fp2_set_const(f3, FROB_F3_A, FROB_F3_B);
    fp4 a, b, c;
    fp4_frob(a, x.a, f3); fp4_frob(b, x.b, f3); fp4_frob(c, x.c, f3);
    fp4_pmul(b, b, f); fp4_pmul(c, c, f2);
    // conj leaves .a lazily negated only: still normalised
    w.a = a; w.b = b; w.c = c;
}
// X * la (dense Fp4 product) and (X.a * l2, X.b * l2) in one routine: X is read once
C12381_HDN void fp4_mul_la_l2(fp4& p, fp2& q0, fp2& q1, const fp4& x, const fp4& la, const fp2& l2) {
    fp2_mul(q0, x.a, l2);
    fp2_mul(q1, x.b, l2);
    fp2 t1, t2, t3, t4;
    fp2_mul(t1, x.a, la.a);
    fp2_mul(t2, x.b, la.b);
    fp2_add(t3, la.b, la.a);
    fp2_add(t4, x.b, x.a);
    fp2_mul(t4, t4, t3);
    fp2_sub(t4, t4, t1);
    fp2_sub(t4, t4, t2);
    fp2_mul_ip(t3, t2);
    fp2_add(t3, t3, t1);
    fp2_norm1(p.b, t4);
    fp2_norm1(p.a, t3);
}
// f *= line, line = [l0, l1] + [0, l2] w^2  (the M-type sparse element of PAIR_line pair_BLS12381.cpp:129-143;
// replaces FP12_ssmul's dense x sparser branch fp12_BLS12381.cpp:440-487).  15 Fp2 products.
//   wa = fa*la + s*(fb*lc),  wb = fb*la + s*(fc*lc),  wc = fc*la + fa*lc,   lc = l2 s:
//   X*lc = ((1+i) x1 l2, x0 l2),   s*(X*lc) = ((1+i) x0 l2, (1+i) x1 l2)
C12381_HDN void fp12_mul_line(fp12& f, const fp2& l0, const fp2& l1, const fp2& l2) {
    fp4 la; la.a = l0; la.b = l1;
    fp4 pa, pb, pc;
    fp2 a0, a1, b0, b1, c0, c1;
    fp4_mul_la_l2(pa, a0, a1, f.a, la, l2);
    fp4_mul_la_l2(pb, b0, b1, f.b, la, l2);
    fp4_mul_la_l2(pc, c0, c1, f.c, la, l2);
    fp2 t;
    fp2_mul_ip(t, b0); fp2_add(pa.a, pa.a, t);
    fp2_mul_ip(t, b1); fp2_add(pa.b, pa.b, t);
    fp2_mul_ip(t, c0); fp2_add(pb.a, pb.a, t);
    fp2_mul_ip(t, c1); fp2_add(pb.b, pb.b, t);
    fp2_mul_ip(t, a1); fp2_add(pc.a, pc.a, t);
    fp2_add(pc.b, pc.b, a0);
    fp4_norm1(f.a, pa); fp4_norm1(f.b, pb); fp4_norm1(f.c, pc);
}
// r = a^|x| for the curve parameter |x| = 0xd201000000010000 (FP12_pow :736-774 specialised to the
// one exponent the final exponentiation uses; unitary input, so plain square-and-multiply with
// Granger-Scott squarings gives the same element as the reference's signed-digit ladder).
C12381_HDN void fp12_pow_x_unitary(fp12& r, const fp12& a) {
    fp12 w = a;
#pragma unroll 1
    for (int i = 62; i >= 0; --i) {
        // the squaring carries the linear term -2 conj(w): the integer representative doubles each
        // step, so re-bound it every 2nd step (data-independent schedule), in the same pass
        fp12_usqr_r(w, w, (i & 1) == 0);
        if ((BLS_X >> i) & 1ull) fp12_mul(w, w, a);
    }
    r = w;
}

// FP12_pow :736-774 for a per-lane exponent e < 2^256 used AS GIVEN (no reduction): signed-digit ladder over
// (3e, e) with Granger-Scott squarings, so — like the reference — it is a power only for unitary inputs.
// Lanes hold different exponents: the schedule is made uniform by always forming both candidates and
// selecting (no divergence); "started" tracks each lane's own top bit.
C12381_HDN void fp12_pow_generic(fp12& r, const fp12& a, const uint32_t (&e)[8]) {
    uint32_t e3[9];
    {
        uint64_t c = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) { c += (uint64_t)e[i] * 3u; e3[i] = (uint32_t)c; c >>= 32; }
        e3[8] = (uint32_t)c;
    }
    int nb = 0;                                     // number of bits of 3e
#pragma unroll 1
    for (int i = 0; i < 9 * 32; ++i) if ((e3[i >> 5] >> (i & 31)) & 1u) nb = i + 1;
    fp12 w = a, ac, one;
    fp12_conj(ac, a);
    fp12_one(one);
#pragma unroll 1
    for (int i = 257; i >= 1; --i) {
        const bool active = i <= nb - 2;            // the reference starts from w = a at bit nb-1
        fp12 t, m, t2;
        fp12_usqr_r(t, w, true);
        const int b3 = (int)((e3[i >> 5] >> (i & 31)) & 1u);
        const int b1 = i < 256 ? (int)((e[i >> 5] >> (i & 31)) & 1u) : 0;
        const int bt = b3 - b1;
        fp12_select(m, bt < 0, ac, a);
        fp12_mul(t2, t, m);
        fp12 nxt;
        fp12_select(nxt, bt != 0, t2, t);
        fp12_select(w, active, nxt, w);
    }
    fp12_select(r, nb == 0, one, w);
}

}  // namespace c12381
