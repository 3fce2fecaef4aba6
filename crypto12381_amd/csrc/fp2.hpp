// Fp2 = Fp[i]/(i^2 + 1)  (replaces FP2_mul/FP2_sqr/FP2_inv/FP2_mul_ip/... of the reference's
// fp2_BLS12381.cpp:185-396).  Element a + b*i.
//
// Multiplication and squaring are lazily reduced: each output coordinate is ONE column scan over
// the sum of its products followed by ONE Montgomery reduction (fp_reduce_cols), so an Fp2 product
// costs 4 x 196 multiply-adds + 2 reductions and — more important on this machine — its outputs are
// already normalised (limb bound 2^28) while its inputs may carry limb bound 2^29 (e.g. one lazy
// add/sub of normalised values) with no carry propagation anywhere in between.
#pragma once
#include "fp.hpp"

namespace c12381 {

struct fp2 { fp a, b; };

C12381_HD void fp2_add(fp2& r, const fp2& x, const fp2& y) { fp_add(r.a, x.a, y.a); fp_add(r.b, x.b, y.b); }
C12381_HD void fp2_sub(fp2& r, const fp2& x, const fp2& y) { fp_sub(r.a, x.a, y.a); fp_sub(r.b, x.b, y.b); }
C12381_HD void fp2_neg(fp2& r, const fp2& x) { fp_neg(r.a, x.a); fp_neg(r.b, x.b); }
C12381_HD void fp2_dbl(fp2& r, const fp2& x) { fp_dbl(r.a, x.a); fp_dbl(r.b, x.b); }
C12381_HD void fp2_conj(fp2& r, const fp2& x) { r.a = x.a; fp_neg(r.b, x.b); }
C12381_HD void fp2_zero(fp2& r) { fp_zero(r.a); fp_zero(r.b); }
C12381_HD void fp2_one(fp2& r) { fp_one(r.a); fp_zero(r.b); }
C12381_HD void fp2_norm1(fp2& r, const fp2& x) { fp_norm1(r.a, x.a); fp_norm1(r.b, x.b); }
C12381_HD void fp2_select(fp2& r, bool c, const fp2& x, const fp2& y) { fp_select(r.a, c, x.a, y.a); fp_select(r.b, c, x.b, y.b); }
C12381_HD void fp2_mul_small(fp2& r, const fp2& x, int32_t k) { fp_mul_small(r.a, x.a, k); fp_mul_small(r.b, x.b, k); }
// multiply by (1 + i)  (FP2_mul_ip :373, QNRI = 0): (a - b) + (a + b) i   — lazy, limb bound doubles
C12381_HD void fp2_mul_ip(fp2& r, const fp2& x) {
    fp ta, tb;
    fp_sub(ta, x.a, x.b);
    fp_add(tb, x.a, x.b);
    r.a = ta; r.b = tb;
}
C12381_HD bool fp2_is_zero(const fp2& x) { return fp_is_zero(x.a) & fp_is_zero(x.b); }

// r = x * y.  Operand limb bounds: LBx * LBy <= 2^58 (e.g. 2^29 each).  Output normalised.
C12381_HD void fp2_mul(fp2& r, const fp2& x, const fp2& y) {
    fp ra, rb, nxb;
    fp_raw_neg(nxb, x.b);
    fp_reduce_cols(ra, [&](int k, int64_t& acc) { fp_col_acc(acc, x.a, y.a, k); fp_col_acc(acc, nxb, y.b, k); });
    fp_reduce_cols(rb, [&](int k, int64_t& acc) { fp_col_acc(acc, x.a, y.b, k); fp_col_acc(acc, x.b, y.a, k); });
    C12381_BOUNDS({ check_actual(x.a, "fp2_mul"); check_actual(x.b, "fp2_mul"); check_actual(y.a, "fp2_mul"); check_actual(y.b, "fp2_mul");
                    set_lazy_bounds(ra, x.a.lb * y.a.lb + x.b.lb * y.b.lb, x.a.vb * y.a.vb + x.b.vb * y.b.vb, "fp2_mul.a");
                    set_lazy_bounds(rb, x.a.lb * y.b.lb + x.b.lb * y.a.lb, x.a.vb * y.b.vb + x.b.vb * y.a.vb, "fp2_mul.b"); })
    r.a = ra; r.b = rb;
}
// r = x * y + (linear terms injected into the two reductions, fp_reduce_cols_inj): inj_a / inj_b add the addends of the real / imaginary
// coordinate; ba / bb carry their bounds for the checker builds (sum |k| VB, sum |k| LB).  The result is normalised: what used to be
// "product, lazy additions, carry round" is one product.
struct fp_injb { double vb, lb; };
#ifdef C12381_CHECK_BOUNDS
#define C12381_INJB(vb_, lb_) c12381::fp_injb{(vb_), (lb_)}
#else
#define C12381_INJB(vb_, lb_) c12381::fp_injb{0.0, 0.0}
#endif
template <class IA, class IB>
C12381_HD void fp2_mul_inj(fp2& r, const fp2& x, const fp2& y, IA inj_a, IB inj_b, const fp_injb& ba, const fp_injb& bb) {
    fp ra, rb, nxb;
    fp_raw_neg(nxb, x.b);
    fp_reduce_cols_inj(ra, [&](int k, int64_t& acc) { fp_col_acc(acc, x.a, y.a, k); fp_col_acc(acc, nxb, y.b, k); }, inj_a);
    fp_reduce_cols_inj(rb, [&](int k, int64_t& acc) { fp_col_acc(acc, x.a, y.b, k); fp_col_acc(acc, x.b, y.a, k); }, inj_b);
    (void)ba; (void)bb;
    C12381_BOUNDS({ check_actual(x.a, "fp2_mul_inj"); check_actual(x.b, "fp2_mul_inj"); check_actual(y.a, "fp2_mul_inj"); check_actual(y.b, "fp2_mul_inj");
                    set_inj_bounds(ra, x.a.lb * y.a.lb + x.b.lb * y.b.lb, x.a.vb * y.a.vb + x.b.vb * y.b.vb, ba.vb, ba.lb, "fp2_mul_inj.a");
                    set_inj_bounds(rb, x.a.lb * y.b.lb + x.b.lb * y.a.lb, x.a.vb * y.b.vb + x.b.vb * y.a.vb, bb.vb, bb.lb, "fp2_mul_inj.b"); })
    r.a = ra; r.b = rb;
}
C12381_HD void fp2_norm1_dbl(fp2& r, const fp2& x) { fp_norm1_dbl(r.a, x.a); fp_norm1_dbl(r.b, x.b); }
// r = x^2 = (a^2 - b^2) + 2ab i.  Operand limb bound <= 2^29.
C12381_HD void fp2_sqr(fp2& r, const fp2& x) {
    fp ra, rb, a2, nb2, nb;
    fp_raw_dbl(a2, x.a);
    fp_raw_neg_dbl(nb2, x.b);
    fp_raw_neg(nb, x.b);
    fp_reduce_cols(ra, [&](int k, int64_t& acc) { fp_col_sqr_acc(acc, x.a, a2, x.a, k); fp_col_sqr_acc(acc, x.b, nb2, nb, k); });
    fp_reduce_cols(rb, [&](int k, int64_t& acc) { fp_col_acc(acc, a2, x.b, k); });
    C12381_BOUNDS({ check_actual(x.a, "fp2_sqr"); check_actual(x.b, "fp2_sqr");
                    set_lazy_bounds(ra, x.a.lb * x.a.lb + x.b.lb * x.b.lb, x.a.vb * x.a.vb + x.b.vb * x.b.vb, "fp2_sqr.a");
                    set_lazy_bounds(rb, 2 * x.a.lb * x.b.lb, 2 * x.a.vb * x.b.vb, "fp2_sqr.b"); })
    r.a = ra; r.b = rb;
}
// r = a*b + c*d  (SUB: a*b - c*d) with ONE reduction per coordinate: four limb products per column instead of two reduced
// Fp2 products (saves 2 x 210 multiply-adds; the lazily reduced sums of the complete addition formulas, like fp_mul2 in G1).
// Needs 14 * sum(LB*LB over the four products of a coordinate) + 14 * 2^56 + 2^40 < 2^63, e.g. one normalised operand
// (2^28) against one of limb bound 2^29 in both products.
template <bool SUB>
C12381_HD void fp2_mul2(fp2& r, const fp2& a, const fp2& b, const fp2& c, const fp2& d) {
    fp ra, rb, nab, ca, cb, ncb;                 // (ca, cb) = +-c, ncb = -(+-c.b)
    fp_raw_neg(nab, a.b);
    if (SUB) { fp_raw_neg(ca, c.a); fp_raw_neg(cb, c.b); ncb = c.b; }
    else { ca = c.a; cb = c.b; fp_raw_neg(ncb, c.b); }
    fp_reduce_cols_static(ra, [&](int k, int64_t& acc) { fp_col_acc(acc, a.a, b.a, k); fp_col_acc(acc, nab, b.b, k);
                                                    fp_col_acc(acc, ca, d.a, k); fp_col_acc(acc, ncb, d.b, k); });
    fp_reduce_cols_static(rb, [&](int k, int64_t& acc) { fp_col_acc(acc, a.a, b.b, k); fp_col_acc(acc, a.b, b.a, k);
                                                    fp_col_acc(acc, ca, d.b, k); fp_col_acc(acc, cb, d.a, k); });
    C12381_BOUNDS({ check_actual(a.a, "fp2_mul2"); check_actual(a.b, "fp2_mul2"); check_actual(b.a, "fp2_mul2"); check_actual(b.b, "fp2_mul2");
                    check_actual(c.a, "fp2_mul2"); check_actual(c.b, "fp2_mul2"); check_actual(d.a, "fp2_mul2"); check_actual(d.b, "fp2_mul2");
                    set_lazy_bounds(ra, a.a.lb * b.a.lb + a.b.lb * b.b.lb + c.a.lb * d.a.lb + c.b.lb * d.b.lb,
                                    a.a.vb * b.a.vb + a.b.vb * b.b.vb + c.a.vb * d.a.vb + c.b.vb * d.b.vb, "fp2_mul2.a");
                    set_lazy_bounds(rb, a.a.lb * b.b.lb + a.b.lb * b.a.lb + c.a.lb * d.b.lb + c.b.lb * d.a.lb,
                                    a.a.vb * b.b.vb + a.b.vb * b.a.vb + c.a.vb * d.b.vb + c.b.vb * d.a.vb, "fp2_mul2.b"); })
    r.a = ra; r.b = rb;
}
// The constants of psi = untwist-Frobenius-twist (ECP2_frob ecp2_BLS12381.cpp:579-590 with X = 1/f) have special shapes on this
// curve (tools/gen_consts.py asserts them): PSI1_X = c i, PSI3_X = -i, PSI1_Y = a (1 - i), PSI3_Y = -PSI1_Y, PSI2_Y = -1.
//   conj(x) * (c i)      = c x.b + c x.a i                               two Fp products instead of an Fp2 product
//   conj(x) * (-i)       = -x.b - x.a i                                  no product
//   conj(y) * a (1 - i)  = a (y.a - y.b) - a (y.a + y.b) i               two Fp products; `neg` negates the result (the sign of the
//                                                                         digit and of psi^3 ride along for free)
C12381_HD void fp2_conj_mul_ci(fp2& r, const fp2& x, const fp& c) { fp ta, tb; fp_mul(ta, x.b, c); fp_mul(tb, x.a, c); r.a = ta; r.b = tb; }
C12381_HD void fp2_conj_mul_neg_i(fp2& r, const fp2& x) { fp ta, tb; fp_neg(ta, x.b); fp_neg(tb, x.a); r.a = ta; r.b = tb; }
C12381_HD void fp2_conj_mul_a1mi(fp2& r, const fp2& y, const fp& a, bool neg) {
    fp d, s, nd, ns;
    fp_sub(d, y.a, y.b);                       // y.a - y.b
    fp_add(s, y.a, y.b); fp_neg(s, s);         // -(y.a + y.b)
    fp_neg(nd, d); fp_neg(ns, s);
    fp_select(d, neg, nd, d); fp_select(s, neg, ns, s);
    fp_mul(r.a, d, a); fp_mul(r.b, s, a);
}
// r = x * s for s in Fp  (FP2_pmul :231)
C12381_HD void fp2_mul_fp(fp2& r, const fp2& x, const fp& s) { fp_mul(r.a, x.a, s); fp_mul(r.b, x.b, s); }
// r = 1/x  (FP2_inv :334): conj(x) / (a^2 + b^2)
C12381_HDN void fp2_inv(fp2& r, const fp2& x) {
    fp n, ni, nb, a2, b2;
    fp_raw_dbl(a2, x.a);
    fp_raw_dbl(b2, x.b);
    fp_reduce_cols(n, [&](int k, int64_t& acc) { fp_col_sqr_acc(acc, x.a, a2, x.a, k); fp_col_sqr_acc(acc, x.b, b2, x.b, k); });
    C12381_BOUNDS(set_lazy_bounds(n, x.a.lb * x.a.lb + x.b.lb * x.b.lb, x.a.vb * x.a.vb + x.b.vb * x.b.vb, "fp2_inv");)
    fp_inv(ni, n);
    fp_mul(r.a, x.a, ni);
    fp_neg(nb, x.b);
    fp_mul(r.b, nb, ni);
}
// FP2_sign :168-181: parity of a, or of b when a == 0
C12381_HD int fp2_sign(const fp2& x) {
    fp ca, cb;
    fp_from_mont_canonical(ca, x.a);
    fp_from_mont_canonical(cb, x.b);
    int32_t o = 0;
#pragma unroll
    for (int i = 0; i < NL; ++i) o |= ca.l[i];
    return o == 0 ? (cb.l[0] & 1) : (ca.l[0] & 1);
}
// FP2_qr :446 + FP2_sqrt :460-521 (complex method) in one pass.  Returns false when the norm a^2+b^2 is a
// non-residue (or 0); on success w is the root of sign 0, exactly as the reference returns it.
C12381_HDN bool fp2_sqrt(fp2& w, const fp2& u) {
    fp n, w1, w1inv, w2, hb, half, ra, rainv, rb;
    fp ua2, ub2;
    fp_raw_dbl(ua2, u.a);
    fp_raw_dbl(ub2, u.b);
    fp_reduce_cols(n, [&](int k, int64_t& acc) { fp_col_sqr_acc(acc, u.a, ua2, u.a, k); fp_col_sqr_acc(acc, u.b, ub2, u.b, k); });
    C12381_BOUNDS(set_lazy_bounds(n, u.a.lb * u.a.lb + u.b.lb * u.b.lb, u.a.vb * u.a.vb + u.b.vb * u.b.vb, "fp2_sqrt");)
    const bool norm_qr = fp_sqrt_progen(w1, w1inv, n);          // w1 = sqrt(a^2 + b^2)
    fp_set_const(half, FP_HALF);
    fp_add(w2, u.a, w1);
    fp_norm1(w2, w2);
    fp_mul(w2, w2, half);                                       // (a + w1) / 2
    fp_mul(hb, u.b, half);                                      // b / 2
    const bool qr = fp_sqrt_progen(ra, rainv, w2);              // ra = sqrt(w2) or sqrt(-w2); rainv = 1/ra
    fp_mul(rb, hb, rainv);                                      // (b/2) / ra
    fp2 r;
    fp_select(r.a, qr, ra, rb);
    fp_select(r.b, qr, rb, ra);
    fp2 nr;
    fp2_neg(nr, r);
    fp2_select(w, fp2_sign(r) != 0, nr, r);
    return norm_qr;
}
C12381_HD void fp2_set_const(fp2& r, const int32_t (&ca)[NL], const int32_t (&cb)[NL]) { fp_set_const(r.a, ca); fp_set_const(r.b, cb); }

}  // namespace c12381
