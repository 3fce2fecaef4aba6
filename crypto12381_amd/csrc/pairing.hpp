// Optimal-ate pairing on BLS12-381: Miller loop and final exponentiation, one pairing per lane
// (replaces PAIR_ate / PAIR_line / PAIR_double / PAIR_add pair_BLS12381.cpp:40-144, 425-505 and
//  PAIR_fexp :629-755 of the reference's vendored MIRACL-core).
//
// The loop parameter is the curve constant |x| = 0xd201000000010000, so the double/add schedule is a
// compile-time constant shared by every lane: 63 doubling steps, 5 addition steps, no divergence.
// Only the value AFTER the final exponentiation is canonical; the Miller value is
// computed as the same field element as the reference's (same line functions, same tower), so the
// final result — including the degenerate inputs the reference does not special-case — is bit-exact.
#pragma once
#include "fp12.hpp"
#include "g2.hpp"

namespace c12381 {

// Doubling step: line through T,T evaluated at P = (px, py), then T = 2T.
// PAIR_double :40-78 + PAIR_line :119-143:  l0 = -2YZ(1+i) * py,  l1 = 3b'Z^2 - Y^2,  l2 = 3X^2 * px
C12381_HDN void miller_dbl_step(g2p& T, fp2& l0, fp2& l1, fp2& l2, const fp& px, const fp& py) {
    fp2 xx, t0, t1, t2b, aa;
    fp2_sqr(xx, T.x);
    g2_dbl_ex(T, t0, t1, t2b);
    fp2_dbl(aa, t1); fp2_neg(aa, aa); fp2_mul_ip(aa, aa);      // -2YZ(1+i), limb bound 2^30
    fp2_mul_fp(l0, aa, py);
    fp2 bb;
    fp2_sub(bb, t2b, t0);
    fp2_norm1(l1, bb);
    fp2 cc;
    fp2_dbl(cc, xx); fp2_add(cc, cc, xx);                       // 3X^2
    fp2_mul_fp(l2, cc, px);
}
// Addition step: line through T,Q (Q affine) at P, then T = T + Q.   PAIR_add :81-116
//   l0 = (X1 - Z1 X2)(1+i) * py,  l1 = (Y1 - Z1 Y2) X2 - (X1 - Z1 X2) Y2,  l2 = -(Y1 - Z1 Y2) * px
C12381_HDN void miller_add_step(g2p& T, const g2p& Q, fp2& l0, fp2& l1, fp2& l2, const fp& px, const fp& py) {
    fp2 zy, zx, aa, cc, t1, bb;
    fp2_mul(zy, T.z, Q.y);
    fp2_mul(zx, T.z, Q.x);
    fp2_sub(aa, T.x, zx); fp2_norm1(aa, aa);
    fp2_sub(cc, T.y, zy); fp2_norm1(cc, cc);
    fp2_mul(t1, aa, Q.y);
    fp2_mul(bb, cc, Q.x);
    fp2_sub(bb, bb, t1);
    fp2_norm1(l1, bb);
    fp2 aai;
    fp2_mul_ip(aai, aa);
    fp2_mul_fp(l0, aai, py);
    fp2 ncc;
    fp2_neg(ncc, cc);
    fp2_mul_fp(l2, ncc, px);
    g2_add(T, Q);
}

// f = Miller_{|x|}(Q, P) conjugated (x < 0).  P affine G1 (px, py) or infinity; Q affine G2 or "infinity".
// G1 infinity gives 1 (PAIR_ate :448-449).  G2 infinity is NOT special-cased by the reference: it runs
// the loop on (0 : 1 : 0) with affine view (0, 1); we do the same so the final value agrees.
C12381_HDN void miller_loop(fp12& f, const fp& px, const fp& py, bool p_inf, const fp2& qx, const fp2& qy, bool q_inf) {
    g2p Q, T;
    Q.x = qx; Q.y = qy; fp2_one(Q.z);
    {
        g2p inf;
        g2_set_inf(inf);
        fp2_select(Q.x, q_inf, inf.x, Q.x);
        fp2_select(Q.y, q_inf, inf.y, Q.y);
        fp2_select(Q.z, q_inf, inf.z, Q.z);
    }
    T = Q;
    fp12_one(f);
    // n = |x|, n3 = 3n; digit_i = bit_i(n3) - bit_i(n), i = 64 .. 1   (PAIR_ate :466-483)
    constexpr unsigned __int128 N1 = (unsigned __int128)BLS_X;
    constexpr unsigned __int128 N3 = N1 * 3;
#pragma unroll 1
    for (int i = 64; i >= 1; --i) {
        fp12_sqr(f, f);
        fp2 l0, l1, l2;
        miller_dbl_step(T, l0, l1, l2, px, py);
        fp12_mul_line(f, l0, l1, l2);
        const int bt = (int)((N3 >> i) & 1) - (int)((N1 >> i) & 1);
        if (bt != 0) {                      // wave-uniform: depends only on the curve constant
            g2p S = Q;
            if (bt < 0) g2_neg(S, Q);
            miller_add_step(T, S, l0, l1, l2, px, py);
            fp12_mul_line(f, l0, l1, l2);
        }
    }
    fp12 c, one;
    fp12_conj(c, f);
    fp12_one(one);
    fp12_select(f, p_inf, one, c);
}

// PAIR_fexp :629-755 (BLS12 branch :711-753, eprint 2020/875): f^((p^12-1)/r * 3)-style exponent, exactly
// the reference's sequence so that the result is the same element of GT.
C12381_HDN void fp12_pow_x(fp12& r, const fp12& a) {      // a^x, x negative: conj(a^|x|) for unitary a
    fp12 t;
    fp12_pow_x_unitary(t, a);
    fp12_conj(r, t);
}
C12381_HDN void final_exp(fp12& r) {
    fp12 t0, y0, y1, t;
    // easy part: r^((p^6 - 1)(p^2 + 1))
    fp12_inv(t0, r);
    fp12_conj(t, r);
    fp12_mul(r, t, t0);
    t0 = r;
    fp12_frob(t, r); fp12_frob(r, t);
    fp12_mul(t, r, t0); r = t;
    // hard part
    fp12_usqr(y1, r); fp12_mul(t, y1, r); y1 = t;                          // r^3
    fp12_pow_x(y0, r); fp12_conj(t0, r); fp12_mul(r, y0, t0);              // r^(x-1)
    fp12_pow_x(y0, r); fp12_conj(t0, r); fp12_mul(t, y0, t0); r = t;       // r^(x-1)
    fp12_pow_x(y0, r); fp12_frob(t0, r); fp12_mul(r, y0, t0);              // ^(x+p)
    fp12_pow_x(y0, r); fp12_pow_x(t, y0); y0 = t;                          // r^(x^2)
    fp12_frob(t, r); fp12_frob(t0, t);                                     // r^(p^2)
    fp12_mul(t, y0, t0); y0 = t;
    fp12_conj(t0, r);
    fp12_mul(r, y0, t0);                                                   // ^(x^2+p^2-1)
    fp12_mul(t, r, y1); r = t;
}

// FP12_isunity fp12_BLS12381.cpp:71
C12381_HD bool fp12_is_one(const fp12& x) {
    fp d;
    fp one;
    fp_one(one);
    fp_sub(d, x.a.a.a, one);
    bool ok = fp_is_zero(d);
    ok = ok & fp_is_zero(x.a.a.b) & fp_is_zero(x.a.b.a) & fp_is_zero(x.a.b.b);
    ok = ok & fp_is_zero(x.b.a.a) & fp_is_zero(x.b.a.b) & fp_is_zero(x.b.b.a) & fp_is_zero(x.b.b.b);
    ok = ok & fp_is_zero(x.c.a.a) & fp_is_zero(x.c.a.b) & fp_is_zero(x.c.b.a) & fp_is_zero(x.c.b.b);
    return ok;
}

// the 12 Fp coordinates in the order of FP12_toOctet fp12_BLS12381.cpp:923-929 (c, b, a; each Fp4 b, a; each Fp2 b, a)
C12381_HD const fp& fp12_coord(const fp12& x, int j) {
    const fp4& q = j < 4 ? x.c : (j < 8 ? x.b : x.a);
    const fp2& d = ((j & 3) < 2) ? q.b : q.a;
    return (j & 1) ? d.a : d.b;
}
C12381_HD fp& fp12_coord_mut(fp12& x, int j) {
    fp4& q = j < 4 ? x.c : (j < 8 ? x.b : x.a);
    fp2& d = ((j & 3) < 2) ? q.b : q.a;
    return (j & 1) ? d.a : d.b;
}

}  // namespace c12381
