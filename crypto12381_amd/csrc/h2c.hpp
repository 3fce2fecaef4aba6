// Hash-to-G1 as crypto12381 defines it (G1Point::from_hash, include/crypto12381/g1_point.hpp:219-234):
//   64-byte digest (SHA3-512 of the caller's serialisation, big-endian integer) -> mod p (fixed_time_mod)
//   -> FP_nres -> ECP_map2point (ecp_BLS12381.cpp:1495-1626: simplified SWU onto the 11-isogenous curve
//   E': y^2 = x^3 + A'x + B' with Z = 11, one exponentiation serving the QR test, the inversion and the square
//   root, then the 11-isogeny in projective form) -> ECP_cfp (:1252-1273: multiplication by 1 - x).
// This is NOT RFC 9380 hash_to_curve (one field element, no expand_message); it shares the RFC's constants.
// The field sequence of the reference is followed step by step so that the degenerate inputs (u = 0, a zero
// denominator: inverse of 0 is 0, 0 counts as a non-residue) give the same point on E' as the reference does.
#pragma once
#include "codec.hpp"
#include "g1.hpp"

namespace c12381 {

// 64 big-endian bytes as 16 raw words -> the digest mod p, Montgomery form
C12381_HD void fp_from_digest64(fp& u, const uint32_t* raw) {
    uint32_t hi[12], lo[12];
#pragma unroll
    for (int i = 0; i < 8; ++i) hi[i] = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) hi[8 + i] = bswap32(raw[i]);
#pragma unroll
    for (int i = 0; i < 12; ++i) lo[i] = bswap32(raw[4 + i]);
    fp h, l, c;
    fp_from_words_be(h, hi);
    fp_from_words_be(l, lo);
    fp_set_const(c, FP_2_384);
    fp_mul(h, h, c);
    fp_add(u, h, l);
    fp_norm1(u, u);
}

// Horner evaluation of a constant-coefficient polynomial (ascending table, `monic`: leading 1 not stored)
template <int N>
C12381_HD void iso_eval(fp& r, const int32_t (&cs)[N][NL], const fp& x, bool monic) {
    fp acc, c;
    fp_set_const(c, cs[N - 1]);
    if (monic) { fp_add(acc, x, c); fp_norm1(acc, acc); } else { acc = c; }
#pragma unroll 1
    for (int i = N - 2; i >= 0; --i) {
        fp_mul(acc, acc, x);
        fp_set_const(c, cs[i]);
        fp_add(acc, acc, c);
        fp_norm1(acc, acc);
    }
    r = acc;
}

// ECP_map2point: u -> a point of E in projective coordinates (not yet in the r-torsion)
C12381_HDN void g1_map_to_point(g1p& out, const fp& u) {
    fp one, A, B, t, w, D, X2, X3, GX1, D2, ad, h, dinv, tu, Dn, wn, hn, Y, s;
    fp_one(one);
    fp_set_const(A, SSWU_A);
    fp_set_const(B, SSWU_B);
    const int sgn = fp_sign(u);
    fp_sqr(t, u);
    fp_mul_small(t, t, SSWU_Z);                       // t = Z u^2
    fp_add(w, t, one); fp_norm1(w, w);
    fp_mul(w, w, t);                                  // w = Z^2 u^4 + Z u^2
    fp_mul(D, A, w);                                  // denominator A' w
    fp_add(w, w, one); fp_norm1(w, w);
    fp_mul(w, w, B);
    fp_neg(X2, w); fp_norm1(X2, X2);                  // numerator of x1: -B'(w + 1)
    fp_mul(X3, t, X2);                                // numerator of x2 = Z u^2 x1
    // g(x1) * D^3 = X2^3 + A' D^2 X2 + B' D^3
    fp_sqr(GX1, X2);
    fp_sqr(D2, D);
    fp_mul(w, A, D2);
    fp_add(GX1, GX1, w); fp_norm1(GX1, GX1);
    fp_mul(GX1, GX1, X2);
    fp_mul(D2, D2, D);
    fp_mul(w, B, D2);
    fp_add(GX1, GX1, w); fp_norm1(GX1, GX1);
    fp_mul(ad, GX1, D);
    // one exponentiation: h = ad^((p-3)/4);  ad is a square iff h^2 ad == 1;  1/ad = h^4 ad
    fp_pow_fixed(h, ad, EXP_P_MINUS_3_DIV_4);
    fp_sqr(s, h);
    fp_mul(w, s, ad);
    fp_sub(w, w, one);
    const bool qr = fp_is_zero(w);
    fp_sqr(s, s);
    fp_mul(dinv, s, ad);                              // 1 / (g D^4)
    fp_mul(dinv, dinv, GX1);                          // 1 / D
    fp_mul(X2, X2, dinv);                             // x1
    fp_mul(X3, X3, dinv);                             // x2
    fp_mul(tu, t, u);                                 // Z u^3
    fp_sqr(D2, dinv);
    fp_mul(Dn, D2, tu);
    fp_mul_small(wn, ad, SSWU_Z);
    fp_set_const(s, SSWU_HINT_Z);
    fp_mul(hn, s, h);                                 // (Z ad)^((p-3)/4)
    fp x, sc, ww, hh;
    fp_select(x, qr, X2, X3);
    fp_select(sc, qr, D2, Dn);
    fp_select(ww, qr, ad, wn);
    fp_select(hh, qr, h, hn);
    fp_mul(Y, hh, ww);                                // square root of ww
    fp_mul(Y, Y, sc);
    fp_neg(s, Y); fp_norm1(s, s);
    fp_select(Y, fp_sign(Y) != sgn, s, Y);            // sign of y follows the sign of u
    // 11-isogeny E' -> E
    fp xnum, xden, ynum, yden;
    iso_eval(xnum, ISO11_XNUM, x, false);
    iso_eval(xden, ISO11_XDEN, x, true);
    iso_eval(ynum, ISO11_YNUM, x, false);
    iso_eval(yden, ISO11_YDEN, x, true);
    fp_mul(ynum, ynum, Y);
    fp_mul(out.x, xnum, yden);
    fp_mul(out.y, ynum, xden);
    fp_mul(out.z, xden, yden);
}

// ECP_cfp: P <- [1 - x]P = [|x| + 1]P, plain double-and-add over the 64-bit public constant
C12381_HDN void g1_clear_cofactor(g1p& p) {
    g1p base, acc;
    g1_norm1(base, p);
    acc = base;
#pragma unroll 1
    for (int i = 62; i >= 0; --i) {
        g1_dbl(acc);
        if ((G1_COFACTOR_W[i >> 5] >> (i & 31)) & 1u) g1_add(acc, base);
    }
    p = acc;
}

C12381_HD void g1_from_digest(g1p& out, const uint32_t* raw16) {
    fp u;
    fp_from_digest64(u, raw16);
    g1_map_to_point(out, u);
    g1_clear_cofactor(out);
}

}  // namespace c12381
