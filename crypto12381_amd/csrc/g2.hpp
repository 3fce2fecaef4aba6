// G2 arithmetic on the M-type sextic twist y^2 = x^3 + 4(1+i) over Fp2
// (replaces ECP2_dbl / ECP2_add / ECP2_affine / ECP2_toOctet ecp2_BLS12381.cpp:358-502, 109-133,
//  184-220 and PAIR_G2mul pair_BLS12381.cpp:927-983 of the reference).
// Same design as g1.hpp: one point per lane, complete projective formulas, no divergent branch.
// Bound contract: stored coordinates carry limb bound <= 2^29; fp2_mul needs LBx*LBy <= 2^58.
#pragma once
#include "fp2.hpp"
#include "g1.hpp"

namespace c12381 {

struct g2p { fp2 x, y, z; };     // (X:Y:Z), infinity = (0:1:0)

C12381_HD void g2_set_inf(g2p& p) { fp2_zero(p.x); fp2_one(p.y); fp2_zero(p.z); }
C12381_HD void g2_norm1(g2p& r, const g2p& p) { fp2_norm1(r.x, p.x); fp2_norm1(r.y, p.y); fp2_norm1(r.z, p.z); }
C12381_HD void g2_neg(g2p& r, const g2p& p) { r.x = p.x; fp2_neg(r.y, p.y); r.z = p.z; }
// 3b' = 12 (1 + i): small multiply (normalising) then the lazy (1+i) map — ecp2_BLS12381.cpp:381-385
C12381_HD void fp2_mul_b3(fp2& r, const fp2& x) { fp2 t; fp2_mul_small(t, x, 12); fp2_mul_ip(r, t); }

// ECP2_dbl :358-409.  Also returns t0 = Y^2, t1 = Y*Z, t2b = 3b' Z^2 for the Miller-loop line.
C12381_HDN void g2_dbl_ex(g2p& p, fp2& t0, fp2& t1, fp2& t2b) {
    fp2 t2, x3, y3, z3, u;
    fp2_sqr(t0, p.y);
    fp2_mul(t1, p.y, p.z);
    fp2_sqr(t2, p.z);
    fp2_mul_small(z3, t0, 8);
    fp2_mul_b3(t2b, t2);
    fp2_mul(x3, t2b, z3);
    fp2_add(y3, t0, t2b);
    fp2_mul(z3, t1, z3);
    fp2_dbl(u, t2b); fp2_add(u, u, t2b);          // 9b' Z^2
    fp2_sub(u, t0, u);
    fp2_norm1(u, u);
    fp2_mul(y3, u, y3);
    fp2_add(y3, y3, x3);
    fp2 xy;
    fp2_mul(xy, p.x, p.y);
    fp2_mul(x3, u, xy);
    fp2_dbl(x3, x3);
    p.x = x3; p.y = y3; p.z = z3;
}
C12381_HD void g2_dbl(g2p& p) { fp2 a, b, c; g2_dbl_ex(p, a, b, c); }

// ECP2_add :413-502 (complete).  P limb bound <= 2^29, Q normalised.
C12381_HDN void g2_add(g2p& p, const g2p& q) {
    fp2 t0, t1, t2, t3, t4, x3, y3, z3;
    fp2_mul(t0, p.x, q.x);
    fp2_mul(t1, p.y, q.y);
    fp2_mul(t2, p.z, q.z);
    fp2_add(t3, p.x, p.y); fp2_norm1(t3, t3); fp2_add(t4, q.x, q.y); fp2_mul(t3, t3, t4);
    fp2_add(t4, t0, t1); fp2_sub(t3, t3, t4); fp2_norm1(t3, t3);
    fp2_add(t4, p.y, p.z); fp2_norm1(t4, t4); fp2_add(x3, q.y, q.z); fp2_mul(t4, t4, x3);
    fp2_add(x3, t1, t2); fp2_sub(t4, t4, x3); fp2_norm1(t4, t4);
    fp2_add(x3, p.x, p.z); fp2_norm1(x3, x3); fp2_add(y3, q.x, q.z); fp2_mul(x3, x3, y3);
    fp2_add(y3, t0, t2); fp2_sub(y3, x3, y3);
    fp2_mul_small(t0, t0, 3);
    fp2_mul_b3(t2, t2);
    fp2_add(z3, t1, t2);
    fp2_sub(t1, t1, t2); fp2_norm1(t1, t1);
    fp2_mul_b3(y3, y3);
    fp2_mul(x3, y3, t4); fp2_mul(t2, t3, t1); fp2_sub(p.x, t2, x3);
    fp2_mul(y3, y3, t0); fp2_mul(t1, t1, z3); fp2_add(p.y, y3, t1);
    fp2_mul(t0, t0, t3); fp2_mul(z3, z3, t4); fp2_add(p.z, z3, t0);
}

// ------------------------------------------------------------------ SoA access
C12381_HD void soa_store_fp2(int32_t* base, size_t stride, size_t idx, const fp2& a) {
    soa_store_fp(base, stride, idx, a.a);
    soa_store_fp(base + (size_t)NL * stride, stride, idx, a.b);
}
C12381_HD void soa_load_fp2(fp2& a, const int32_t* base, size_t stride, size_t idx) {
    soa_load_fp(a.a, base, stride, idx);
    soa_load_fp(a.b, base + (size_t)NL * stride, stride, idx);
}
C12381_HD void soa_store_g2(int32_t* base, size_t stride, size_t idx, const g2p& p) {
    soa_store_fp2(base, stride, idx, p.x);
    soa_store_fp2(base + (size_t)2 * NL * stride, stride, idx, p.y);
    soa_store_fp2(base + (size_t)4 * NL * stride, stride, idx, p.z);
}
C12381_HD void soa_load_g2(g2p& p, const int32_t* base, size_t stride, size_t idx) {
    soa_load_fp2(p.x, base, stride, idx);
    soa_load_fp2(p.y, base + (size_t)2 * NL * stride, stride, idx);
    soa_load_fp2(p.z, base + (size_t)4 * NL * stride, stride, idx);
}

constexpr int G2_WIN = 4;
constexpr int G2_TAB = 1 << G2_WIN;
constexpr int G2_TAB_DWORDS = G2_TAB * 6 * NL;

// [k]Q, Q affine or infinity; 4-bit fixed windows over k mod r (255 bits), table of 16 multiples in HBM.
// (The reference uses a 4-dimensional GS decomposition, pair_BLS12381.cpp:814-873 + ECP2_mul4; only
// the resulting group element is observable.)
C12381_HDN void g2_scalar_mul(g2p& acc, const fp2& qx, const fp2& qy, bool q_is_inf, const uint32_t (&kin)[8],
                             int32_t* tab, size_t stride, size_t lane) {
    uint32_t k[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) k[i] = kin[i];
    scalar_mod_r(k);
    g2p base, t;
    g2_set_inf(t);
    base.x = qx; base.y = qy; fp2_one(base.z);
    fp2_select(base.x, q_is_inf, t.x, base.x);
    fp2_select(base.y, q_is_inf, t.y, base.y);
    fp2_select(base.z, q_is_inf, t.z, base.z);
    const size_t ent = (size_t)6 * NL * stride;
    soa_store_g2(tab, stride, lane, t);
    soa_store_g2(tab + ent, stride, lane, base);
    t = base;
    g2_dbl(t);
    {
        g2p n;
        g2_norm1(n, t);
        soa_store_g2(tab + 2 * ent, stride, lane, n);
        t = n;
    }
#pragma unroll 1
    for (int j = 3; j < G2_TAB; ++j) {
        g2_add(t, base);
        g2p n;
        g2_norm1(n, t);
        soa_store_g2(tab + (size_t)j * ent, stride, lane, n);
        t = n;
    }
    g2_set_inf(acc);
#pragma unroll 1
    for (int w = 256 / G2_WIN - 1; w >= 0; --w) {
        g2_dbl(acc); g2_dbl(acc); g2_dbl(acc); g2_dbl(acc);
        const uint32_t d = (k[w >> 3] >> ((w & 7) * 4)) & 15u;
        g2p q;
        soa_load_g2(q, tab + (size_t)d * ent, stride, lane);
        g2_add(acc, q);
    }
}

// ECP2_setx ecp2_BLS12381.cpp:322-344: y = sqrt(x^3 + 4(1+i)) with FP2_sign(y) == s.
C12381_HD bool g2_set_x(fp2& y, const fp2& x, int s) {
    fp2 x2, x3, b, rhs, r, nr;
    fp2_sqr(x2, x); fp2_mul(x3, x2, x);
    fp_set_const(b.a, FP_FOUR); fp_set_const(b.b, FP_FOUR);
    fp2_add(rhs, x3, b);
    const bool ok = fp2_sqrt(r, rhs);
    fp2_neg(nr, r);
    fp2_select(y, fp2_sign(r) != s, nr, r);
    return ok;
}

}  // namespace c12381
