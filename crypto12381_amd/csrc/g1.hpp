// G1 arithmetic and the per-lane scalar multiplication of the batched path
// (replaces ECP_dbl / ECP_add ecp_BLS12381.cpp:550-588, 750-812, ECP_affine :329,
//  ECP_toOctet :445-488 and PAIR_G1mul + glv pair_BLS12381.cpp:759-810, 876-924 of the
//  reference's vendored MIRACL-core).
//
// Design for CDNA4: one (point, scalar) per lane, homogeneous projective coordinates with
// the Renes–Costello–Batina COMPLETE formulas for a = 0 — no exceptional cases, hence no
// data-dependent branch and no wavefront divergence (P+P, P+(-P), infinity operands and
// zero digits all go through the same instruction stream).  GLV: k = k0 + k1*x^2 with
// [x^2](x,y) = (beta*x, -y), so both 128-bit halves share ONE 16-entry window table that
// lives in HBM in limb-major SoA order (coalesced when written, per-lane gathered when read).
#pragma once
#include "fp.hpp"

namespace c12381 {

struct g1p { fp x, y, z; };      // (X:Y:Z), infinity = (0:1:0)

C12381_HD void g1_set_inf(g1p& p) { fp_zero(p.x); fp_one(p.y); fp_zero(p.z); }
C12381_HD bool g1_is_inf(const g1p& p) { return fp_is_zero(p.z); }
C12381_HD void g1_norm1(g1p& r, const g1p& p) { fp_norm1(r.x, p.x); fp_norm1(r.y, p.y); fp_norm1(r.z, p.z); }

// P = 2P.  6M + 2S + one small-constant multiply.  Operand limb bound: <= 2^29.
C12381_HD void g1_dbl(g1p& p) {
    fp t0, t1, t2, x3, y3, z3;
    fp_sqr(t0, p.y);
    fp_mul(t1, p.y, p.z);
    fp_sqr(t2, p.z);
    fp_dbl(z3, t0); fp_dbl(z3, z3); fp_dbl(z3, z3);      // 8 Y^2 (limbs < 2^31)
    fp_mul_small(t2, t2, 12);                            // 3b Z^2
    fp_mul(x3, t2, z3);
    fp_add(y3, t0, t2);
    fp_mul(z3, t1, z3);
    fp_dbl(t1, t2); fp_add(t2, t2, t1);                  // 9b Z^2
    fp_sub(t0, t0, t2);
    fp_mul(y3, t0, y3);
    fp_add(y3, y3, x3);
    fp_mul(t1, p.x, p.y);
    fp_mul(x3, t0, t1);
    fp_dbl(x3, x3);
    p.x = x3; p.y = y3; p.z = z3;
}

// P = P + Q (complete).  12M + three small-constant multiplies.
// Operand limb bounds: P <= 2^29, Q <= 2^28 (+ carry slack): table entries are stored normalised.
C12381_HD void g1_add(g1p& p, const g1p& q) {
    fp t0, t1, t2, t3, t4, x3, y3, z3;
    fp_mul(t0, p.x, q.x);
    fp_mul(t1, p.y, q.y);
    fp_mul(t2, p.z, q.z);
    fp_add(t3, p.x, p.y); fp_add(t4, q.x, q.y); fp_mul(t3, t3, t4);
    fp_add(t4, t0, t1); fp_sub(t3, t3, t4);
    fp_add(t4, p.y, p.z); fp_add(x3, q.y, q.z); fp_mul(t4, t4, x3);
    fp_add(x3, t1, t2); fp_sub(t4, t4, x3);
    fp_add(x3, p.x, p.z); fp_add(y3, q.x, q.z); fp_mul(x3, x3, y3);
    fp_add(y3, t0, t2); fp_sub(y3, x3, y3);
    fp_mul_small(t0, t0, 3);
    fp_mul_small(t2, t2, 12);
    fp_add(z3, t1, t2); fp_sub(t1, t1, t2);
    fp_mul_small(y3, y3, 12);
    fp_mul(x3, y3, t4); fp_mul(t2, t3, t1); fp_sub(p.x, t2, x3);
    fp_mul(y3, y3, t0); fp_mul(t1, t1, z3); fp_add(p.y, y3, t1);
    fp_mul(t0, t0, t3); fp_mul(z3, z3, t4); fp_add(p.z, z3, t0);
}

// (X:Y:Z) -> (beta*X : -Y : Z) = [x^2](X:Y:Z)
C12381_HD void g1_endo_x2(g1p& r, const g1p& p) {
    fp beta;
    fp_set_const(beta, FP_BETA_A);
    fp_mul(r.x, p.x, beta);
    fp_neg(r.y, p.y);
    r.z = p.z;
}

// ------------------------------------------------------------------ scalars
// k (8 little-endian 32-bit words, any value < 2^256) -> k mod r.  r > 2^254, so at most
// three subtractions (the reference reduces first too: pair_BLS12381.cpp:879-881).
C12381_HD void scalar_mod_r(uint32_t (&k)[8]) {
#pragma unroll 1
    for (int round = 0; round < 3; ++round) {
        uint32_t d[8];
        uint64_t bw = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            uint64_t t = (uint64_t)k[i] - ORDER_R[i] - bw;
            d[i] = (uint32_t)t;
            bw = (t >> 32) & 1;
        }
        const bool ge = bw == 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) k[i] = ge ? d[i] : k[i];
    }
}
// k < r  ->  k = k0 + k1 * x^2 with 0 <= k0 < x^2 < 2^128 and k1 < 2^128  (restoring division;
// replaces glv() pair_BLS12381.cpp:793-805, whose u1 = r - (e div x^2) belongs to the opposite sign
// convention of the endomorphism).
C12381_HD void scalar_glv_split(uint32_t (&k0)[4], uint32_t (&k1)[4], const uint32_t (&k)[8]) {
    uint32_t rem[5] = {0, 0, 0, 0, 0};
    uint32_t q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll 1
    for (int bit = 255; bit >= 0; --bit) {
        const uint32_t in = (k[bit >> 5] >> (bit & 31)) & 1u;
#pragma unroll
        for (int i = 4; i >= 1; --i) rem[i] = (rem[i] << 1) | (rem[i - 1] >> 31);
        rem[0] = (rem[0] << 1) | in;
        uint32_t d[5];
        uint64_t bw = 0;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            uint64_t t = (uint64_t)rem[i] - (i < 4 ? GLV_X2[i] : 0u) - bw;
            d[i] = (uint32_t)t;
            bw = (t >> 32) & 1;
        }
        const bool ge = bw == 0;
#pragma unroll
        for (int i = 0; i < 5; ++i) rem[i] = ge ? d[i] : rem[i];
        q[bit >> 5] |= (ge ? 1u : 0u) << (bit & 31);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) { k0[i] = rem[i]; k1[i] = q[i]; }
}

// ------------------------------------------------------------------ limb-major SoA access
// element `idx` of an array of fp stored as limb[NL][stride]
C12381_HD void soa_store_fp(int32_t* base, size_t stride, size_t idx, const fp& a) {
#pragma unroll
    for (int i = 0; i < NL; ++i) base[(size_t)i * stride + idx] = a.l[i];
}
C12381_HD void soa_load_fp(fp& a, const int32_t* base, size_t stride, size_t idx) {
#pragma unroll
    for (int i = 0; i < NL; ++i) a.l[i] = base[(size_t)i * stride + idx];
    C12381_BOUNDS(a.lb = 268435456.0 + 8.0; a.vb = 4.0; check_actual(a, "soa_load_fp");)
}
C12381_HD void soa_store_g1(int32_t* base, size_t stride, size_t idx, const g1p& p) {
    soa_store_fp(base, stride, idx, p.x);
    soa_store_fp(base + (size_t)NL * stride, stride, idx, p.y);
    soa_store_fp(base + (size_t)2 * NL * stride, stride, idx, p.z);
}
C12381_HD void soa_load_g1(g1p& p, const int32_t* base, size_t stride, size_t idx) {
    soa_load_fp(p.x, base, stride, idx);
    soa_load_fp(p.y, base + (size_t)NL * stride, stride, idx);
    soa_load_fp(p.z, base + (size_t)2 * NL * stride, stride, idx);
}

constexpr int G1_WIN = 4;                       // window width
constexpr int G1_TAB = 1 << G1_WIN;             // entries 0..15 (entry 0 = infinity)
constexpr int G1_TAB_DWORDS = G1_TAB * 3 * NL;  // per lane

// [k]P for an AFFINE input point (x, y) or infinity.  `tab` is this launch's table slab
// (G1_TAB_DWORDS x stride dwords), `lane` this thread's column in it.
C12381_HD void g1_scalar_mul(g1p& acc, const fp& px, const fp& py, bool p_is_inf, const uint32_t (&kin)[8],
                             int32_t* tab, size_t stride, size_t lane) {
    uint32_t k[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) k[i] = kin[i];
    scalar_mod_r(k);
    uint32_t k0[4], k1[4];
    scalar_glv_split(k0, k1, k);

    // window table T[j] = j*P, j = 0..15, stored normalised (limb bound 2^28 + slack)
    g1p base, t;
    g1_set_inf(t);
    base.x = px; base.y = py; fp_one(base.z);
    {   // infinity input: use (0:1:0) as the base so every multiple is infinity
        g1p inf;
        g1_set_inf(inf);
        fp_select(base.x, p_is_inf, inf.x, base.x);
        fp_select(base.y, p_is_inf, inf.y, base.y);
        fp_select(base.z, p_is_inf, inf.z, base.z);
    }
    const size_t ent = (size_t)3 * NL * stride;
    soa_store_g1(tab, stride, lane, t);                 // T[0]
    soa_store_g1(tab + ent, stride, lane, base);        // T[1]
    t = base;
    g1_dbl(t);
    {
        g1p n;
        g1_norm1(n, t);
        soa_store_g1(tab + 2 * ent, stride, lane, n);   // T[2]
        t = n;
    }
#pragma unroll 1
    for (int j = 3; j < G1_TAB; ++j) {
        g1_add(t, base);
        g1p n;
        g1_norm1(n, t);
        soa_store_g1(tab + (size_t)j * ent, stride, lane, n);
        t = n;
    }

    g1_set_inf(acc);
#pragma unroll 1
    for (int w = 128 / G1_WIN - 1; w >= 0; --w) {
        g1_dbl(acc); g1_dbl(acc); g1_dbl(acc); g1_dbl(acc);
        const uint32_t d0 = (k0[w >> 3] >> ((w & 7) * 4)) & 15u;
        const uint32_t d1 = (k1[w >> 3] >> ((w & 7) * 4)) & 15u;
        g1p q;
        soa_load_g1(q, tab + (size_t)d0 * ent, stride, lane);
        g1_add(acc, q);
        soa_load_g1(q, tab + (size_t)d1 * ent, stride, lane);
        g1p e;
        g1_endo_x2(e, q);
        g1_add(acc, e);
    }
}

// ------------------------------------------------------------------ affine output
// x = X/Z, y = Y/Z given zinv = 1/Z; compressed tag 0x02|parity(y)  (ECP_toOctet :445-488)
C12381_HD void g1_to_affine(fp& ax, fp& ay, const g1p& p, const fp& zinv) {
    fp_mul(ax, p.x, zinv);
    fp_mul(ay, p.y, zinv);
}

}  // namespace c12381
