// G1 arithmetic and the per-lane scalar multiplication of the batched path
// (replaces ECP_dbl / ECP_add ecp_BLS12381.cpp:550-588, 750-812, ECP_affine :329,
//  ECP_toOctet :445-488 and PAIR_G1mul + glv pair_BLS12381.cpp:759-810, 876-924 of the
//  reference's vendored MIRACL-core).
//
// Design for CDNA4: one (point, scalar) per lane, homogeneous projective coordinates with
// the Renes–Costello–Batina COMPLETE formulas for a = 0 — no exceptional cases, hence no
// data-dependent branch and no wavefront divergence (P+P, P+(-P), infinity operands and
// zero digits all go through the same instruction stream).  GLV: k = k0 + k1*x^2 with
// [x^2](x,y) = (beta*x, -y), so both 128-bit halves share ONE table of 8 multiples of P (signed
// 4-bit windows) that lives in HBM as one contiguous 1408-byte record per lane.
#pragma once
#include "fp.hpp"

namespace c12381 {

struct g1p { fp x, y, z; };      // (X:Y:Z), infinity = (0:1:0)

C12381_HD void g1_set_inf(g1p& p) { fp_zero(p.x); fp_one(p.y); fp_zero(p.z); }
C12381_HD bool g1_is_inf(const g1p& p) { return fp_is_zero(p.z); }
C12381_HD void g1_norm1(g1p& r, const g1p& p) { fp_norm1(r.x, p.x); fp_norm1(r.y, p.y); fp_norm1(r.z, p.z); }

// P = 2P.  6M + 2S with 7 reductions (Y3 is a lazily reduced sum of two products).
// Operand limb bound: <= 2^29.
// Round 4: u = Y^2 - 9b Z^2 leaves the reduction of Y^2 with -3 (3b Z^2) injected (fp_reduce_cols_inj) — normalised, no lazy
// sum and no carry round; Y^2 itself is u + 3 (3b Z^2), used lazily (the lazy form of both formulas: profiles/r04_ab_injection_switches.txt).
C12381_HD void g1_dbl(g1p& p) {
    fp t0, t1, t2, z8, u, y3, x3, z3;
    const int32_t cm3 = fp_opaque_const(-3);
    fp_mul(t1, p.y, p.z);
    fp_sqr(t2, p.z);
    fp_mul_small(t2, t2, 12);                            // 3b Z^2
    fp_sqr_inj(u, p.y, [&](int i, int64_t& acc) { fp_inj(acc, t2, i, cm3); }, C12381_BV(3 * t2.vb), C12381_BV(3 * t2.lb));          // Y^2 - 9b Z^2
    fp_dbl(t0, t2); fp_add(t0, t0, t2); fp_add(t0, t0, u);                                                    // Y^2 = u + 9b Z^2 (limbs < 2^30)
    fp_mul_small(z8, t0, 8);                             // 8 Y^2
    fp_add(y3, t0, t2);
    fp_mul(z3, t1, z8);
    fp_mul2<false>(y3, u, y3, t2, z8);                   // (Y^2 - 9bZ^2)(Y^2 + 3bZ^2) + 3bZ^2 * 8Y^2
    fp_mul(t1, p.x, p.y);
    fp_mul(x3, u, t1);
    fp_dbl(x3, x3);
    p.x = x3; p.y = y3; p.z = z3;
}

// P = P + Q (complete).  12 products, 9 reductions.
// Operand limb bounds: P <= 2^29, Q <= 2^28 (+ carry slack): table entries are stored normalised.
C12381_HD void g1_add(g1p& p, const g1p& q) {
    fp t0, t1, t2, t3, t4, x3, y3, z3;
    fp_mul(t0, p.x, q.x);
    fp_mul(t1, p.y, q.y);
    fp_mul(t2, p.z, q.z);
    {   // the Karatsuba corrections -t0 - t1, -t1 - t2 ride in the reductions: t3, t4 normalised without a lazy sum or a carry round
        const int32_t cm1 = fp_opaque_const(-1);
        fp sa, sb;
        fp_add(sa, p.x, p.y); fp_add(sb, q.x, q.y);
        fp_mul_inj(t3, sa, sb, [&](int i, int64_t& acc) { fp_inj(acc, t0, i, cm1); fp_inj(acc, t1, i, cm1); }, C12381_BV(t0.vb + t1.vb), C12381_BV(t0.lb + t1.lb));
        fp_add(sa, p.y, p.z); fp_add(sb, q.y, q.z);
        fp_mul_inj(t4, sa, sb, [&](int i, int64_t& acc) { fp_inj(acc, t1, i, cm1); fp_inj(acc, t2, i, cm1); }, C12381_BV(t1.vb + t2.vb), C12381_BV(t1.lb + t2.lb));
    }
    fp_add(x3, p.x, p.z); fp_add(y3, q.x, q.z); fp_mul(x3, x3, y3);
    fp_add(y3, t0, t2); fp_sub(y3, x3, y3);
    fp_mul_small(t0, t0, 3);
    fp_mul_small(t2, t2, 12);
    fp_add(z3, t1, t2); fp_sub(t1, t1, t2);
    fp_mul_small(y3, y3, 12);
    fp_mul2<true>(p.x, t3, t1, y3, t4);                  // X3 = t3*t1 - y3*t4
    fp_mul2<false>(p.y, y3, t0, t1, z3);                 // Y3 = y3*t0 + t1*z3
    fp_mul2<false>(p.z, z3, t4, t0, t3);                 // Z3 = z3*t4 + t0*t3
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
// k < r  ->  k = k0 + k1 * x^2 with 0 <= k0 < x^2 < 2^128 and k1 < 2^128  (replaces glv() pair_BLS12381.cpp:793-805,
// whose u1 = r - (e div x^2) belongs to the opposite sign convention of the endomorphism).
// Barrett division by the 128-bit constant x^2 (top bit set): q^ = floor(floor(k / 2^127) * mu / 2^129) with
// mu = floor(2^256 / x^2) satisfies q - 2 <= q^ <= q for k < 2^256, so two conditional subtractions finish it:
// ~60 multiply-adds instead of a 256-step restoring division (which was more than half of the MSM preparation kernel).
C12381_HD void scalar_glv_split(uint32_t (&k0)[4], uint32_t (&k1)[4], const uint32_t (&k)[8]) {
    uint32_t a[5], prod[10];
#pragma unroll
    for (int i = 0; i < 5; ++i) a[i] = (k[i + 3] >> 31) | (i + 4 < 8 ? k[i + 4] << 1 : 0u);
#pragma unroll
    for (int i = 0; i < 10; ++i) prod[i] = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        uint64_t carry = 0;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const uint64_t t = (uint64_t)a[i] * GLV_MU[j] + prod[i + j] + carry;
            prod[i + j] = (uint32_t)t;
            carry = t >> 32;
        }
        prod[i + 5] = (uint32_t)carry;
    }
    uint32_t q[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = (prod[4 + i] >> 1) | (prod[5 + i] << 31);
    // rem = k - q * x^2, exact in 160 bits (0 <= rem < 3 x^2)
    uint32_t qd[5] = {0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint64_t carry = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (i + j < 5) {
                const uint64_t t = (uint64_t)q[i] * GLV_X2[j] + qd[i + j] + carry;
                qd[i + j] = (uint32_t)t;
                carry = t >> 32;
            }
        }
        if (i + 4 < 5) qd[i + 4] = (uint32_t)carry;
    }
    uint32_t rem[5];
    {
        uint64_t bw = 0;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const uint64_t t = (uint64_t)k[i] - qd[i] - bw;
            rem[i] = (uint32_t)t;
            bw = (t >> 32) & 1;
        }
    }
#pragma unroll
    for (int round = 0; round < 2; ++round) {
        uint32_t d[5];
        uint64_t bw = 0;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const uint64_t t = (uint64_t)rem[i] - (i < 4 ? GLV_X2[i] : 0u) - bw;
            d[i] = (uint32_t)t;
            bw = (t >> 32) & 1;
        }
        const bool ge = bw == 0;
#pragma unroll
        for (int i = 0; i < 5; ++i) rem[i] = ge ? d[i] : rem[i];
        uint64_t c = ge ? 1u : 0u;
#pragma unroll
        for (int i = 0; i < 4; ++i) { c += q[i]; q[i] = (uint32_t)c; c >>= 32; }
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

// ------------------------------------------------------------------ per-lane window table
// Signed 5-bit windows: entries 1..16 of multiples of P (4-bit: 1..8), each entry X|Y|Z = 42 dwords padded to 44 (176 B,
// eleven 16-byte accesses).  A lane's whole table is one contiguous 2816-byte record, so a gather of one
// entry touches 176 consecutive bytes of HBM instead of 42 scattered dwords (the limb-major layout of the
// first version moved ~16x the algorithmic bytes: profiles/r01_pmc_summary_before_table_fix.txt).
// Per scalar multiplication: 8 + 125 doublings and 7 + 52 additions (4-bit windows: 4 + 128 and 3 + 66).
constexpr int G1_WIN = 5;                               // 4 or 5; A/B on MI355X (profiles/r02_ab_g1_window5.txt): 5 is 3.5 % faster
static_assert(G1_WIN == 4 || G1_WIN == 5, "window width");
constexpr int G1_TAB = 1 << (G1_WIN - 1);              // entries 1..8 (1..16)
constexpr int G1_WINDOWS = G1_WIN == 4 ? 33 : 26;      // 4: 32 biased nibbles + the carry nibble; 5: 26 biased fields cover 130 bits
constexpr int G1_ENT_DWORDS = 44;
constexpr int G1_TAB_DWORDS = G1_TAB * G1_ENT_DWORDS;  // 352 dwords = 1408 B per lane (5-bit windows: 2816 B)
struct alignas(16) q4 { int32_t v[4]; };

C12381_HD void tab_store_g1(int32_t* ent, const g1p& p) {
    int32_t w[G1_ENT_DWORDS];
#pragma unroll
    for (int i = 0; i < NL; ++i) { w[i] = p.x.l[i]; w[NL + i] = p.y.l[i]; w[2 * NL + i] = p.z.l[i]; }
    w[42] = 0; w[43] = 0;
    q4* dst = reinterpret_cast<q4*>(ent);
#pragma unroll
    for (int i = 0; i < G1_ENT_DWORDS / 4; ++i) { q4 t; t.v[0] = w[4 * i]; t.v[1] = w[4 * i + 1]; t.v[2] = w[4 * i + 2]; t.v[3] = w[4 * i + 3]; dst[i] = t; }
}
C12381_HD void tab_load_g1(g1p& p, const int32_t* ent) {
    int32_t w[G1_ENT_DWORDS];
    const q4* src = reinterpret_cast<const q4*>(ent);
#pragma unroll
    for (int i = 0; i < G1_ENT_DWORDS / 4; ++i) { q4 t = src[i]; w[4 * i] = t.v[0]; w[4 * i + 1] = t.v[1]; w[4 * i + 2] = t.v[2]; w[4 * i + 3] = t.v[3]; }
#pragma unroll
    for (int i = 0; i < NL; ++i) { p.x.l[i] = w[i]; p.y.l[i] = w[NL + i]; p.z.l[i] = w[2 * NL + i]; }
    C12381_BOUNDS(p.x.lb = p.y.lb = p.z.lb = 268435456.0 + 8.0; p.x.vb = p.y.vb = p.z.vb = 4.0;
                  check_actual(p.x, "tab_load_g1"); check_actual(p.y, "tab_load_g1"); check_actual(p.z, "tab_load_g1");)
}
// signed digit of window w of k' = k + 0x888...8 (32 nibbles): d = nibble - 8 in [-8, 7]; window 32 is the
// carry nibble (0 or 1, no bias).  Sum_w d_w 16^w = k.
// 5-bit windows: k' = k + sum_w 16 * 32^w (w < 26; k < 2^128, so k' < 2^130 and there is no carry window):
// d = field - 16 in [-16, 15].
constexpr uint32_t glv_bias_word5(int i) {
    uint32_t v = 0;
    for (int w = 0; w < 26; ++w) { const int bit = 5 * w + 4; if ((bit >> 5) == i) v |= 1u << (bit & 31); }
    return v;
}
C12381_HD int glv_digit(const uint32_t (&kb)[5], int w) {
    if (G1_WIN == 4) {
        const int nib = (int)((kb[w >> 3] >> ((w & 7) * 4)) & 15u);
        return w == 32 ? nib : nib - 8;
    }
    const int bit = 5 * w, word = bit >> 5, sh = bit & 31;
    uint32_t v = kb[word] >> sh;
    if (sh > 27) v |= kb[word + 1] << (32 - sh);
    return (int)(v & 31u) - 16;
}
C12381_HD void glv_bias(uint32_t (&kb)[5], const uint32_t (&k)[4]) {
    uint64_t c = 0;
    if (G1_WIN == 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { c += (uint64_t)k[i] + 0x88888888u; kb[i] = (uint32_t)c; c >>= 32; }
        kb[4] = (uint32_t)c;
    } else {
        constexpr uint32_t B[5] = {glv_bias_word5(0), glv_bias_word5(1), glv_bias_word5(2), glv_bias_word5(3), glv_bias_word5(4)};
#pragma unroll
        for (int i = 0; i < 4; ++i) { c += (uint64_t)k[i] + B[i]; kb[i] = (uint32_t)c; c >>= 32; }
        kb[4] = (uint32_t)c + B[4];
    }
}
// the table record a digit selects (|d| = 0 reads entry 1 and is replaced by the point at infinity afterwards)
C12381_HD const int32_t* g1_digit_entry(const int32_t* lane_tab, int d) {
    const int mag = d < 0 ? -d : d;
    return lane_tab + ((mag == 0 ? 1 : mag) - 1) * G1_ENT_DWORDS;
}
// q = sign(d) * T[|d|], or its image under the endomorphism; d == 0 gives the point at infinity (same instruction stream)
C12381_HD void g1_digit_fix(g1p& r, g1p q, int d, bool endo);
C12381_HD void g1_digit_point(g1p& r, const int32_t* lane_tab, int d, bool endo) {
    g1p q;
    tab_load_g1(q, g1_digit_entry(lane_tab, d));
    g1_digit_fix(r, q, d, endo);
}
C12381_HD void g1_digit_fix(g1p& r, g1p q, int d, bool endo) {
    const int mag = d < 0 ? -d : d;
    fp ny, zero, one;
    fp_neg(ny, q.y);
    fp_select(q.y, d < 0, ny, q.y);
    fp_zero(zero); fp_one(one);
    const bool isz = mag == 0;
    fp_select(q.x, isz, zero, q.x); fp_select(q.y, isz, one, q.y); fp_select(q.z, isz, zero, q.z);
    C12381_BOUNDS(q.x.lb = q.y.lb = q.z.lb = 268435456.0 + 8.0;)
    if (endo) g1_endo_x2(r, q); else r = q;          // compile-time constant at every call site
}
// acc += sign(d) * T[|d|]
C12381_HD void g1_add_digit(g1p& acc, const int32_t* lane_tab, int d, bool endo) {
    g1p e;
    g1_digit_point(e, lane_tab, d, endo);
    g1_add(acc, e);
}

// p <- [|x|]p, plain double-and-add over the 64-bit curve parameter (weight 6)
C12381_HDN void g1_mul_absx(g1p& p) {
    g1p base, acc;
    g1_norm1(base, p);
    acc = base;
#pragma unroll 1
    for (int i = 62; i >= 0; --i) {
        g1_dbl(acc);
        if ((BLS_X_W[i >> 5] >> (i & 31)) & 1u) g1_add(acc, base);
    }
    p = acc;
}
// The reference's glv() (pair_BLS12381.cpp:793-805) returns u1 = r - (k div x^2), and PAIR_G1mul's sign minimisation
// (:899-906) maps that back to k div x^2 — except when k div x^2 == 0: then u1 stays r and ECP_mul2 adds [r]phi(P),
// phi(P) = (beta x, y).  That term is the point at infinity for P in G1 and a point of the cofactor part otherwise.
// Reproduced (lanes with k < x^2 only; see scalar_below_x2) so results agree with `multiply` on every curve point.
// phi(P) is in G1 iff phi(phi(P)) = [-x^2]phi(P); off the subgroup [r]Q = [x^2]([x^2]Q) - [x^2]Q + Q  (r = x^4 - x^2 + 1).
C12381_HDN void g1_glv_small_scalar_term(g1p& acc, const g1p& base) {
    fp beta;
    fp_set_const(beta, FP_BETA_A);
    g1p q, s1, t;
    fp_mul(q.x, base.x, beta); q.y = base.y; q.z = base.z;            // phi(P)
    g1_norm1(q, q);
    s1 = q;
    g1_mul_absx(s1); g1_mul_absx(s1);                                  // [x^2]phi(P)
    g1_norm1(s1, s1);
    fp_mul(t.x, q.x, beta); t.y = q.y; t.z = q.z;                      // phi(phi(P))
    g1_norm1(t, t);
    {
        g1p u = s1;
        g1_add(u, t);
        if (g1_is_inf(u)) return;                                      // P in G1: the extra term vanishes
    }
    g1p s2 = s1;
    g1_mul_absx(s2); g1_mul_absx(s2);                                  // [x^4]phi(P)
    fp_neg(t.y, s1.y); t.x = s1.x; t.z = s1.z;
    g1_norm1(t, t);
    g1_add(s2, t);
    g1_norm1(s2, s2);
    g1_add(s2, q);
    g1_norm1(s2, s2);
    g1_norm1(acc, acc);
    g1_add(acc, s2);
}

// [k]P for an AFFINE input point (x, y) or infinity.  `lane_tab` = this lane's table record (G1_TAB_DWORDS).
C12381_HD void g1_scalar_mul(g1p& acc, const fp& px, const fp& py, bool p_is_inf, const uint32_t (&kin)[8], int32_t* lane_tab) {
    uint32_t k[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) k[i] = kin[i];
    scalar_mod_r(k);
    uint32_t k0[4], k1[4], kb0[5], kb1[5];
    scalar_glv_split(k0, k1, k);
    glv_bias(kb0, k0);
    glv_bias(kb1, k1);

    // table T[j] = j*P, j = 1..G1_TAB, stored normalised (limb bound 2^28 + slack)
    g1p base, t;
    base.x = px; base.y = py; fp_one(base.z);
    {   // infinity input: use (0:1:0) as the base so every multiple is infinity
        g1p inf;
        g1_set_inf(inf);
        fp_select(base.x, p_is_inf, inf.x, base.x);
        fp_select(base.y, p_is_inf, inf.y, base.y);
        fp_select(base.z, p_is_inf, inf.z, base.z);
    }
    tab_store_g1(lane_tab, base);                               // T[1]
    t = base;
    g1_dbl(t);
    {
        g1p n;
        g1_norm1(n, t);
        tab_store_g1(lane_tab + G1_ENT_DWORDS, n);              // T[2]
        t = n;
    }
    // even multiples by doubling the entry half as large (8 products + 7 reductions against 12 + 9 for an addition; the entry comes
    // back from the lane's own record), odd ones by adding P to the previous entry: 8 doublings + 7 additions for 16 entries
#pragma unroll 1
    for (int j = 3; j <= G1_TAB; ++j) {
        if ((j & 1) == 0) {                                     // wave-uniform
            tab_load_g1(t, lane_tab + (j / 2 - 1) * G1_ENT_DWORDS);
            g1_dbl(t);
        } else {
            g1_add(t, base);
        }
        g1p n;
        g1_norm1(n, t);
        tab_store_g1(lane_tab + (j - 1) * G1_ENT_DWORDS, n);
        t = n;
    }

    // the top window starts the accumulator with its first digit's entry (an addition to the point at infinity would compute the same
    // point): 51 + 7 additions in all
    g1_digit_point(acc, lane_tab, glv_digit(kb0, G1_WINDOWS - 1), false);
    {
        g1p n;
        g1_norm1(n, acc);
        acc = n;
    }
    g1_add_digit(acc, lane_tab, glv_digit(kb1, G1_WINDOWS - 1), true);
    // (round 4, profiles/r04_ab_g1_prefetch.txt) the record of an addition is requested one operation ahead — the first digit's before the window's
    // doublings (44 registers across them), the second digit's before the first addition — instead of at the head of the addition that needs
    // it, where the whole latency of the gather (a 176-byte record somewhere in a slab of gigabytes) was exposed twice per window.
#pragma unroll 1
    for (int w = G1_WINDOWS - 2; w >= 0; --w) {
#if defined(__HIP_DEVICE_COMPILE__)
        const int d0 = glv_digit(kb0, w), d1 = glv_digit(kb1, w);
        g1p q0, q1, e;
        tab_load_g1(q0, g1_digit_entry(lane_tab, d0));
        __builtin_amdgcn_sched_barrier(0);               // the loads stay in front of the doublings
        g1_dbl(acc); g1_dbl(acc); g1_dbl(acc); g1_dbl(acc);
        if (G1_WIN == 5) g1_dbl(acc);
        tab_load_g1(q1, g1_digit_entry(lane_tab, d1));
        __builtin_amdgcn_sched_barrier(0);
        g1_digit_fix(e, q0, d0, false);
        g1_add(acc, e);
        g1_digit_fix(e, q1, d1, true);
        g1_add(acc, e);
#else
        g1_dbl(acc); g1_dbl(acc); g1_dbl(acc); g1_dbl(acc);
        if (G1_WIN == 5) g1_dbl(acc);
        g1_add_digit(acc, lane_tab, glv_digit(kb0, w), false);
        g1_add_digit(acc, lane_tab, glv_digit(kb1, w), true);
#endif
    }
}
// k mod r < x^2, i.e. k div x^2 == 0: the lanes that owe g1_glv_small_scalar_term (evaluated by a separate, almost always
// empty fix-up kernel so that the main loop's register allocation does not pay for the rare branch)
C12381_HD bool scalar_below_x2(const uint32_t (&kin)[8]) {
    uint32_t k[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) k[i] = kin[i];
    scalar_mod_r(k);
    uint64_t bw = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) { uint64_t t = (uint64_t)k[i] - GLV_X2[i] - bw; bw = (t >> 32) & 1; }
    return (k[4] | k[5] | k[6] | k[7]) == 0u && bw == 1;
}

// ------------------------------------------------------------------ affine output
// x = X/Z, y = Y/Z given zinv = 1/Z; compressed tag 0x02|parity(y)  (ECP_toOctet :445-488)
C12381_HD void g1_to_affine(fp& ax, fp& ay, const g1p& p, const fp& zinv) {
    fp_mul(ax, p.x, zinv);
    fp_mul(ay, p.y, zinv);
}

// ------------------------------------------------------------------ compressed-point decoding
// ECP_setx ecp_BLS12381.cpp:302-323: y = sqrt(x^3 + 4) with parity s; fails when the right-hand side is
// not a residue (0 counts as a non-residue: FP_qr).  x is taken mod p (FP_nres), no subgroup check.
C12381_HD bool g1_set_x(fp& y, const fp& x, int s) {
    fp x2, x3, four, rhs, c, cinv, ny;
    fp_sqr(x2, x); fp_mul(x3, x2, x);
    fp_set_const(four, FP_FOUR);
    fp_add(rhs, x3, four);
    const bool qr = fp_sqrt_progen(c, cinv, rhs);
    fp_neg(ny, c);
    fp_select(y, fp_sign(c) != s, ny, c);
    return qr;
}

}  // namespace c12381
