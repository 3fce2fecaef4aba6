// Fixed-base scalar multiplication for the columns of a batch that share ONE public point (the public parameters of
// BBS+ verification, examples/bbs-plus/src/bbs+.cpp:57-73: g2^x, h0^r, h_i^{m_i}).  The reference multiplies every
// element generically (PAIR_G1mul / PAIR_G2mul); for a base IN the order-r subgroup the result is the same point,
// so a table of the base's multiples replaces all doublings:
//   G1: k mod r = k0 + k1 x^2 (the GLV split of g1.hpp), 16 byte-windows per half,
//       [k]B = sum_j T[j][k0_j] + sum_j endo(T[j][k1_j]),  T[j][d] = [d 2^(8j)]B affine, endo(x, y) = (beta x, -y)
//   G2: k mod r = u0 + u1|x| + u2|x|^2 + u3|x|^3 (the GS split of g2.hpp), 8 byte-windows per digit,
//       [k]Q = sum_i (-1)^i sum_j psi^i(T[j][u_i,j])
// 32 mixed additions and no doubling per element.  The tables (4080 x 112 B, 2040 x 224 B) are built on the device,
// kept in the context and rebuilt only when the base bytes change.  A base that is not a subgroup point (or is the
// point at infinity, or not on the curve) leaves the table unused: `ok` stays 0 and the generic kernels run instead,
// which reproduce the reference for every input.
#pragma once
#include "g2.hpp"
#include "msm.hpp"

namespace c12381 {

constexpr int FB_ENTRIES = 255;                       // digits 1..255 of an 8-bit window
constexpr int FB_G1_WINDOWS = 16, FB_G2_WINDOWS = 8;
constexpr int FB_G1_DWORDS = MSM_PT_DWORDS;           // affine (x, y) Montgomery, 112 B
constexpr int FB_G2_DWORDS = 4 * NL;                  // affine (x.a, x.b, y.a, y.b), 224 B
constexpr int FB_HEADER_DWORDS = 64;                  // cached base bytes (up to 192) + state words (kernels.hpp: HDR_*), in front of the table

// P in G1  <=>  phi(P) = [-x^2]P  <=>  [x^2]P + (beta X : Y : Z) = infinity   (kernel of phi - lambda has order r)
C12381_HDN bool g1_in_subgroup(const g1p& p) {
    fp beta;
    fp_set_const(beta, FP_BETA_A);
    g1p s, t;
    g1_norm1(s, p);
    g1_mul_absx(s); g1_mul_absx(s);
    g1_norm1(s, s);
    fp_mul(t.x, p.x, beta); t.y = p.y; t.z = p.z;
    g1_norm1(t, t);
    g1_add(s, t);
    return g1_is_inf(s);
}
// Q in G2  <=>  psi(Q) = [x]Q = -[|x|]Q
C12381_HDN bool g2_in_subgroup(const g2p& q) {
    g2p s, t, n;
    g2_norm1(n, q);
    s = n;
    g2_mul_absx(s);
    g2_norm1(s, s);
    g2_psi<1>(t, n);
    g2_norm1(t, t);
    g2_add(s, t);
    return g2_is_inf(s);
}

// [d 2^shift]B with complete formulas (d < 256): 8 ladder steps, then `shift` doublings
C12381_HDN void g1_fixed_entry(g1p& acc, const g1p& base, uint32_t d, int shift) {
    g1p b;
    g1_norm1(b, base);
    g1_set_inf(acc);
#pragma unroll 1
    for (int bit = 7; bit >= 0; --bit) {
        g1_dbl(acc);
        if ((d >> bit) & 1u) g1_add(acc, b);
    }
#pragma unroll 1
    for (int s = 0; s < shift; ++s) g1_dbl(acc);
}
C12381_HDN void g2_fixed_entry(g2p& acc, const g2p& base, uint32_t d, int shift) {
    g2p b;
    g2_norm1(b, base);
    g2_set_inf(acc);
#pragma unroll 1
    for (int bit = 7; bit >= 0; --bit) {
        g2_dbl(acc);
        if ((d >> bit) & 1u) g2_add(acc, b);
    }
    if (shift > 0) g2_dbl_n(acc, shift);
}

C12381_HD void fb_store_g2(int32_t* dst, const fp2& x, const fp2& y) {
    msm_store_pt(dst, x.a, x.b);
    msm_store_pt(dst + MSM_PT_DWORDS, y.a, y.b);
}
C12381_HD void fb_load_g2(fp2& x, fp2& y, const int32_t* src) {
    msm_load_pt(x.a, x.b, src);
    msm_load_pt(y.a, y.b, src + MSM_PT_DWORDS);
}

// acc = [k]B from the table (k: 8 little-endian words, any value; reduced mod r here)
C12381_HD void g1_fixed_eval(g1p& acc, const int32_t* tab, const uint32_t (&kin)[8]) {
    uint32_t k[8], k0[4], k1[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) k[i] = kin[i];
    scalar_mod_r(k);
    scalar_glv_split(k0, k1, k);
    fp beta;
    fp_set_const(beta, FP_BETA_A);
    g1_set_inf(acc);
#pragma unroll 1
    for (int j = 0; j < FB_G1_WINDOWS; ++j) {
        const uint32_t d0 = (k0[j >> 2] >> (8 * (j & 3))) & 255u, d1 = (k1[j >> 2] >> (8 * (j & 3))) & 255u;
        if (d0) {
            fp x, y;
            msm_load_pt(x, y, tab + ((size_t)j * FB_ENTRIES + (d0 - 1)) * FB_G1_DWORDS);
            g1_add_affine(acc, x, y);
        }
        if (d1) {
            fp x, y, bx, ny;
            msm_load_pt(x, y, tab + ((size_t)j * FB_ENTRIES + (d1 - 1)) * FB_G1_DWORDS);
            fp_mul(bx, x, beta);
            fp_neg(ny, y); fp_norm1(ny, ny);
            g1_add_affine(acc, bx, ny);
        }
    }
}
template <int I>
C12381_HD void g2_fixed_digit(g2p& acc, const int32_t* tab, const uint32_t (&u)[2]) {
#pragma unroll 1
    for (int j = 0; j < FB_G2_WINDOWS; ++j) {
        const uint32_t d = (u[j >> 2] >> (8 * (j & 3))) & 255u;
        if (d) {
            g2p q, e;
            fb_load_g2(q.x, q.y, tab + ((size_t)j * FB_ENTRIES + (d - 1)) * FB_G2_DWORDS);
            fp2_one(q.z);
            g2_psi_signed<I>(e, q, (I & 1) != 0);
            g2_norm1(e, e);
            g2_add(acc, e);
        }
    }
}
C12381_HDN void g2_fixed_eval(g2p& acc, const int32_t* tab, const uint32_t (&kin)[8]) {
    uint32_t k[8], u[4][2];
#pragma unroll
    for (int i = 0; i < 8; ++i) k[i] = kin[i];
    scalar_mod_r(k);
    scalar_gs_split(u, k);
    g2_set_inf(acc);
    g2_fixed_digit<0>(acc, tab, u[0]);
    g2_fixed_digit<1>(acc, tab, u[1]);
    g2_fixed_digit<2>(acc, tab, u[2]);
    g2_fixed_digit<3>(acc, tab, u[3]);
}

}  // namespace c12381
