// Bucket-method (Pippenger) multi-scalar multiplication in G1 — the fast path behind c12381_g1_msm.
// (The reference's Π[n](g[i]^x[i]) is n full scalar multiplications, g1_point.hpp:389-401; its bucket routine
//  ECP_muln ecp_BLS12381.cpp:1112-1148 is exported but unused.  Only the final point is observable.)
//
// Every term is evaluated exactly as PAIR_G1mul does — [k mod x^2] P + [k div x^2] (-phi(P)), plus the [r]phi(P) that
// multiply() adds for scalars below x^2: [r] and phi are homomorphisms, so those terms add up to [r]phi(S) for S = the plain
// sum of the points with a small scalar, which is ONE more bucket (key W << c) and one evaluation next to the Horner lane
// (msm_small_term) — so the sum equals the reference's chain of multiply() results for every input, points outside the
// order-r subgroup included.  Structure:
//   prep     per point: parse, on-curve check, Montgomery form of P and of P' = (beta x, -y); per (half, window)
//            one (key = window << c | digit, value = 2 i + half) entry, laid out window by window (msm_prep_one);
//            one more entry per point in the small-scalar segment
//   sort     the window positions are known when the entries are written, so only the c-bit digit is sorted: one rocPRIM
//            radix sort per window segment on bits [0, c) — two 8-bit passes instead of the three a 20-bit composite key
//            needs (small products: one call over all entries)
//   bucket   one lane per (window, digit): sum of its run of points with the complete MIXED addition
//   wreduce  per window sum_d d B_d by running sums over chunks of buckets, chunk offset by double-and-add
//   horner   sum_w 2^(c w) R_w
#pragma once
#include "g1.hpp"
#include "codec.hpp"

namespace c12381 {

constexpr int MSM_PT_DWORDS = 2 * NL;          // affine (x, y) in Montgomery form, normalised limbs: 112 B, seven 16-byte words
// stride of the records in the MSM's point array: 128 B, one cache line per gathered point.  Packed 112-byte records straddle two
// lines more often than not — the bucket kernel's gathers then fetched 16.6 GB per 2^22-term product for 7.5 GB of records
// (profiles/r03_pmc_summary.json; the A/B against the packed stride of 28: profiles/r03_ab_msm_record_stride_g2_local_point.txt).
constexpr int MSM_PT_STRIDE = 32;
constexpr int MSM_CHUNK = 8;                   // buckets per lane in the window reduction (8: measured 1.1 % faster than 16 at 2^22 terms, profiles/r04_ab_msm_front.txt)
// entries a bucket lane sums at most: twice the mean run + 32 (uniform scalars: mean + 11 sigma or more, so nothing is
// cut); the rest of a longer run is cut into overflow segments of half that length, one lane each (k_g1.hip).  A lane
// with a long run finishes alone at single-wavefront latency (~9 us per addition), hence a cap relative to the mean.
C12381_HD uint32_t msm_run_cap(size_t n, int c) { const size_t mean = (2 * n) >> c; return (uint32_t)(2 * mean + 32); }
C12381_HD int msm_window_bits(size_t n) {
    // 16 bits = 8 windows over the 128-bit halves with no narrow top window (a top window of t < c bits has 2^t buckets
    // with 2^(c-t) times longer runs); measured on MI355X, 2^12 .. 2^22 terms: c = 16 is the fastest width throughout
    // (tools/msm_sweep.py).  Below 2^12 terms the 2^19 buckets of c = 16 are mostly empty work for the window reduction:
    // c = 8 (16 windows, again no narrow one) — 2.7 ms against 3.3 ms, and against 4.3 ms for n scalar multiplications.
    return n >= 4096 ? 16 : 8;
}
C12381_HD int msm_windows(int c) { return (128 + c - 1) / c; }
C12381_HD size_t msm_entries(size_t n, int W) { return (size_t)(2 * W + 1) * n; }      // 2W digit entries + the small-scalar entry per term

// P + Q for an AFFINE Q = (qx, qy) that is not the point at infinity (Renes-Costello-Batina algorithm 8, a = 0):
// 11 products, 8 reductions.  P limb bound <= 2^29, Q normalised.
C12381_HD void g1_add_affine(g1p& p, const fp& qx, const fp& qy) {
    fp t0, t1, t2, t3, t4, y3, z3;
    fp_mul(t0, p.x, qx);
    fp_mul(t1, p.y, qy);
    {   // round 4: the linear terms ride in the reductions (fp_mul_inj): t3, t4 normalised without lazy sums or carry rounds
        const int32_t c1 = fp_opaque_const(1), cm1 = fp_opaque_const(-1);
        fp sa, sb;
        fp_add(sa, qx, qy); fp_add(sb, p.x, p.y);
        fp_mul_inj(t3, sa, sb, [&](int i, int64_t& acc) { fp_inj(acc, t0, i, cm1); fp_inj(acc, t1, i, cm1); }, C12381_BV(t0.vb + t1.vb), C12381_BV(t0.lb + t1.lb));    // X1 Y2 + X2 Y1
        fp_mul_inj(t4, qy, p.z, [&](int i, int64_t& acc) { fp_inj(acc, p.y, i, c1); }, C12381_BV(p.y.vb), C12381_BV(p.y.lb));                                         // Y2 Z1 + Y1
        fp_mul_inj(y3, qx, p.z, [&](int i, int64_t& acc) { fp_inj(acc, p.x, i, c1); }, C12381_BV(p.x.vb), C12381_BV(p.x.lb));                                         // X2 Z1 + X1
    }
    fp_mul_small(t0, t0, 3);
    fp_mul_small(t2, p.z, 12);                                                // b3 Z1
    fp_add(z3, t1, t2); fp_sub(t1, t1, t2);
    fp_mul_small(y3, y3, 12);
    fp_mul2<true>(p.x, t3, t1, y3, t4);                                       // X3 = t3 t1 - y3 t4
    fp_mul2<false>(p.y, y3, t0, t1, z3);                                      // Y3 = y3 t0 + t1 z3
    fp_mul2<false>(p.z, z3, t4, t0, t3);                                      // Z3 = z3 t4 + t0 t3
}

C12381_HD void msm_store_pt(int32_t* dst, const fp& x, const fp& y) {
    int32_t w[MSM_PT_DWORDS];
#pragma unroll
    for (int i = 0; i < NL; ++i) { w[i] = x.l[i]; w[NL + i] = y.l[i]; }
    q4* d = reinterpret_cast<q4*>(dst);
#pragma unroll
    for (int i = 0; i < MSM_PT_DWORDS / 4; ++i) { q4 t; t.v[0] = w[4 * i]; t.v[1] = w[4 * i + 1]; t.v[2] = w[4 * i + 2]; t.v[3] = w[4 * i + 3]; d[i] = t; }
}
C12381_HD void msm_load_pt(fp& x, fp& y, const int32_t* src) {
    int32_t w[MSM_PT_DWORDS];
#if defined(__HIP_DEVICE_COMPILE__)
    // the records live in global memory (point array of the MSM, line tables of a fixed G2 argument): said explicitly, because an
    // out-of-line routine sees a generic pointer and would use flat loads, whose every wait is a full vmcnt(0) + lgkmcnt(0) drain
    const __attribute__((address_space(1))) q4* s = (const __attribute__((address_space(1))) q4*)(const void*)src;
#else
    const q4* s = reinterpret_cast<const q4*>(src);
#endif
#pragma unroll
    for (int i = 0; i < MSM_PT_DWORDS / 4; ++i) { q4 t = s[i]; w[4 * i] = t.v[0]; w[4 * i + 1] = t.v[1]; w[4 * i + 2] = t.v[2]; w[4 * i + 3] = t.v[3]; }
#pragma unroll
    for (int i = 0; i < NL; ++i) { x.l[i] = w[i]; y.l[i] = w[NL + i]; }
    C12381_BOUNDS(x.lb = y.lb = 268435456.0 + 8.0; x.vb = y.vb = 2.0; check_actual(x, "msm_load_pt"); check_actual(y, "msm_load_pt");)
}
// c-bit digit of window w of a 128-bit value (4 little-endian words)
C12381_HD uint32_t msm_digit(const uint32_t (&k)[4], int w, int c) {
    const int bit = w * c;
    const int wi = bit >> 5, sh = bit & 31;
    uint64_t v = k[wi];
    if (wi + 1 < 4) v |= (uint64_t)k[wi + 1] << 32;
    uint32_t d = (uint32_t)(v >> sh);
    if (c < 32) d &= (1u << c) - 1u;
    if (bit + c > 128) d &= (1u << (128 - bit)) - 1u;
    return d;
}
// The value of the entry at position x of a window segment (half 0 of term i at x = i, half 1 at x = n + i): the index of its point record.
// It depends on the position alone, so the large products never store the unsorted values: the sort reads them from this function
// through an iterator (c12381_hip.hip) — 68 B of writes per term and as many bytes of the first sorting pass saved.
C12381_HD uint32_t msm_entry_value(uint32_t x, uint32_t n) { return x < n ? 2u * x : 2u * (x - n) + 1u; }
// prep: returns false if the point is not on the curve.  pts2 holds P at 2i and P' = (beta x, -y) at 2i+1.
// K = uint32_t: key = window << c | digit, sorted in one call over all entries (small products);  K = uint16_t: key = the digit alone —
// the window of an entry is its position, every window segment is sorted on its own and 16-bit keys halve the key traffic of the two
// radix passes (round 4; c <= 16).  vals may be null (the values are positional: msm_entry_value).
template <class K>
C12381_HD bool msm_prep_one(size_t i, size_t n, const uint32_t* raw_pt /*24 words*/, const uint32_t* raw_sc /*8 words*/, int c, int W,
                            int32_t* pts2, K* keys, uint32_t* vals) {
    constexpr bool WIDE = sizeof(K) == 4;
    fp px, py;
    const bool inf = raw_all_zero(raw_pt, 24);
    fp_from_raw48(px, raw_pt); fp_from_raw48(py, raw_pt + 12);
    bool ok = true;
    if (!inf) {
        fp x2, x3, y2, four, rhs;
        fp_sqr(x2, px); fp_mul(x3, x2, px);
        fp_set_const(four, FP_FOUR);
        fp_add(rhs, x3, four);
        fp_sqr(y2, py);
        ok = fp_equal(y2, rhs);
    }
    const bool usable = ok && !inf;
    fp beta, bx, ny, nyn;
    fp_set_const(beta, FP_BETA_A);
    fp_mul(bx, px, beta);
    fp_neg(ny, py);
    fp_norm1(nyn, ny);
    msm_store_pt(pts2 + (2 * i) * MSM_PT_STRIDE, px, py);
    msm_store_pt(pts2 + (2 * i + 1) * MSM_PT_STRIDE, bx, nyn);
    uint32_t k[8];
    scalar_from_raw32(k, raw_sc);
    scalar_mod_r(k);
    uint32_t k0[4], k1[4];
    scalar_glv_split(k0, k1, k);
    // Entry layout (what lets the sort work on c bits only): window w owns the 2n consecutive entries [2 w n, 2 (w + 1) n) —
    // half 0 of term i at 2 w n + i, half 1 at 2 w n + n + i — with key = w << c | digit; digit 0 (nothing to add: a zero
    // digit, the point at infinity, a point that is not on the curve) sorts to the front of its window and is skipped by
    // msm_ranges.  The last n entries are the small-scalar segment: key = W << c | 1 if the term owes [r]phi(P), else W << c.
    const bool small = (k1[0] | k1[1] | k1[2] | k1[3]) == 0u;             // k mod r < x^2: multiply() owes [r]phi(P)
    keys[(size_t)(2 * W) * n + i] = (K)((WIDE ? ((uint32_t)W << c) : 0u) | ((usable && small) ? 1u : 0u));
    if (vals) vals[(size_t)(2 * W) * n + i] = (uint32_t)(2 * i);
    for (int w = 0; w < W; ++w) {
        const uint32_t d0 = msm_digit(k0, w, c), d1 = msm_digit(k1, w, c);
        const size_t e0 = ((size_t)(2 * w)) * n + i, e1 = e0 + n;
        const uint32_t hi_bits = WIDE ? ((uint32_t)w << c) : 0u;
        keys[e0] = (K)(hi_bits | (usable ? d0 : 0u));
        keys[e1] = (K)(hi_bits | (usable ? d1 : 0u));
        if (vals) { vals[e0] = (uint32_t)(2 * i); vals[e1] = (uint32_t)(2 * i + 1); }
    }
    return ok;
}
// sorted entry j -> the bucket whose run it starts / ends.  Buckets: w << c | digit for the digit windows, W << c for the
// small-scalar bucket; entries with digit 0 belong to no bucket.
C12381_HD void msm_ranges_one(size_t j, size_t E, const uint32_t* keys, int c, int W, uint32_t* lo, uint32_t* hi) {
    const uint32_t k = keys[j], d = k & ((1u << c) - 1u);
    if (d == 0) return;
    const uint32_t b = (k >> c) < (uint32_t)W ? k : ((uint32_t)W << c);
    if (j == 0 || keys[j - 1] != k) lo[b] = (uint32_t)j;
    if (j + 1 == E || keys[j + 1] != k) hi[b] = (uint32_t)(j + 1);
}
// bucket: sum of the points whose (sorted) entries lie in [lo, hi)
// The gather (index -> 112-byte record somewhere in a table of 2n records) is a two-step dependent load of a few
// microseconds; it is software-pipelined: while point j is added, point j+1 and index j+2 are already in flight.
C12381_HD void msm_bucket_one(g1p& acc, size_t lo, size_t hi, const uint32_t* vals_sorted, const int32_t* pts2) {
    g1_set_inf(acc);
    if (lo >= hi) return;
    fp xn, yn;
    msm_load_pt(xn, yn, pts2 + (size_t)vals_sorted[lo] * MSM_PT_STRIDE);
    uint32_t idx_next = lo + 1 < hi ? vals_sorted[lo + 1] : 0u;
#pragma unroll 1
    for (size_t j = lo; j < hi; ++j) {
        const fp x = xn, y = yn;
        if (j + 1 < hi) {
            msm_load_pt(xn, yn, pts2 + (size_t)idx_next * MSM_PT_STRIDE);
            idx_next = j + 2 < hi ? vals_sorted[j + 2] : 0u;
        }
        g1_add_affine(acc, x, y);
    }
}
// window reduction for the chunk of MSM_CHUNK consecutive digits starting at d0 (d0 multiple of MSM_CHUNK):
//   sum_{j} (d0 + j) B_{d0+j} = [d0] S + sum_j j B_{d0+j},  S = sum_j B_{d0+j}
// `bk` = this window's buckets as 44-dword records (digit-indexed), nb = 2^c
C12381_HD void msm_wreduce_one(g1p& out, const int32_t* bk, uint32_t d0, uint32_t nb) {
    g1p run, acc, b;
    g1_set_inf(run); g1_set_inf(acc);
    for (int j = MSM_CHUNK - 1; j >= 1; --j) {
        const uint32_t d = d0 + (uint32_t)j;
        if (d < nb) {                              // uniform across the wavefront except in the last chunk
            tab_load_g1(b, bk + (size_t)d * G1_ENT_DWORDS);
            g1_add(run, b);
            g1p nn; g1_norm1(nn, run); run = nn;
            g1_add(acc, run);
            g1_norm1(nn, acc); acc = nn;
        }
    }
    if (d0 < nb) { tab_load_g1(b, bk + (size_t)d0 * G1_ENT_DWORDS); g1_add(run, b); g1p nn; g1_norm1(nn, run); run = nn; }
    // [d0] S by double-and-add (d0 < 2^16), selects instead of branches.  d0 is a multiple of the chunk length: its low bits are
    // doublings only (round 4: log2(MSM_CHUNK) additions of the point at infinity less)
    constexpr int LOW = MSM_CHUNK >= 16 ? 4 : (MSM_CHUNK >= 8 ? 3 : (MSM_CHUNK >= 4 ? 2 : (MSM_CHUNK >= 2 ? 1 : 0)));
    static_assert((1 << LOW) == MSM_CHUNK || MSM_CHUNK > 16, "MSM_CHUNK is a power of two up to 16 (or a larger multiple of 16)");
    g1p t, inf;
    g1_set_inf(t); g1_set_inf(inf);
    for (int bit = 15; bit >= LOW; --bit) {
        g1_dbl(t);
        g1p s;
        const bool on = (d0 >> bit) & 1u;
        fp_select(s.x, on, run.x, inf.x); fp_select(s.y, on, run.y, inf.y); fp_select(s.z, on, run.z, inf.z);
        g1p tn; g1_norm1(tn, t);
        g1_add(tn, s);
        t = tn;
    }
    for (int bit = 0; bit < LOW; ++bit) g1_dbl(t);
    g1p tn; g1_norm1(tn, t);
    g1_add(tn, acc);
    g1_norm1(out, tn);
}
// R = sum_w 2^(c w) R_w   (R_w given as limb-major SoA, element w)
C12381_HD void msm_horner(g1p& acc, const int32_t* rw, size_t stride, int W, int c) {
    soa_load_g1(acc, rw, stride, (size_t)(W - 1));
    for (int w = W - 2; w >= 0; --w) {
        for (int b = 0; b < c; ++b) g1_dbl(acc);
        g1p q, nn;
        soa_load_g1(q, rw, stride, (size_t)w);
        g1_norm1(nn, acc);
        g1_add(nn, q);
        acc = nn;
    }
    g1p nn; g1_norm1(nn, acc); acc = nn;
}

// the [r]phi(S) owed for S = the sum of the points whose scalar is below x^2 (bucket W << c); infinity when S is in G1
C12381_HDN void msm_small_term(g1p& term, const g1p& S) {
    g1_set_inf(term);
    if (!g1_is_inf(S)) g1_glv_small_scalar_term(term, S);
}

}  // namespace c12381
