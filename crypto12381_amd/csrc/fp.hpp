// Fp arithmetic for BLS12-381 on CDNA4 — the leaf of the whole hot path
// (replaces FP_mul/FP_sqr/FP_add/FP_sub/FP_neg/FP_imul/FP_inv/FP_sqrt of the reference's
// fp_BLS12381.cpp:396-936 and BIG_mul/BIG_sqr/BIG_monty of big_B384_58.cpp:570-981).
//
// Number format (designed for gfx950, not MIRACL's 7x58-bit/__int128 format):
//   * 14 SIGNED limbs of 28 bits, little-endian; Montgomery radix R = 2^392.
//   * limbs are carried lazily: add/sub/neg are 14 independent v_add/v_sub with no carry
//     chain and no conditional subtraction; values may be negative or exceed p.
//   * a product is scanned column by column into ONE 64-bit accumulator with
//     v_mad_i64_i32 (measured inside the kernel, profiles/r03_issue_mix.txt: one per 4.125 cycles and SIMD — and so is every other
//     vector instruction of a mixed stream at two wavefronts per SIMD: what counts is the NUMBER of instructions),
//     Montgomery reduction interleaved in the same columns — no carry flags anywhere.
//   * invariants are tracked as two bounds per element: LB = max |limb| and VB = |value|/p.
//     fp_mul needs 14*LBa*LBb + 14*2^56 + 2^40 < 2^63 and returns limbs 0..12 in [0,2^28) with
//     |value| < (VBa*VBb*p/R + 1) p, i.e. within (-p, 2p) whenever VBa*VBb <= 2^11.  Builds with C12381_CHECK_BOUNDS (host
//     simulation used by the CPU tests) carry the bounds at run time and assert them; the
//     bounds depend only on the operation sequence, never on the data.
#pragma once
#include <cstdint>
#include <type_traits>
#include <utility>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define C12381_HD __host__ __device__ __forceinline__
#define C12381_HDN __host__ __device__ __noinline__ inline      // big tower/curve routines: real calls, operands in scratch
#define C12381_CONST __device__ constexpr
#else
#define C12381_HD inline
#define C12381_HDN inline
#define C12381_CONST constexpr
#endif

// A normalised limb (x & LMASK) is known to be non-negative, and LLVM then canonicalises its sign extension to a ZERO
// extension; once such a value crosses a basic-block boundary (every accumulator of a loop does) instruction
// selection no longer sees that bit 31 is clear, cannot use v_mad_i64_i32 for sext(a) * zext(b) and emits TWO
// v_mad_u64_u32 plus two moves per product.  Hiding the mask result behind an empty asm keeps both operands of
// every product plain signed 32-bit values: one v_mad_i64_i32 each.  (No instruction is emitted for the asm.)
#if defined(__HIP_DEVICE_COMPILE__)
#define C12381_LIMB(x) c12381::limb_opaque(x)
namespace c12381 { __device__ __forceinline__ int32_t limb_opaque(int32_t v) { asm("" : "+v"(v)); return v; } }
#else
#define C12381_LIMB(x) (x)
#endif
// How the column sums reach the hardware as ONE linear v_mad_i64_i32 chain (round 4): left alone, LLVM's Reassociate pass sums every column
// from zero and adds the carry of the previous column afterwards (one v_lshl_add_u64 "join" per column, 26-30 per reduction) and splits the signed
// limb products and the non-negative m * p products into two chains.  The build switches that pass off (-mllvm -opt-disable=reassociate,
// crypto12381_amd/build.py): the source order below — carry first, then the products — survives to instruction selection, fp_mul is 463
// instructions (hand count 460) and needs 44 registers instead of 84.
// What did NOT work: an empty asm on the accumulator after every multiply-add (round 3, profiles/r03_ab_acc_fence_g2_inline.txt) — same chain, but the
// hazard recognizer pads every edge from an inline asm that defines a VGPR to the instruction that reads it with one s_nop (345 s_nop for 208
// joins saved in the bucket kernel; pairing kernel 18.1 -> 24.4 ms); every step as its own one-instruction asm statement (round 4): exactly the
// hand-written stream and one s_nop after EVERY instruction (asm -> asm edges are padded too).  A whole multiplier in one asm statement is out
// of reach in HIP C++ (42 register operands against the limit of 30).

#include "consts.hpp"

#ifdef C12381_CHECK_BOUNDS
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <execinfo.h>
#define C12381_BOUNDS(...) __VA_ARGS__
#define C12381_BV(x) (x)                        // a bound expression as a function argument (0 in device builds: fp has no bound fields there)
// operation counters of the host simulation (tools/count_ops.py): columns of 14 x 14 limb products scanned (27 per product) and
// Montgomery reductions — the work a device routine does in THIS number format, and, priced at 144 / 156 multiply-adds, the
// algorithmic MAC32 count of SURVEY.md 8(d) for the operation sequence as built
namespace c12381 { inline std::atomic<unsigned long long> g_ops_cols{0}, g_ops_reds{0}; }
#define C12381_COUNT(cols, reds) do { c12381::g_ops_cols.fetch_add((cols), std::memory_order_relaxed); c12381::g_ops_reds.fetch_add((reds), std::memory_order_relaxed); } while (0)
#else
#define C12381_BOUNDS(...)
#define C12381_BV(x) 0.0
#define C12381_COUNT(cols, reds)
#endif

namespace c12381 {

struct fp {
    int32_t l[NL];
#ifdef C12381_CHECK_BOUNDS
    double lb = 0, vb = 0;      // declared bounds: max |limb|, |value| / p
#endif
};

#ifdef C12381_CHECK_BOUNDS
inline void bounds_fail(const char* what, double a, double b) {
    std::fprintf(stderr, "C12381 bound violation: %s (%.4g, %.4g)\n", what, a, b);
    void* frames[48];
    int nf = backtrace(frames, 48);
    backtrace_symbols_fd(frames, nf, 2);      // resolve with addr2line -e libsim.so <offsets> (build with -g)
    std::abort();
}
constexpr double P_OVER_R = 0.000396;          // p / 2^392 < this
constexpr double TOP_PER_P = 106514.0;         // p / 2^364 < this: |top limb| <= VB * TOP_PER_P + 1
inline void check_actual(const fp& a, const char* where) {
    for (int i = 0; i < NL; ++i)
        if (std::fabs((double)a.l[i]) > a.lb) bounds_fail(where, (double)a.l[i], a.lb);
}
inline void check_actual_vb(const fp& a, const char* where);
inline void set_inj_bounds(fp& r, double sum_lblb, double sum_vbvb, double inj_abs, double inj_lb, const char* where);
inline void set_bounds(fp& r, double lb, double vb, const char* where) {
    r.lb = lb; r.vb = vb;
    if (lb > 2147483648.0) bounds_fail("limb bound exceeds int32", lb, vb);
    check_actual(r, where);
}
#endif

// ------------------------------------------------------------------ limb-wise lazy ops
C12381_HD void fp_add(fp& r, const fp& a, const fp& b) {
#pragma unroll
    for (int i = 0; i < NL; ++i) r.l[i] = a.l[i] + b.l[i];
    C12381_BOUNDS(set_bounds(r, a.lb + b.lb, a.vb + b.vb, "fp_add");)
}
C12381_HD void fp_sub(fp& r, const fp& a, const fp& b) {
#pragma unroll
    for (int i = 0; i < NL; ++i) r.l[i] = a.l[i] - b.l[i];
    C12381_BOUNDS(set_bounds(r, a.lb + b.lb, a.vb + b.vb, "fp_sub");)
}
C12381_HD void fp_neg(fp& r, const fp& a) {
#pragma unroll
    for (int i = 0; i < NL; ++i) r.l[i] = -a.l[i];
    C12381_BOUNDS(set_bounds(r, a.lb, a.vb, "fp_neg");)
}
C12381_HD void fp_dbl(fp& r, const fp& a) { fp_add(r, a, a); }
C12381_HD void fp_zero(fp& r) {
#pragma unroll
    for (int i = 0; i < NL; ++i) r.l[i] = 0;
    C12381_BOUNDS(r.lb = 0; r.vb = 0;)
}
C12381_HD void fp_set_const(fp& r, const int32_t (&c)[NL]) {
#pragma unroll
    for (int i = 0; i < NL; ++i) r.l[i] = c[i];
    C12381_BOUNDS(r.lb = 268435456.0; r.vb = 1.0;)
}
C12381_HD void fp_one(fp& r) { fp_set_const(r, FP_R1); }
// r = c ? a : b   (lane-wise select, no divergence)
C12381_HD void fp_select(fp& r, bool c, const fp& a, const fp& b) {
#pragma unroll
    for (int i = 0; i < NL; ++i) r.l[i] = c ? a.l[i] : b.l[i];
    C12381_BOUNDS(r.lb = a.lb > b.lb ? a.lb : b.lb; r.vb = a.vb > b.vb ? a.vb : b.vb;)
}

// One parallel carry round: limbs 0..12 -> [0, 2^28) + incoming carry, top limb keeps the sign.
// Output LB = 2^28 + (LB_in >> 28) + 1.  No dependency chain between limbs.
C12381_HD void fp_norm1(fp& r, const fp& a) {
    int32_t c[NL];
#pragma unroll
    for (int i = 0; i < NL - 1; ++i) c[i] = a.l[i] >> LB;
    int32_t t13 = a.l[NL - 1] + c[NL - 2];
#pragma unroll
    for (int i = NL - 2; i >= 1; --i) r.l[i] = C12381_LIMB((int32_t)((uint32_t)a.l[i] & LMASK)) + c[i - 1];
    r.l[0] = C12381_LIMB((int32_t)((uint32_t)a.l[0] & LMASK));
    r.l[NL - 1] = t13;
    C12381_BOUNDS({ double top = a.vb * TOP_PER_P + 4.0 + std::floor(a.lb / 268435456.0);
                    double lb = 268435456.0 + std::floor(a.lb / 268435456.0) + 1.0;
                    set_bounds(r, lb > top ? lb : top, a.vb, "fp_norm1"); })
}

// r = k * a for a small non-negative integer k (k * LB may exceed int32: carries are
// propagated exactly with a 64-bit running value).  Replaces FP_imul fp_BLS12381.cpp:420.
C12381_HD void fp_mul_small(fp& r, const fp& a, int32_t k) {
    int64_t t = 0;
#pragma unroll
    for (int i = 0; i < NL - 1; ++i) {
        t += (int64_t)a.l[i] * k;
        r.l[i] = C12381_LIMB((int32_t)((uint32_t)t & LMASK));
        t >>= LB;
    }
    t += (int64_t)a.l[NL - 1] * k;
    r.l[NL - 1] = (int32_t)t;
    C12381_BOUNDS({ double top = a.vb * k * TOP_PER_P + 2.0;
                    set_bounds(r, top > 268435456.0 ? top : 268435456.0, a.vb * k, "fp_mul_small"); })
}

// Weak reduction: subtract the multiple q*p suggested by the top limb so that |value| drops to
// about p/2 + 2^365, and renormalise all limbs exactly (sequential carries).  ~90 simple ops — used
// where a value is carried through additions only (no product to re-bound it), e.g. the linear
// terms of the cyclotomic squaring.  Plays the role of FP_reduce fp_BLS12381.cpp:549-579.
C12381_HD void fp_weak_reduce(fp& r, const fp& a) {
    fp n;
    fp_norm1(n, a);
    // q ~ value / p ~ top / 106513.08  (2^32 / 106513.08 = 40323.6); off-by-one is harmless
    const int32_t q = (int32_t)(((int64_t)n.l[NL - 1] * 40324 + ((int64_t)1 << 31)) >> 32);
    int64_t t = 0;
#pragma unroll
    for (int i = 0; i < NL - 1; ++i) {
        t += (int64_t)n.l[i] - (int64_t)q * FP_P[i];
        r.l[i] = C12381_LIMB((int32_t)((uint32_t)t & LMASK));
        t >>= LB;
    }
    t += (int64_t)n.l[NL - 1] - (int64_t)q * FP_P[NL - 1];
    r.l[NL - 1] = (int32_t)t;
    C12381_BOUNDS({ if (a.vb > 2400.0) bounds_fail("fp_weak_reduce input value bound", a.vb, 0);
                    set_bounds(r, 268435456.0, 1.6, "fp_weak_reduce"); })
}

// ------------------------------------------------------------------ Montgomery multiply / square
#ifdef C12381_CHECK_BOUNDS
inline void check_mul_operands(const fp& a, const fp& b, const char* where) {
    check_actual(a, where); check_actual(b, where);
    double col = 14.0 * a.lb * b.lb + 14.0 * 72057594037927936.0 + 1099511627776.0;
    if (col >= 9223372036854775808.0) bounds_fail(where, a.lb, b.lb);
    if (a.vb * b.vb > 1.0e6) bounds_fail("fp_mul value bound", a.vb, b.vb);   // keeps |result| < 400 p; the exact bound is tracked in vb
}
#endif

// r = a * b / R mod p.   14x14 product scanning + interleaved Montgomery reduction:
// 392 v_mad_i64_i32 + 14 v_mul_lo_u32 and ~70 shifts/masks per call.
C12381_HD void fp_mul(fp& r, const fp& a, const fp& b) {
    C12381_BOUNDS(check_mul_operands(a, b, "fp_mul");)
    C12381_COUNT(27, 1);
    int32_t m[NL];
    int32_t out[NL];
    int64_t acc = 0;
#pragma unroll
    for (int k = 0; k < NL; ++k) {
#pragma unroll
        for (int i = 0; i <= k; ++i) { acc += (int64_t)a.l[i] * b.l[k - i]; }
#pragma unroll
        for (int i = 0; i < k; ++i) { acc += (int64_t)m[i] * FP_P[k - i]; }
        m[k] = (int32_t)(((uint32_t)acc * FP_N0) & LMASK);
        acc += (int64_t)m[k] * FP_P[0];
        acc >>= LB;
    }
#pragma unroll
    for (int k = NL; k < 2 * NL - 1; ++k) {
#pragma unroll
        for (int i = k - NL + 1; i < NL; ++i) { acc += (int64_t)a.l[i] * b.l[k - i]; }
#pragma unroll
        for (int i = k - NL + 1; i < NL; ++i) { acc += (int64_t)m[i] * FP_P[k - i]; }
        out[k - NL] = C12381_LIMB((int32_t)((uint32_t)acc & LMASK));
        acc >>= LB;
    }
    out[NL - 1] = (int32_t)acc;
#pragma unroll
    for (int i = 0; i < NL; ++i) r.l[i] = out[i];
    C12381_BOUNDS({ double vb = a.vb * b.vb * P_OVER_R + 1.0;
                    double top = vb * TOP_PER_P + 2.0;
                    set_bounds(r, top > 268435456.0 ? top : 268435456.0, vb, "fp_mul"); })
}

// r = a^2 / R mod p.  Cross terms once, doubled per column: 105 + 196 multiply-adds.
C12381_HD void fp_sqr(fp& r, const fp& a) {
    C12381_BOUNDS(check_mul_operands(a, a, "fp_sqr");)
    C12381_COUNT(27, 1);
    int32_t m[NL];
    int32_t out[NL];
    int32_t a2[NL];                      // 2a: the cross terms accumulate straight into the column (limbs <= 2^30)
#pragma unroll
    for (int i = 0; i < NL; ++i) a2[i] = 2 * a.l[i];
    int64_t acc = 0;
#pragma unroll
    for (int k = 0; k < NL; ++k) {
#pragma unroll
        for (int i = 0; 2 * i < k; ++i) { acc += (int64_t)a2[i] * a.l[k - i]; }
        if ((k & 1) == 0) { acc += (int64_t)a.l[k / 2] * a.l[k / 2]; }
#pragma unroll
        for (int i = 0; i < k; ++i) { acc += (int64_t)m[i] * FP_P[k - i]; }
        m[k] = (int32_t)(((uint32_t)acc * FP_N0) & LMASK);
        acc += (int64_t)m[k] * FP_P[0];
        acc >>= LB;
    }
#pragma unroll
    for (int k = NL; k < 2 * NL - 1; ++k) {
#pragma unroll
        for (int i = k - NL + 1; 2 * i < k; ++i) { acc += (int64_t)a2[i] * a.l[k - i]; }
        if ((k & 1) == 0) { acc += (int64_t)a.l[k / 2] * a.l[k / 2]; }
#pragma unroll
        for (int i = k - NL + 1; i < NL; ++i) { acc += (int64_t)m[i] * FP_P[k - i]; }
        out[k - NL] = C12381_LIMB((int32_t)((uint32_t)acc & LMASK));
        acc >>= LB;
    }
    out[NL - 1] = (int32_t)acc;
#pragma unroll
    for (int i = 0; i < NL; ++i) r.l[i] = out[i];
    C12381_BOUNDS({ double vb = a.vb * a.vb * P_OVER_R + 1.0;
                    double top = vb * TOP_PER_P + 2.0;
                    set_bounds(r, top > 268435456.0 ? top : 268435456.0, vb, "fp_sqr"); })
}

// ------------------------------------------------------------------ lazy reduction: sums of products
// One Montgomery reduction for a whole bilinear form  r = (sum_t +-A_t*B_t) / R mod p
// (the reference's FP2_mul does the same trick with double-length BIGs, fp2_BLS12381.cpp:266-302).
// `col(k, acc)` ADDS the k-th column  sum_{i+j=k} (...)  of the un-reduced form to the running accumulator, k = 0..26
// (straight into it: a column that is summed on the side and then added costs a 64-bit add per column and form);
// the engine interleaves the reduction exactly like fp_mul.  Column sums must stay below 2^63:
// with |limbs| <= LBa, LBb that is  14 * T * LBa * LBb + 14 * 2^56 + 2^40 < 2^63  for T products.
// Differences are formed by negating one operand's limbs once (fp_raw_neg), squares use a pre-doubled copy.
C12381_HD void fp_col_acc(int64_t& acc, const fp& a, const fp& b, int k) {
    C12381_COUNT(1, 0);
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const int j = k - i;
        if (j >= 0 && j < NL) { acc += (int64_t)a.l[i] * b.l[j]; }
    }
}
// acc += column k of s * a^2 given a2 = 2 s a and ad = s a  (s = +1 or -1): cross terms once, diagonal term
C12381_HD void fp_col_sqr_acc(int64_t& acc, const fp& a, const fp& a2, const fp& ad, int k) {
    C12381_COUNT(1, 0);
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const int j = k - i;
        if (j > i && j < NL) { acc += (int64_t)a2.l[i] * a.l[j]; }
    }
    if ((k & 1) == 0 && k / 2 < NL) { acc += (int64_t)ad.l[k / 2] * a.l[k / 2]; }
}
// limb-wise -a, 2a, -2a: operands of the column scans only (never normalised, never stored)
C12381_HD void fp_raw_neg(fp& r, const fp& a) {
#pragma unroll
    for (int i = 0; i < NL; ++i) r.l[i] = -a.l[i];
    C12381_BOUNDS(r.lb = a.lb; r.vb = a.vb;)
}
C12381_HD void fp_raw_dbl(fp& r, const fp& a) {
#pragma unroll
    for (int i = 0; i < NL; ++i) r.l[i] = 2 * a.l[i];
    C12381_BOUNDS(r.lb = 2 * a.lb; r.vb = 2 * a.vb;)
}
C12381_HD void fp_raw_neg_dbl(fp& r, const fp& a) {
#pragma unroll
    for (int i = 0; i < NL; ++i) r.l[i] = -2 * a.l[i];
    C12381_BOUNDS(r.lb = 2 * a.lb; r.vb = 2 * a.vb;)
}
template <class ColFn>
C12381_HD void fp_reduce_cols(fp& r, ColFn col) {
    C12381_COUNT(0, 1);
    int32_t m[NL];
    int32_t out[NL];
    int64_t acc = 0;
#pragma unroll
    for (int k = 0; k < NL; ++k) {
        col(k, acc);
#pragma unroll
        for (int i = 0; i < k; ++i) { acc += (int64_t)m[i] * FP_P[k - i]; }
        m[k] = (int32_t)(((uint32_t)acc * FP_N0) & LMASK);
        acc += (int64_t)m[k] * FP_P[0];
        acc >>= LB;
    }
#pragma unroll
    for (int k = NL; k < 2 * NL - 1; ++k) {
        col(k, acc);
#pragma unroll
        for (int i = k - NL + 1; i < NL; ++i) { acc += (int64_t)m[i] * FP_P[k - i]; }
        out[k - NL] = C12381_LIMB((int32_t)((uint32_t)acc & LMASK));
        acc >>= LB;
    }
    out[NL - 1] = (int32_t)acc;
#pragma unroll
    for (int i = 0; i < NL; ++i) r.l[i] = out[i];
}
// The same engine with the column index as a compile-time constant (an index pack instead of `#pragma unroll` loops): forms with
// four products per column are past the size at which LLVM still honours the unroll request for the outer loops — it then emits
// real loops with indexed register access (s_set_gpr_idx) and 64-bit unsigned multiply-adds.  Here nothing is left to the
// unroller's heuristics: col() sees a literal k and its own limb loops have 14 constant iterations with foldable conditions.
template <int... Ks, class Fn>
C12381_HD void fp_static_for(std::integer_sequence<int, Ks...>, Fn&& fn) { (fn(std::integral_constant<int, Ks>{}), ...); }
template <class ColFn>
C12381_HD void fp_reduce_cols_static(fp& r, ColFn col) {
    C12381_COUNT(0, 1);
    int32_t m[NL];
    int32_t out[NL];
    int64_t acc = 0;
    fp_static_for(std::make_integer_sequence<int, NL>{}, [&](auto kc) {
        constexpr int k = decltype(kc)::value;
        col(k, acc);
#pragma unroll
        for (int i = 0; i < k; ++i) { acc += (int64_t)m[i] * FP_P[k - i]; }
        m[k] = (int32_t)(((uint32_t)acc * FP_N0) & LMASK);
        acc += (int64_t)m[k] * FP_P[0];
        acc >>= LB;
    });
    fp_static_for(std::make_integer_sequence<int, NL - 1>{}, [&](auto kc) {
        constexpr int k = NL + decltype(kc)::value;
        col(k, acc);
#pragma unroll
        for (int i = k - NL + 1; i < NL; ++i) { acc += (int64_t)m[i] * FP_P[k - i]; }
        out[k - NL] = C12381_LIMB((int32_t)((uint32_t)acc & LMASK));
        acc >>= LB;
    });
    out[NL - 1] = (int32_t)acc;
#pragma unroll
    for (int i = 0; i < NL; ++i) r.l[i] = out[i];
}
// ------------------------------------------------------------------ reductions with linear terms injected (round 4)
// R = 2^(28 * 14): an addend c enters a Montgomery reduction as c * R, i.e. as c's limb i in column 14 + i —
//     (T + R * sum_j k_j c_j + m p) / R  =  T / R + sum_j k_j c_j     (mod p, and as integers up to the usual + m p / R < p).
// One multiply-add per limb and addend (the multiplier k_j is a small constant or a per-lane register: signs and role-dependent choices
// ride on it), and the sum comes out of the reduction NORMALISED: the lazy additions after a reduction, the carry round they force
// before the next product and the selects around them disappear.  inj(i, acc) adds the limbs i of the addends (i = 0..13; limb 13 lands
// on the top limb, behind the last column).  A multiple of p rides along the same way (fp_inj_p): a linear term that passes through
// unmultiplied — the 2 conj(x) of the cyclotomic squaring — is re-bounded by "- q p" with q = round(term / p) from its top limb, 14
// multiply-adds instead of a weak reduction (~90 instructions) every other call.
// Bounds: the addends' limbs are noise against the 2^56-sized products of a column (|k| LB <= 2^32 against a headroom of >= 2^59, asserted
// by the checker through the callers' declarations); the VALUE bound of the result is declared by the caller (set_inj_bounds).
C12381_HD void fp_inj(int64_t& acc, const fp& c, int i, int32_t k) { acc += (int64_t)c.l[i] * k; }
C12381_HD void fp_inj_p(int64_t& acc, int i, int32_t k) { acc += (int64_t)FP_P[i] * k; }
// round(top / (p / 2^364)) for a value whose lower limbs are normalised: the multiple of p nearest to the value (within 2 p / 106513);
// the same estimate fp_weak_reduce uses
C12381_HD int32_t fp_quot_top(int32_t top) { return (int32_t)(((int64_t)top * 40324 + ((int64_t)1 << 31)) >> 32); }
// a small integer the optimiser cannot see through: multipliers of injected terms must reach instruction selection as REGISTER operands
// of v_mad_i64_i32 (a literal 2 or -1 is strength-reduced into sign extension + 64-bit shift / add: two or three instructions)
#if defined(__HIP_DEVICE_COMPILE__)
C12381_HD int32_t fp_opaque_const(int32_t v) { asm("" : "+s"(v)); return v; }
#else
C12381_HD int32_t fp_opaque_const(int32_t v) { return v; }
#endif
template <class ColFn, class InjFn>
C12381_HD void fp_reduce_cols_inj(fp& r, ColFn col, InjFn inj) {
    C12381_COUNT(0, 1);
    int32_t m[NL];
    int32_t out[NL];
    int64_t acc = 0;
#pragma unroll
    for (int k = 0; k < NL; ++k) {
        col(k, acc);
#pragma unroll
        for (int i = 0; i < k; ++i) { acc += (int64_t)m[i] * FP_P[k - i]; }
        m[k] = (int32_t)(((uint32_t)acc * FP_N0) & LMASK);
        acc += (int64_t)m[k] * FP_P[0];
        acc >>= LB;
    }
#pragma unroll
    for (int k = NL; k < 2 * NL - 1; ++k) {
        col(k, acc);
        inj(k - NL, acc);
#pragma unroll
        for (int i = k - NL + 1; i < NL; ++i) { acc += (int64_t)m[i] * FP_P[k - i]; }
        out[k - NL] = C12381_LIMB((int32_t)((uint32_t)acc & LMASK));
        acc >>= LB;
    }
    inj(NL - 1, acc);
    out[NL - 1] = (int32_t)acc;
#pragma unroll
    for (int i = 0; i < NL; ++i) r.l[i] = out[i];
}
// r = a * b / R + (injected addends): one product, one reduction, linear terms injected; bounds for the checker: inj_vb = sum |k| VB(c)
// (or what the caller proves), inj_lb = sum |k| LB(c)
template <class InjFn>
C12381_HD void fp_mul_inj(fp& r, const fp& a, const fp& b, InjFn inj, double inj_vb, double inj_lb) {
    C12381_COUNT(27, 0);
    fp t;
    fp_reduce_cols_inj(t, [&](int k, int64_t& acc) {
#pragma unroll
        for (int i = 0; i < NL; ++i) { const int j = k - i; if (j >= 0 && j < NL) { acc += (int64_t)a.l[i] * b.l[j]; } }
    }, inj);
    (void)inj_vb; (void)inj_lb;
    C12381_BOUNDS({ check_actual(a, "fp_mul_inj"); check_actual(b, "fp_mul_inj"); set_inj_bounds(t, a.lb * b.lb, a.vb * b.vb, inj_vb, inj_lb, "fp_mul_inj"); })
    r = t;
}
// r = a^2 / R + (injected addends): cross terms once against a doubled copy, like fp_sqr
template <class InjFn>
C12381_HD void fp_sqr_inj(fp& r, const fp& a, InjFn inj, double inj_vb, double inj_lb) {
    C12381_COUNT(27, 0);
    fp t, a2;
    fp_raw_dbl(a2, a);
    fp_reduce_cols_inj(t, [&](int k, int64_t& acc) {
#pragma unroll
        for (int i = 0; i < NL; ++i) { const int j = k - i; if (j > i && j < NL) { acc += (int64_t)a2.l[i] * a.l[j]; } }
        if ((k & 1) == 0 && k / 2 < NL) { acc += (int64_t)a.l[k / 2] * a.l[k / 2]; }
    }, inj);
    (void)inj_vb; (void)inj_lb;
    C12381_BOUNDS({ check_actual(a, "fp_sqr_inj"); set_inj_bounds(t, a.lb * a.lb, a.vb * a.vb, inj_vb, inj_lb, "fp_sqr_inj"); })
    r = t;
}
// 2 a, normalised: one parallel carry round on the doubled limbs (the doubling rides on the shifts: same count as fp_norm1)
C12381_HD void fp_norm1_dbl(fp& r, const fp& a) {
    int32_t c[NL];
#pragma unroll
    for (int i = 0; i < NL - 1; ++i) c[i] = a.l[i] >> (LB - 1);
    const int32_t t13 = 2 * a.l[NL - 1] + c[NL - 2];
#pragma unroll
    for (int i = NL - 2; i >= 1; --i) r.l[i] = C12381_LIMB((int32_t)(((uint32_t)a.l[i] << 1) & LMASK)) + c[i - 1];
    r.l[0] = C12381_LIMB((int32_t)(((uint32_t)a.l[0] << 1) & LMASK));
    r.l[NL - 1] = t13;
    C12381_BOUNDS({ double top = 2 * a.vb * TOP_PER_P + 6.0 + std::floor(2 * a.lb / 268435456.0);
                    double lb = 268435456.0 + std::floor(2 * a.lb / 268435456.0) + 1.0;
                    set_bounds(r, lb > top ? lb : top, 2 * a.vb, "fp_norm1_dbl"); })
}

// r = k0 a + k1 b + k2 c + kp p, carried exactly (limbs 0..12 in [0, 2^28), signed top limb): one 64-bit running sum, four multiply-adds,
// a mask and a shift per limb.  The multipliers are registers (per-lane choices and signs ride on them), so a role-dependent linear
// combination needs no select and its result no carry round.  Value bound declared by the caller (`vb`: what it can prove, e.g. after
// a "- q p" term); limbs of the inputs may be lazy.
C12381_HD void fp_lincomb3p(fp& r, const fp& a, int32_t k0, const fp& b, int32_t k1, const fp& c, int32_t k2, int32_t kp, double vb) {
    int64_t t = 0;
    int32_t out[NL];
#pragma unroll
    for (int i = 0; i < NL - 1; ++i) {
        t += (int64_t)a.l[i] * k0; t += (int64_t)b.l[i] * k1; t += (int64_t)c.l[i] * k2; t += (int64_t)FP_P[i] * kp;
        out[i] = C12381_LIMB((int32_t)((uint32_t)t & LMASK));
        t >>= LB;
    }
    t += (int64_t)a.l[NL - 1] * k0; t += (int64_t)b.l[NL - 1] * k1; t += (int64_t)c.l[NL - 1] * k2; t += (int64_t)FP_P[NL - 1] * kp;
    out[NL - 1] = (int32_t)t;
#pragma unroll
    for (int i = 0; i < NL; ++i) r.l[i] = out[i];
    (void)vb;
    C12381_BOUNDS({ double top = vb * TOP_PER_P + 2.0;
                    set_bounds(r, top > 268435456.0 ? top : 268435456.0, vb, "fp_lincomb3p"); check_actual_vb(r, "fp_lincomb3p value"); })
}

// r = (a*b + c*d) / R  or  (a*b - c*d) / R  with ONE reduction (saves 196 + 14 multiply-adds over two
// fp_mul).  Needs 14*(LBa*LBb + LBc*LBd) + 14*2^56 + 2^40 < 2^63.
template <bool SUB>
C12381_HD void fp_mul2(fp& r, const fp& a, const fp& b, const fp& c, const fp& d);

#ifdef C12381_CHECK_BOUNDS
// declare the bounds of a lazily reduced form: sum_lblb = sum over products of LBa*LBb, sum_vbvb likewise
inline void set_lazy_bounds(fp& r, double sum_lblb, double sum_vbvb, const char* where) {
    double col = 14.0 * sum_lblb + 14.0 * 72057594037927936.0 + 1099511627776.0;
    if (col >= 9223372036854775808.0) bounds_fail(where, sum_lblb, sum_vbvb);
    if (sum_vbvb > 1.0e6) bounds_fail("lazy form value bound", sum_vbvb, 0);
    double vb = sum_vbvb * P_OVER_R + 1.0;
    double top = vb * TOP_PER_P + 2.0;
    set_bounds(r, top > 268435456.0 ? top : 268435456.0, vb, where);
}
// the same for a reduction with injected addends (fp_reduce_cols_inj): sum_vbvb of the product part, inj_abs = a bound on
// |sum_j k_j c_j| / p AS THE CALLER PROVES IT (terms that cancel exactly may be left out: the integer identity is exact),
// inj_lb = sum_j |k_j| LB(c_j) for the column check
inline void set_inj_bounds(fp& r, double sum_lblb, double sum_vbvb, double inj_abs, double inj_lb, const char* where) {
    double col = 14.0 * sum_lblb + 14.0 * 72057594037927936.0 + 1099511627776.0 + inj_lb;
    if (col >= 9223372036854775808.0) bounds_fail(where, sum_lblb, inj_lb);
    if (sum_vbvb > 1.0e6) bounds_fail("injected form value bound", sum_vbvb, 0);
    double vb = sum_vbvb * P_OVER_R + 1.0 + inj_abs;
    double top = vb * TOP_PER_P + 2.0;
    set_bounds(r, top > 268435456.0 ? top : 268435456.0, vb, where);
    check_actual_vb(r, where);               // the caller's proof is checked against the data as well
}
// |value| / p of an element as it stands (test builds: the declared value bound is checked against the data, too)
inline double fp_actual_vb(const fp& a) {
    long double v = 0, w = 1;
    for (int i = 0; i < NL; ++i) { v += (long double)a.l[i] * w; w *= 268435456.0L; }
    long double pp = 0; w = 1;
    for (int i = 0; i < NL; ++i) { pp += (long double)FP_P[i] * w; w *= 268435456.0L; }
    return (double)fabsl(v / pp);
}
inline void check_actual_vb(const fp& a, const char* where) { const double v = fp_actual_vb(a); if (v > a.vb * 1.0000001 + 1e-9) bounds_fail(where, v, a.vb); }
#endif

template <bool SUB>
C12381_HD void fp_mul2(fp& r, const fp& a, const fp& b, const fp& c, const fp& d) {
    fp t, cs;
    if (SUB) fp_raw_neg(cs, c); else cs = c;
    fp_reduce_cols(t, [&](int k, int64_t& acc) { fp_col_acc(acc, a, b, k); fp_col_acc(acc, cs, d, k); });
    C12381_BOUNDS({ check_actual(a, "fp_mul2"); check_actual(b, "fp_mul2"); check_actual(c, "fp_mul2"); check_actual(d, "fp_mul2");
                    set_lazy_bounds(t, a.lb * b.lb + c.lb * d.lb, a.vb * b.vb + c.vb * d.vb, "fp_mul2"); })
    r = t;
}

// ------------------------------------------------------------------ canonical form, tests
// Leaves Montgomery form and fully reduces: r = a / R mod p as canonical limbs in [0, p).
// (FP_redc fp_BLS12381.cpp:234 + FP_reduce :549).  Cost: one fp_mul + compare/subtract.
C12381_HD void fp_from_mont_canonical(fp& r, const fp& a) {
    fp one;
    fp_zero(one);
    one.l[0] = 1;
    C12381_BOUNDS(one.lb = 1; one.vb = 1e-100;)
    fp t;
    fp_mul(t, a, one);            // value in [0, p], limbs normalised, top limb >= 0
    // t == p ?  (only when a is a non-zero multiple of p)
    bool eqp = true;
#pragma unroll
    for (int i = 0; i < NL; ++i) eqp = eqp && (t.l[i] == FP_P[i]);
#pragma unroll
    for (int i = 0; i < NL; ++i) r.l[i] = eqp ? 0 : t.l[i];
    C12381_BOUNDS(r.lb = 268435456.0; r.vb = 1.0;)
}
// a == 0 mod p  (FP_iszilch fp_BLS12381.cpp:305)
C12381_HD bool fp_is_zero(const fp& a) {
    fp t;
    fp_from_mont_canonical(t, a);
    int32_t o = 0;
#pragma unroll
    for (int i = 0; i < NL; ++i) o |= t.l[i];
    return o == 0;
}
C12381_HD bool fp_equal(const fp& a, const fp& b) {
    fp d;
    fp_sub(d, a, b);
    return fp_is_zero(d);
}

// ------------------------------------------------------------------ bytes <-> limbs
// 48 big-endian bytes -> Montgomery form; the integer is taken mod p like FP_nres
// (fp_BLS12381.cpp:223 after BIG_fromBytes big_B384_58.cpp:186).
C12381_HD void fp_from_words_be(fp& r, const uint32_t (&w)[12]) {
    // w[0] is the most significant 32-bit word (already byte-swapped to host order)
    fp t;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const int bit = i * LB;
        const int wi = bit / 32, sh = bit % 32;            // word index from the least significant end
        uint64_t lo = w[11 - wi];
        if (wi + 1 < 12) lo |= (uint64_t)w[11 - (wi + 1)] << 32;
        t.l[i] = C12381_LIMB((int32_t)((uint32_t)(lo >> sh) & LMASK));
    }
    C12381_BOUNDS(t.lb = 268435456.0; t.vb = 10.0;)       // < 2^384 < 10 p
    fp r2;
    fp_set_const(r2, FP_R2);
    fp_mul(r, t, r2);
}
// Montgomery form -> canonical integer as 12 big-endian-ordered 32-bit words (w[0] most significant)
C12381_HD void fp_to_words_be(uint32_t (&w)[12], const fp& a) {
    fp t;
    fp_from_mont_canonical(t, a);
#pragma unroll
    for (int j = 0; j < 12; ++j) {
        const int bit = j * 32;
        const int li = bit / LB, sh = bit % LB;
        uint64_t v = (uint64_t)(uint32_t)t.l[li] >> sh;
        if (li + 1 < NL) v |= (uint64_t)(uint32_t)t.l[li + 1] << (LB - sh);
        if (li + 2 < NL && 2 * LB - sh < 32) v |= (uint64_t)(uint32_t)t.l[li + 2] << (2 * LB - sh);
        w[11 - j] = (uint32_t)v;
    }
}
// parity of the canonical residue (FP_sign fp_BLS12381.cpp:912-936)
C12381_HD int fp_sign(const fp& a) {
    fp t;
    fp_from_mont_canonical(t, a);
    return t.l[0] & 1;
}

// ------------------------------------------------------------------ fixed exponentiations
// r = a^e for a public 384-bit exponent, 4-bit fixed window (exponent is a compile-time
// table in constant memory, so the schedule is identical in every lane).
C12381_HDN void fp_pow_fixed(fp& r, const fp& a, const uint32_t (&e)[12]) {
    fp tab[16];
    fp_one(tab[0]);
    {
        fp n;
        fp_norm1(n, a);
        tab[1] = n;
    }
#pragma unroll 1
    for (int i = 2; i < 16; ++i) fp_mul(tab[i], tab[i - 1], tab[1]);
    fp acc;
    fp_one(acc);
#pragma unroll 1
    for (int wi = 95; wi >= 0; --wi) {
        fp_sqr(acc, acc); fp_sqr(acc, acc); fp_sqr(acc, acc); fp_sqr(acc, acc);
        const uint32_t d = (e[wi / 8] >> (4 * (wi % 8))) & 15u;
        fp_mul(acc, acc, tab[d]);
    }
    r = acc;
}
// Fermat inversion a^(p-2); 0 -> 0 like the reference (FP_inv fp_BLS12381.cpp:817): 380 squarings + 96 products,
// 153 K multiply-adds.  Kept as the independent check of fp_inv (tests/host_sim).
C12381_HD void fp_inv_fermat(fp& r, const fp& a) { fp_pow_fixed(r, a, EXP_P_MINUS_2); }

// ------------------------------------------------------------------ inversion by divsteps
// a^-1 mod p with the constant-time "safegcd" iteration of Bernstein and Yang (eprint 2019/266) in the batched form that
// libsecp256k1's modinv32 popularised: 30 divsteps at a time on the low 32 bits of (f, g) produce a 2x2 transition matrix
// (entries below 2^30), which is then applied to the full-width f, g and — modulo p — to the Bezout coefficients d, e.
// Every lane runs the same instruction stream (masks, no data-dependent branch).  Numbers are 13 signed limbs of 30 bits.
// 879 divsteps suffice for a 381-bit modulus with the delta = 1/2 start (Theorem 11.2: floor((45907 * 381 + 26313) / 19929));
// 30 rounds = 900.  Cost: ~24 K simple instructions + 4 K multiply-adds against 153 K multiply-adds for Fermat.
// 0 -> 0 (g = 0 leaves d = 0), as FP_inv.
constexpr int SG_N = 13;
struct sg30 { int32_t v[SG_N]; };
C12381_HD int32_t sg_divsteps_30(int32_t zeta, uint32_t f0, uint32_t g0, int32_t (&t)[4]) {
    uint32_t u = 1, v = 0, q = 0, r = 1, f = f0, g = g0;
#pragma unroll 1
    for (int i = 0; i < 30; ++i) {
        uint32_t c1 = (uint32_t)(zeta >> 31);                 // all ones iff zeta < 0
        const uint32_t c2 = (uint32_t)0 - (g & 1u);           // all ones iff g odd
        const uint32_t x = (f ^ c1) - c1, y = (u ^ c1) - c1, z = (v ^ c1) - c1;   // conditionally negated f, u, v
        g += x & c2; q += y & c2; r += z & c2;
        c1 &= c2;
        zeta = (int32_t)((uint32_t)zeta ^ c1) - 1;
        f += g & c1; u += q & c1; v += r & c1;
        g >>= 1; u <<= 1; v <<= 1;
    }
    t[0] = (int32_t)u; t[1] = (int32_t)v; t[2] = (int32_t)q; t[3] = (int32_t)r;
    return zeta;
}
// (d, e) <- t * (d, e) / 2^30 mod p; d, e stay in (-2p, p)
C12381_HD void sg_update_de(sg30& d, sg30& e, const int32_t (&t)[4]) {
    constexpr int32_t M30 = (int32_t)0x3fffffff;
    const int32_t u = t[0], v = t[1], q = t[2], r = t[3];
    const int32_t sd = d.v[SG_N - 1] >> 31, se = e.v[SG_N - 1] >> 31;
    int32_t md = (u & sd) + (v & se), me = (q & sd) + (r & se);
    int32_t di = d.v[0], ei = e.v[0];
    int64_t cd = (int64_t)u * di + (int64_t)v * ei;
    int64_t ce = (int64_t)q * di + (int64_t)r * ei;
    md -= (int32_t)((SG_PINV30 * (uint32_t)cd + (uint32_t)md) & (uint32_t)M30);
    me -= (int32_t)((SG_PINV30 * (uint32_t)ce + (uint32_t)me) & (uint32_t)M30);
    cd += (int64_t)SG_P30[0] * md;
    ce += (int64_t)SG_P30[0] * me;
    cd >>= 30; ce >>= 30;                                     // the low 30 bits are zero by construction
#pragma unroll
    for (int i = 1; i < SG_N; ++i) {
        di = d.v[i]; ei = e.v[i];
        cd += (int64_t)u * di + (int64_t)v * ei;
        ce += (int64_t)q * di + (int64_t)r * ei;
        cd += (int64_t)SG_P30[i] * md;
        ce += (int64_t)SG_P30[i] * me;
        d.v[i - 1] = (int32_t)cd & M30; cd >>= 30;
        e.v[i - 1] = (int32_t)ce & M30; ce >>= 30;
    }
    d.v[SG_N - 1] = (int32_t)cd; e.v[SG_N - 1] = (int32_t)ce;
}
// (f, g) <- t * (f, g) / 2^30 (exact)
C12381_HD void sg_update_fg(sg30& f, sg30& g, const int32_t (&t)[4]) {
    constexpr int32_t M30 = (int32_t)0x3fffffff;
    const int32_t u = t[0], v = t[1], q = t[2], r = t[3];
    int32_t fi = f.v[0], gi = g.v[0];
    int64_t cf = (int64_t)u * fi + (int64_t)v * gi;
    int64_t cg = (int64_t)q * fi + (int64_t)r * gi;
    cf >>= 30; cg >>= 30;
#pragma unroll
    for (int i = 1; i < SG_N; ++i) {
        fi = f.v[i]; gi = g.v[i];
        cf += (int64_t)u * fi + (int64_t)v * gi;
        cg += (int64_t)q * fi + (int64_t)r * gi;
        f.v[i - 1] = (int32_t)cf & M30; cf >>= 30;
        g.v[i - 1] = (int32_t)cg & M30; cg >>= 30;
    }
    f.v[SG_N - 1] = (int32_t)cf; g.v[SG_N - 1] = (int32_t)cg;
}
// bring r from (-2p, p) into [0, p), negated first when sign < 0
C12381_HD void sg_normalize(sg30& r, int32_t sign) {
    constexpr int32_t M30 = (int32_t)0x3fffffff;
    int32_t w[SG_N];
#pragma unroll
    for (int i = 0; i < SG_N; ++i) w[i] = r.v[i];
    int32_t cond_add = w[SG_N - 1] >> 31;
#pragma unroll
    for (int i = 0; i < SG_N; ++i) w[i] += SG_P30[i] & cond_add;
    const int32_t cond_negate = sign >> 31;
#pragma unroll
    for (int i = 0; i < SG_N; ++i) w[i] = (w[i] ^ cond_negate) - cond_negate;
#pragma unroll
    for (int i = 0; i < SG_N - 1; ++i) { w[i + 1] += w[i] >> 30; w[i] &= M30; }
    cond_add = w[SG_N - 1] >> 31;
#pragma unroll
    for (int i = 0; i < SG_N; ++i) w[i] += SG_P30[i] & cond_add;
#pragma unroll
    for (int i = 0; i < SG_N - 1; ++i) { w[i + 1] += w[i] >> 30; w[i] &= M30; }
#pragma unroll
    for (int i = 0; i < SG_N; ++i) r.v[i] = w[i];
}
// r = a^-1 (Montgomery form in, Montgomery form out); 0 -> 0
C12381_HDN void fp_inv_divsteps(fp& r, const fp& a) {
    fp c;
    fp_from_mont_canonical(c, a);                             // the integer a / R in [0, p), limbs in [0, 2^28)
    sg30 d, e, f, g;
#pragma unroll
    for (int j = 0; j < SG_N; ++j) {                          // repack 14 x 28 bits -> 13 x 30 bits
        const int bit = 30 * j, li = bit / LB, sh = bit % LB;
        uint64_t v = (uint64_t)(uint32_t)c.l[li] >> sh;
        if (li + 1 < NL) v |= (uint64_t)(uint32_t)c.l[li + 1] << (LB - sh);
        if (li + 2 < NL && 2 * LB - sh < 30) v |= (uint64_t)(uint32_t)c.l[li + 2] << (2 * LB - sh);
        g.v[j] = (int32_t)((uint32_t)v & 0x3fffffffu);
        f.v[j] = SG_P30[j];
        d.v[j] = 0; e.v[j] = 0;
    }
    e.v[0] = 1;
    int32_t zeta = -1;
#pragma unroll 1
    for (int it = 0; it < 30; ++it) {
        int32_t t[4];
        zeta = sg_divsteps_30(zeta, (uint32_t)f.v[0], (uint32_t)g.v[0], t);
        sg_update_de(d, e, t);
        sg_update_fg(f, g, t);
    }
    sg_normalize(d, f.v[SG_N - 1]);                           // f = +-1 (or p when a = 0): d = +-(a / R)^-1
    fp x;
#pragma unroll
    for (int i = 0; i < NL; ++i) {                            // 13 x 30 bits -> 14 x 28 bits
        const int bit = LB * i, li = bit / 30, sh = bit % 30;
        uint64_t v = (uint64_t)(uint32_t)d.v[li] >> sh;
        if (li + 1 < SG_N) v |= (uint64_t)(uint32_t)d.v[li + 1] << (30 - sh);
        x.l[i] = C12381_LIMB((int32_t)((uint32_t)v & LMASK));
    }
    C12381_BOUNDS(x.lb = 268435456.0; x.vb = 1.0;)
    fp r2;
    fp_set_const(r2, FP_R2);
    fp_mul(r, x, r2);                                         // (a / R)^-1 * R^2 / R = a^-1 R
}
C12381_HD void fp_inv(fp& r, const fp& a) {
    fp_inv_divsteps(r, a);
}
// candidate square root a^((p+1)/4) (p = 3 mod 4); caller verifies r^2 == a
C12381_HD void fp_sqrt_candidate(fp& r, const fp& a) { fp_pow_fixed(r, a, EXP_P_PLUS_1_DIV_4); }
// One exponentiation h = a^((p-3)/4) (MIRACL's "progen", FP_progen fp_BLS12381.cpp:782-797) yields all of
//   c = h*a = a^((p+1)/4)  (a square root of a or of -a),   chi = c*h = a^((p-1)/2) = +-1 (0 for a = 0),
//   1/c = chi*h.   Returns is_qr = (chi == 1), i.e. FP_qr :800-813 (0 is reported as a non-residue).
C12381_HD bool fp_sqrt_progen(fp& c, fp& cinv, const fp& a) {
    fp h, an, chi, one, d;
    fp_norm1(an, a);
    fp_pow_fixed(h, an, EXP_P_MINUS_3_DIV_4);
    fp_mul(c, h, an);
    fp_mul(chi, c, h);
    fp_one(one);
    fp_sub(d, chi, one);
    const bool qr = fp_is_zero(d);
    fp nh;
    fp_neg(nh, h);
    fp_select(cinv, qr, h, nh);
    return qr;
}

}  // namespace c12381
