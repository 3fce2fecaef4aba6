// G2 arithmetic on the M-type sextic twist y^2 = x^3 + 4(1+i) over Fp2
// (replaces ECP2_dbl / ECP2_add / ECP2_affine / ECP2_toOctet ecp2_BLS12381.cpp:358-502, 109-133,
//  184-220 and PAIR_G2mul pair_BLS12381.cpp:927-983 of the reference).
// Same design as g1.hpp: one point per lane, complete projective formulas, no divergent branch.
// Bound contract: stored coordinates carry limb bound <= 2^29; fp2_mul needs LBx*LBy <= 2^58.
#pragma once
#include "fp2.hpp"
#include "g1.hpp"

namespace c12381 {

// The point arithmetic below is written once for two element types: fp2 (one point per lane) and fp2h (fp2h.hpp: HALF an
// Fp2 element per lane, two adjacent lanes per point — 42 dwords per point and lane instead of 84, so the complete
// addition fits the register file).  Both provide the same fp2_* operations.
template <class F> struct g2pt { F x, y, z; };     // (X:Y:Z), infinity = (0:1:0)
using g2p = g2pt<fp2>;

template <class F> C12381_HD void g2_set_inf(g2pt<F>& p) { fp2_zero(p.x); fp2_one(p.y); fp2_zero(p.z); }
template <class F> C12381_HD void g2_norm1(g2pt<F>& r, const g2pt<F>& p) { fp2_norm1(r.x, p.x); fp2_norm1(r.y, p.y); fp2_norm1(r.z, p.z); }
template <class F> C12381_HD void g2_neg(g2pt<F>& r, const g2pt<F>& p) { r.x = p.x; fp2_neg(r.y, p.y); r.z = p.z; }
// 3b' = 12 (1 + i): small multiply (normalising) then the lazy (1+i) map — ecp2_BLS12381.cpp:381-385
template <class F> C12381_HD void fp2_mul_b3(F& r, const F& x) { F t; fp2_mul_small(t, x, 12); fp2_mul_ip(r, t); }

// ECP2_dbl :358-409.  Also returns t0 = Y^2, t1 = Y*Z, t2b = 3b' Z^2 for the Miller-loop line.
// Operand limb bounds: X <= 2^29, Y and Z normalised (every Y / Z this file produces is a reduction output).
// Y3 = (Y^2 - 9b'Z^2)(Y^2 + 3b'Z^2) + 3b'Z^2 * 8Y^2 is ONE lazily reduced form (fp2_mul2): 7 reductions for 8 products.
template <class F> C12381_HD void g2_dbl_core(g2pt<F>& p, F& t0, F& t1, F& t2b) {
    F t2, x3, y3, z3, z8, u, s;
    fp2_sqr(t0, p.y);
    fp2_mul(t1, p.z, p.y);                        // second operands p.y (twice), z8 (twice), u (twice): prepared once each in the two-lane form
    fp2_sqr(t2, p.z);
    fp2_mul_small(z8, t0, 8);
    fp2_mul_b3(t2b, t2);
    fp2_add(s, t0, t2b);
    fp2_norm1(s, s);                              // Y^2 + 3b'Z^2
    fp2_mul(z3, t1, z8);
    fp2_dbl(u, t2b); fp2_add(u, u, t2b);          // 9b' Z^2
    fp2_sub(u, t0, u);
    fp2_norm1(u, u);
    fp2_mul2<false>(y3, s, u, t2b, z8);
    F xy;
    fp2_mul(xy, p.x, p.y);
    fp2_mul(x3, xy, u);
    fp2_dbl(x3, x3);
    p.x = x3; p.y = y3; p.z = z3;
}
template <class F> C12381_HDN void g2_dbl_ex(g2pt<F>& p, F& t0, F& t1, F& t2b) { g2_dbl_core(p, t0, t1, t2b); }
template <class F> C12381_HDN void g2_dbl(g2pt<F>& p) { F a, b, c; g2_dbl_core(p, a, b, c); }
// n successive doublings with the point held in registers (one load and one store of the 84 dwords per call)
template <class F> C12381_HDN void g2_dbl_n(g2pt<F>& p, int n) {
    g2pt<F> q = p;
#pragma unroll 1
    for (int i = 0; i < n; ++i) { F a, b, c; g2_dbl_core(q, a, b, c); }
    p = q;
}

// ECP2_add :413-502 (complete).  P: X <= 2^29, Y and Z normalised; Q normalised.  The three output coordinates are sums of two
// products each and are reduced once (fp2_mul2): 9 reductions for 12 products, as g1_add.
template <class F> C12381_HD void g2_add_core(g2pt<F>& p, const g2pt<F>& q) {
    F t0, t1, t2, t3, t4, x3, y3, z3;
    fp2_mul(t0, p.x, q.x);
    fp2_mul(t1, p.y, q.y);
    fp2_mul(t2, p.z, q.z);
    fp2_add(t3, p.x, p.y); fp2_norm1(t3, t3); fp2_add(t4, q.x, q.y); fp2_mul(t3, t3, t4);
    fp2_add(t4, t0, t1); fp2_sub(t3, t3, t4); fp2_norm1(t3, t3);
    fp2_add(t4, p.y, p.z); fp2_add(x3, q.y, q.z); fp2_mul(t4, t4, x3);
    fp2_add(x3, t1, t2); fp2_sub(t4, t4, x3); fp2_norm1(t4, t4);
    fp2_add(x3, p.x, p.z); fp2_norm1(x3, x3); fp2_add(y3, q.x, q.z); fp2_mul(x3, x3, y3);
    fp2_add(y3, t0, t2); fp2_sub(y3, x3, y3);
    fp2_mul_small(t0, t0, 3);
    fp2_mul_b3(t2, t2);
    fp2_add(z3, t1, t2); fp2_norm1(z3, z3);
    fp2_sub(t1, t1, t2); fp2_norm1(t1, t1);
    fp2_mul_b3(y3, y3);
    // (second operands t1, t4 | t0, t1 | t4, t0: the two-lane form prepares a second operand once — fp2h_lane_uv — and reuses it)
    fp2_mul2<true>(p.x, t3, t1, y3, t4);           // X3 = t3 t1 - y3 t4
    fp2_mul2<false>(p.y, y3, t0, z3, t1);          // Y3 = y3 t0 + z3 t1
    fp2_mul2<false>(p.z, z3, t4, t3, t0);          // Z3 = z3 t4 + t3 t0
}
template <class F> C12381_HDN void g2_add(g2pt<F>& p, const g2pt<F>& q) { g2_add_core(p, q); }
// element types whose scalar-multiplication loop keeps the running point in registers: doublings and additions inlined into the
// window loop (like g1_scalar_mul) instead of out-of-line routines that take the point through private memory.  Pays where the
// whole addition fits the register file — the two-lane form (fp2h.hpp specialises this to true).
template <class F> struct g2_inline_loop { static constexpr bool value = false; };

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

// ------------------------------------------------------------------ per-lane window table (one 2688-byte record)
constexpr int G2_WIN = 5;                                // 4 or 5 (signed windows over the four 64-bit GS digits); A/B on MI355X: profiles/r03_ab_g2_window5.txt
static_assert(G2_WIN == 4 || G2_WIN == 5, "window width");
constexpr int G2_TAB = 1 << (G2_WIN - 1);               // entries 1..8 (1..16)
constexpr int G2_WINDOWS = G2_WIN == 4 ? 17 : 13;       // 4: 16 biased nibbles + the carry nibble; 5: 13 biased fields cover 65 bits
constexpr int G2_ENT_DWORDS = 6 * NL;                   // 84 dwords = 21 16-byte accesses
constexpr int G2_TAB_DWORDS = G2_TAB * G2_ENT_DWORDS;   // per lane (one-lane form): 2688 B (4-bit) / 5376 B (5-bit)
C12381_HD constexpr int g2_ent_dwords(const g2p&) { return G2_ENT_DWORDS; }

C12381_HD void tab_store_g2(int32_t* ent, const g2p& p) {
    const fp* c[6] = {&p.x.a, &p.x.b, &p.y.a, &p.y.b, &p.z.a, &p.z.b};
    int32_t w[G2_ENT_DWORDS];
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int i = 0; i < NL; ++i) w[j * NL + i] = c[j]->l[i];
    q4* dst = reinterpret_cast<q4*>(ent);
#pragma unroll
    for (int i = 0; i < G2_ENT_DWORDS / 4; ++i) { q4 t; t.v[0] = w[4 * i]; t.v[1] = w[4 * i + 1]; t.v[2] = w[4 * i + 2]; t.v[3] = w[4 * i + 3]; dst[i] = t; }
}
C12381_HD void tab_load_g2(g2p& p, const int32_t* ent) {
    int32_t w[G2_ENT_DWORDS];
    const q4* src = reinterpret_cast<const q4*>(ent);
#pragma unroll
    for (int i = 0; i < G2_ENT_DWORDS / 4; ++i) { q4 t = src[i]; w[4 * i] = t.v[0]; w[4 * i + 1] = t.v[1]; w[4 * i + 2] = t.v[2]; w[4 * i + 3] = t.v[3]; }
    fp* c[6] = {&p.x.a, &p.x.b, &p.y.a, &p.y.b, &p.z.a, &p.z.b};
#pragma unroll
    for (int j = 0; j < 6; ++j) {
#pragma unroll
        for (int i = 0; i < NL; ++i) c[j]->l[i] = w[j * NL + i];
        C12381_BOUNDS(c[j]->lb = 268435456.0 + 8.0; c[j]->vb = 4.0; check_actual(*c[j], "tab_load_g2");)
    }
}

// psi^I(X,Y,Z) = (conj^I(X) c_x, conj^I(Y) c_y, conj^I(Z))   (ECP2_frob ecp2_BLS12381.cpp:579-590 applied I times with
// X = 1/f as PAIR_G2mul does for the M-type twist, pair_BLS12381.cpp:944-947).  psi is a group homomorphism of the
// twist, so psi^I(d Q) = d psi^I(Q): one table of multiples of Q serves all four sub-scalars.
// The constants have special shapes (fp2.hpp; tools/gen_consts.py asserts them): c_x = c i | N(c) in Fp | -i and
// c_y = a(1 - i) | -1 | -a(1 - i) for I = 1 | 2 | 3, so psi^1 costs four Fp products, psi^2 and psi^3 two each (the generic
// form: two Fp2 products = eight Fp half-products).  r = (-1)^neg psi^I(p): the sign rides on the Y coordinate for free.
template <int I, class F>
C12381_HD void g2_psi_signed(g2pt<F>& r, const g2pt<F>& p, bool neg) {
    if (I == 0 || I == 2) {
        F ny;
        fp2_neg(ny, p.y);
        fp2_select(r.y, neg != (I == 2), ny, p.y);                    // psi^2: c_y = -1
        r.z = p.z;
        if (I == 0) { r.x = p.x; return; }
        fp cx;
        fp_set_const(cx, PSI2_X);
        fp2_mul_fp(r.x, p.x, cx);
        return;
    }
    fp cy;
    fp_set_const(cy, PSI1_Y_A);
    fp2_conj_mul_a1mi(r.y, p.y, cy, neg != (I == 3));                 // psi^3: c_y = -a(1 - i)
    if (I == 1) { fp cx; fp_set_const(cx, PSI1_X_B); fp2_conj_mul_ci(r.x, p.x, cx); }
    else fp2_conj_mul_neg_i(r.x, p.x);
    fp2_conj(r.z, p.z);
}
template <int I, class F> C12381_HD void g2_psi(g2pt<F>& r, const g2pt<F>& p) { g2_psi_signed<I>(r, p, false); }

// 128-by-64-bit division step for the normalised 64-bit constant |x| (top bit set) with its precomputed reciprocal
// v = floor((2^128 - 1) / |x|) - 2^64 (Moeller-Granlund, "Improved division by invariant integers", algorithm 4):
// (u1, u0) with u1 < |x|  ->  quotient (one word), remainder in u1.
C12381_HD uint64_t mulhi64(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(a, b);
#else
    return (uint64_t)(((unsigned __int128)a * b) >> 64);
#endif
}
C12381_HD uint64_t div_by_absx(uint64_t& u1, uint64_t u0) {
    uint64_t q0 = BLS_X_RECIP * u1, q1 = mulhi64(BLS_X_RECIP, u1);
    q0 += u0; q1 += u1 + (q0 < u0 ? 1u : 0u);                 // (q1, q0) = v u1 + (u1, u0)
    q1 += 1;
    uint64_t r = u0 - q1 * BLS_X;
    if (r > q0) { q1 -= 1; r += BLS_X; }
    if (r >= BLS_X) { q1 += 1; r -= BLS_X; }
    u1 = r;
    return q1;
}
// k (< r, 8 words) -> base-|x| digits u0..u3 (each < 2^64): k = u0 + u1|x| + u2|x|^2 + u3|x|^3   (gs() pair_BLS12381.cpp:814-873)
// Three long divisions by the one-word constant |x|, word by word from the top (nine division steps in all).
C12381_HD void scalar_gs_split(uint32_t (&u)[4][2], const uint32_t (&k)[8]) {
    uint64_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = (uint64_t)k[2 * i] | ((uint64_t)k[2 * i + 1] << 32);
#pragma unroll
    for (int lvl = 0; lvl < 3; ++lvl) {
        uint64_t rem = 0;
#pragma unroll
        for (int i = 3 - lvl; i >= 0; --i) w[i] = div_by_absx(rem, w[i]);      // quotient of level lvl has 4 - lvl - 1 words (+ a zero top word)
        u[lvl][0] = (uint32_t)rem; u[lvl][1] = (uint32_t)(rem >> 32);
    }
    u[3][0] = (uint32_t)w[0]; u[3][1] = (uint32_t)(w[0] >> 32);
}
// signed digit of window w of the biased digit u':  4-bit windows: u' = u + 0x8888888888888888, d = nibble - 8 in [-8, 7], window 16
// is the carry nibble (0 or 1);  5-bit windows: u' = u + sum_{w<13} 16 * 32^w < 2^65 (u < |x| < 0.83 * 2^64), d = field - 16 in
// [-16, 15], no carry window.  sum_w d_w 2^(WIN w) = u.
constexpr uint32_t gs_bias_word5(int i) {
    uint32_t v = 0;
    for (int w = 0; w < 13; ++w) { const int bit = 5 * w + 4; if ((bit >> 5) == i) v |= 1u << (bit & 31); }
    return v;
}
C12381_HD int gs_digit(const uint32_t (&ub)[3], int w) {
    if (G2_WIN == 4) {
        const int nib = (int)((ub[w >> 3] >> ((w & 7) * 4)) & 15u);
        return w == 16 ? nib : nib - 8;
    }
    const int bit = 5 * w, word = bit >> 5, sh = bit & 31;
    uint32_t v = ub[word] >> sh;
    if (sh > 27 && word < 2) v |= ub[word + 1] << (32 - sh);
    return (int)(v & 31u) - 16;
}
C12381_HD void gs_bias(uint32_t (&ub)[3], const uint32_t (&u)[2]) {
    const uint32_t b0 = G2_WIN == 4 ? 0x88888888u : gs_bias_word5(0), b1 = G2_WIN == 4 ? 0x88888888u : gs_bias_word5(1);
    uint64_t c = (uint64_t)u[0] + b0; ub[0] = (uint32_t)c; c >>= 32;
    c += (uint64_t)u[1] + b1; ub[1] = (uint32_t)c; ub[2] = (uint32_t)(c >> 32) + (G2_WIN == 4 ? 0u : gs_bias_word5(2));
}
// e = (-1)^I sign(d) psi^I(T[|d|]); d == 0 gives the point at infinity
// the table record a digit selects (|d| = 0 reads entry 1 and is replaced by the point at infinity afterwards)
template <class F>
C12381_HD void g2_digit_load(g2pt<F>& q, const int32_t* lane_tab, int d) {
    const int mag = d < 0 ? -d : d;
    tab_load_g2(q, lane_tab + ((mag == 0 ? 1 : mag) - 1) * g2_ent_dwords(q));
}
template <int I, class F> C12381_HD void g2_digit_fix(g2pt<F>& e, const g2pt<F>& q, int d);
template <int I, class F>
C12381_HD void g2_digit_point(g2pt<F>& e, const int32_t* lane_tab, int d) {
    g2pt<F> q;
    g2_digit_load(q, lane_tab, d);
    g2_digit_fix<I>(e, q, d);
}
template <int I, class F>
C12381_HD void g2_digit_fix(g2pt<F>& e, const g2pt<F>& q, int d) {
    const int mag = d < 0 ? -d : d;
    g2pt<F> inf;
    g2_psi_signed<I>(e, q, (d < 0) != ((I & 1) != 0));
    g2_set_inf(inf);
    const bool isz = mag == 0;
    fp2_select(e.x, isz, inf.x, e.x); fp2_select(e.y, isz, inf.y, e.y); fp2_select(e.z, isz, inf.z, e.z);
}
// acc += (-1)^I sign(d) psi^I(T[|d|])
template <int I, class F>
C12381_HD void g2_add_digit(g2pt<F>& acc, const int32_t* lane_tab, int d) {
    constexpr bool INL = g2_inline_loop<F>::value;
    g2pt<F> e;
    g2_digit_point<I>(e, lane_tab, d);
    if (INL) g2_add_core(acc, e); else g2_add(acc, e);
}

template <class F> C12381_HD bool g2_is_inf(const g2pt<F>& p) { return fp2_is_zero(p.z); }
// p <- [|x|]p, plain double-and-add over the 64-bit curve parameter
template <class F> C12381_HDN void g2_mul_absx(g2pt<F>& p) {
    g2pt<F> base, acc;
    g2_norm1(base, p);
    acc = base;
#pragma unroll 1
    for (int i = 62; i >= 0; --i) {
        g2_dbl(acc);
        if ((BLS_X_W[i >> 5] >> (i & 31)) & 1u) g2_add(acc, base);
    }
    p = acc;
}
// gs() negates the odd digits mod r (pair_BLS12381.cpp:868-871) and BIG_modneg(0) = r; PAIR_G2mul's sign minimisation
// (:962-971) undoes the negation for non-zero digits only, so a ZERO odd digit u_i makes ECP2_mul4 add [r]psi^i(Q):
// infinity for Q in G2, a point of the cofactor part otherwise.  Reproduced here (lanes with u1 == 0 or u3 == 0 only,
// a divergent branch) so results agree with `multiply` on every point of the twist.
// Q is in G2 iff psi(Q) = [x]Q = -[|x|]Q; off the subgroup [r]Q = [x^4]Q - [x^2]Q + Q and [r]psi^i(Q) = psi^i([r]Q).
template <class F> C12381_HDN void g2_gs_zero_digit_terms(g2pt<F>& acc, const g2pt<F>& base, bool z1, bool z3) {
    g2pt<F> q, a1, a2, a4, t;
    g2_norm1(q, base);
    a1 = q;
    g2_mul_absx(a1);
    g2_norm1(a1, a1);
    g2_psi<1>(t, q);
    g2_norm1(t, t);
    {
        g2pt<F> u = a1;
        g2_add(u, t);
        if (g2_is_inf(u)) return;                  // Q in G2: the extra terms vanish
    }
    a2 = a1;
    g2_mul_absx(a2);
    g2_norm1(a2, a2);
    a4 = a2;
    g2_mul_absx(a4); g2_mul_absx(a4);
    g2_neg(t, a2);
    g2_norm1(t, t);
    g2_add(a4, t);
    g2_norm1(a4, a4);
    g2_add(a4, q);
    g2_norm1(a4, a4);                              // [r]Q
    g2_norm1(acc, acc);
    if (z1) { g2_psi<1>(t, a4); g2_norm1(t, t); g2_add(acc, t); g2_norm1(acc, acc); }
    if (z3) { g2_psi<3>(t, a4); g2_norm1(t, t); g2_add(acc, t); }
}

// PAIR_G2mul pair_BLS12381.cpp:927-983: R = u0 Q - u1 psi(Q) + u2 psi^2(Q) - u3 psi^3(Q) for the base-|x| digits of
// k mod r — exactly what the reference evaluates (ECP2_mul4 after gs() and the sign minimisation), on ANY point of
// the twist; for Q in G2 it equals [k]Q.  Signed 5-bit windows: 60 doublings + 52 additions on one 16-entry table of multiples of Q
// (8 doublings + 7 additions to build; 4-bit windows: 64 + 68 on 8 entries, 4 + 3).
// in_g2: the caller asserts Q lies in G2 (C12381_F_IN_SUBGROUP) — the [r]psi^i(Q) terms are then the point at infinity and
// their evaluation (a membership test of 64 doublings per affected lane) is skipped.
template <class F> C12381_HDN void g2_scalar_mul(g2pt<F>& acc, const F& qx, const F& qy, bool q_is_inf, const uint32_t (&kin)[8], int32_t* lane_tab, bool in_g2 = false) {
    uint32_t k[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) k[i] = kin[i];
    scalar_mod_r(k);
    uint32_t u[4][2], ub[4][3];
    scalar_gs_split(u, k);
#pragma unroll
    for (int i = 0; i < 4; ++i) gs_bias(ub[i], u[i]);
    g2pt<F> base, t;
    g2_set_inf(t);
    base.x = qx; base.y = qy; fp2_one(base.z);
    fp2_select(base.x, q_is_inf, t.x, base.x);
    fp2_select(base.y, q_is_inf, t.y, base.y);
    fp2_select(base.z, q_is_inf, t.z, base.z);
    tab_store_g2(lane_tab, base);
    t = base;
    g2_dbl(t);
    {
        g2pt<F> n;
        g2_norm1(n, t);
        tab_store_g2(lane_tab + g2_ent_dwords(n), n);
        t = n;
    }
    // even multiples by doubling the entry half as large, odd ones by adding Q to the previous entry (as g1_scalar_mul)
#pragma unroll 1
    for (int j = 3; j <= G2_TAB; ++j) {
        if ((j & 1) == 0) {                                     // wave-uniform
            tab_load_g2(t, lane_tab + (j / 2 - 1) * g2_ent_dwords(t));
            g2_dbl(t);
        } else {
            g2_add(t, base);
        }
        g2pt<F> n;
        g2_norm1(n, t);
        tab_store_g2(lane_tab + (j - 1) * g2_ent_dwords(n), n);
        t = n;
    }
    // the running point is a LOCAL of this loop: `acc` is a reference into the caller's memory, and a table load may alias it as far as
    // the compiler knows — through the reference every doubling and addition stored the point back (19 KB of private-memory writes per
    // lane and scalar multiplication, 6 GB per launch: profiles/r03_pmc_summary.json)
    g2pt<F> run;
    g2_set_inf(run);
#pragma unroll 1
    for (int w = G2_WINDOWS - 1; w >= 0; --w) {
        if (w != G2_WINDOWS - 1) {
            if (g2_inline_loop<F>::value) {
#pragma unroll 1
                for (int i = 0; i < G2_WIN; ++i) { F a, b, c; g2_dbl_core(run, a, b, c); }
            } else g2_dbl_n(run, G2_WIN);
            g2_add_digit<0>(run, lane_tab, gs_digit(ub[0], w));
        } else {
            g2_digit_point<0>(run, lane_tab, gs_digit(ub[0], w));       // the top window starts the accumulator: infinity + entry = entry
        }
        g2_add_digit<1>(run, lane_tab, gs_digit(ub[1], w));
        g2_add_digit<2>(run, lane_tab, gs_digit(ub[2], w));
        g2_add_digit<3>(run, lane_tab, gs_digit(ub[3], w));
    }
    acc = run;
    const bool z1 = (u[1][0] | u[1][1]) == 0u, z3 = (u[3][0] | u[3][1]) == 0u;
    if ((z1 || z3) && !in_g2) g2_gs_zero_digit_terms(acc, base, z1, z3);
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
