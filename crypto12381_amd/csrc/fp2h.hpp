// HALF an Fp2 element per lane: lanes 2j and 2j+1 of a wavefront hold the real and the imaginary part of the same
// element (role h = lane & 1).  A G2 point is then 3 x 14 dwords per lane instead of 84 — the register footprint of a
// G1 point — and the complete addition of g2.hpp, which spills ~4.7 KB per lane with whole elements, stays in
// registers.  The operations carry the names of fp2.hpp so the templated point arithmetic of g2.hpp serves both types.
//
//   product      re = ar br - ai bi,  im = ar bi + ai br:  every lane forms a*U + pa*V (a = own half, pa = the partner's half
//                of the first operand) with (U, V) = (b, -pb) on the real lane and (pb, b) on the imaginary lane — two
//                products, ONE reduction per lane, the same 4 products + 2 reductions per element as fp2_mul
//   product pair a*b +- c*d: four products, ONE reduction per lane (fp2_mul2)
//   square       complex squaring: (a + pa)(a - pa) on the real lane, (2 pa) a on the imaginary lane — ONE product per
//                lane; operand limb bound 2^28.5 (a normalised value)
//   add/sub/neg  per lane, no communication
//   partner      one DPP move per dword (quad_perm [1,0,3,2]): adjacent lanes swap inside the VALU, no LDS traffic
//
// Every operation is written once as a per-lane routine on (own half, partner half, role).  The device type holds one
// half and fetches the partner by DPP; a host build (tests/host_sim, C12381_CHECK_BOUNDS) holds BOTH halves in one object
// and runs the per-lane routine twice, so the limb / value bounds of exactly these formulas are asserted on the CPU.
#pragma once
#include "fp2.hpp"

namespace c12381 {

// ------------------------------------------------------------------ per-lane routines (role: im = this lane holds the imaginary part)
C12381_HD void fp2h_lane_uv(fp& U, fp& V, const fp& yo, const fp& yp, bool im) {
    fp nyp;
    fp_raw_neg(nyp, yp);
    fp_select(U, im, yp, yo);
    fp_select(V, im, yo, nyp);
}
C12381_HD void fp2h_lane_mul(fp& r, const fp& xo, const fp& xp, const fp& yo, const fp& yp, bool im) {
    fp U, V;
    fp2h_lane_uv(U, V, yo, yp, im);
    fp_mul2<false>(r, xo, U, xp, V);
}
template <bool SUB>
C12381_HD void fp2h_lane_mul2(fp& r, const fp& ao, const fp& ap, const fp& bo, const fp& bp, const fp& co, const fp& cp, const fp& d_o, const fp& dp, bool im) {
    fp U1, V1, U2, V2, cs, cps, t;
    fp2h_lane_uv(U1, V1, bo, bp, im);
    fp2h_lane_uv(U2, V2, d_o, dp, im);
    if (SUB) { fp_raw_neg(cs, co); fp_raw_neg(cps, cp); } else { cs = co; cps = cp; }
    fp_reduce_cols_static(t, [&](int k, int64_t& acc) { fp_col_acc(acc, ao, U1, k); fp_col_acc(acc, ap, V1, k); fp_col_acc(acc, cs, U2, k); fp_col_acc(acc, cps, V2, k); });
    C12381_BOUNDS({ check_actual(ao, "fp2h_mul2"); check_actual(ap, "fp2h_mul2"); check_actual(U1, "fp2h_mul2"); check_actual(V1, "fp2h_mul2");
                    check_actual(cs, "fp2h_mul2"); check_actual(cps, "fp2h_mul2"); check_actual(U2, "fp2h_mul2"); check_actual(V2, "fp2h_mul2");
                    set_lazy_bounds(t, ao.lb * U1.lb + ap.lb * V1.lb + cs.lb * U2.lb + cps.lb * V2.lb,
                                    ao.vb * U1.vb + ap.vb * V1.vb + cs.vb * U2.vb + cps.vb * V2.vb, "fp2h_mul2"); })
    r = t;
}
C12381_HD void fp2h_lane_sqr(fp& r, const fp& xo, const fp& xp, bool im) {
    fp s, d, p2, L, R;
    fp_add(s, xo, xp);
    fp_sub(d, xo, xp);
    fp_raw_dbl(p2, xp);
    fp_select(L, im, p2, s);
    fp_select(R, im, xo, d);
    fp_mul(r, L, R);
}
C12381_HD void fp2h_lane_mul_ip(fp& r, const fp& xo, const fp& xp, bool im) {      // (1 + i) x = (a - b) + (a + b) i
    fp s, d;
    fp_sub(d, xo, xp);
    fp_add(s, xo, xp);
    fp_select(r, im, s, d);
}
C12381_HD void fp2h_lane_conj(fp& r, const fp& xo, bool im) { fp n; fp_neg(n, xo); fp_select(r, im, n, xo); }
// conj(y) * a (1 - i), negated when neg: real lane a (y.a - y.b), imaginary lane -a (y.a + y.b)
C12381_HD void fp2h_lane_conj_mul_a1mi(fp& r, const fp& yo, const fp& yp, const fp& a, bool neg, bool im) {
    fp m, t, nt;
    fp2h_lane_conj(m, yo, im);             // real: y.a ; imaginary: -y.b
    fp_sub(t, m, yp);                      // real: y.a - y.b ; imaginary: -y.b - y.a
    fp_neg(nt, t);
    fp_select(t, neg, nt, t);
    fp_mul(r, t, a);
}

#if defined(__HIPCC__)
// ------------------------------------------------------------------ device type: one half per lane
#define C12381_D __device__ __forceinline__
struct fp2h { fp v; };

__device__ __forceinline__ int fp2h_role() { return (int)(threadIdx.x & 1u); }
__device__ __forceinline__ int32_t pair_swap(int32_t v) { return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true); }
__device__ __forceinline__ void fp_partner(fp& r, const fp& a) {
#pragma unroll
    for (int i = 0; i < NL; ++i) r.l[i] = pair_swap(a.l[i]);
}
__device__ __forceinline__ bool pair_and(bool b) { return b && (pair_swap(b ? 1 : 0) != 0); }

C12381_D void fp2_add(fp2h& r, const fp2h& x, const fp2h& y) { fp_add(r.v, x.v, y.v); }
C12381_D void fp2_sub(fp2h& r, const fp2h& x, const fp2h& y) { fp_sub(r.v, x.v, y.v); }
C12381_D void fp2_neg(fp2h& r, const fp2h& x) { fp_neg(r.v, x.v); }
C12381_D void fp2_dbl(fp2h& r, const fp2h& x) { fp_dbl(r.v, x.v); }
C12381_D void fp2_zero(fp2h& r) { fp_zero(r.v); }
C12381_D void fp2_one(fp2h& r) { fp one, zero; fp_one(one); fp_zero(zero); fp_select(r.v, fp2h_role() == 0, one, zero); }
C12381_D void fp2_norm1(fp2h& r, const fp2h& x) { fp_norm1(r.v, x.v); }
C12381_D void fp2_select(fp2h& r, bool c, const fp2h& x, const fp2h& y) { fp_select(r.v, c, x.v, y.v); }
C12381_D void fp2_mul_small(fp2h& r, const fp2h& x, int32_t k) { fp_mul_small(r.v, x.v, k); }
C12381_D void fp2_conj(fp2h& r, const fp2h& x) { fp2h_lane_conj(r.v, x.v, fp2h_role() != 0); }
C12381_D void fp2_mul_ip(fp2h& r, const fp2h& x) { fp p; fp_partner(p, x.v); fp2h_lane_mul_ip(r.v, x.v, p, fp2h_role() != 0); }
C12381_D bool fp2_is_zero(const fp2h& x) { return pair_and(fp_is_zero(x.v)); }
// r = x * y.  Operand limb bounds as fp2_mul: LBx * LBy <= 2^58.  Output normalised.
C12381_D void fp2_mul(fp2h& r, const fp2h& x, const fp2h& y) {
    fp px, py;
    fp_partner(px, x.v); fp_partner(py, y.v);
    fp2h_lane_mul(r.v, x.v, px, y.v, py, fp2h_role() != 0);
}
template <bool SUB>
C12381_D void fp2_mul2(fp2h& r, const fp2h& a, const fp2h& b, const fp2h& c, const fp2h& d) {
    fp pa, pb, pc, pd;
    fp_partner(pa, a.v); fp_partner(pb, b.v); fp_partner(pc, c.v); fp_partner(pd, d.v);
    fp2h_lane_mul2<SUB>(r.v, a.v, pa, b.v, pb, c.v, pc, d.v, pd, fp2h_role() != 0);
}
C12381_D void fp2_sqr(fp2h& r, const fp2h& x) { fp p; fp_partner(p, x.v); fp2h_lane_sqr(r.v, x.v, p, fp2h_role() != 0); }
C12381_D void fp2_mul_fp(fp2h& r, const fp2h& x, const fp& s) { fp_mul(r.v, x.v, s); }
C12381_D void fp2_conj_mul_ci(fp2h& r, const fp2h& x, const fp& c) { fp p; fp_partner(p, x.v); fp_mul(r.v, p, c); }
C12381_D void fp2_conj_mul_neg_i(fp2h& r, const fp2h& x) { fp p; fp_partner(p, x.v); fp_neg(r.v, p); }
C12381_D void fp2_conj_mul_a1mi(fp2h& r, const fp2h& y, const fp& a, bool neg) {
    fp p; fp_partner(p, y.v);
    fp2h_lane_conj_mul_a1mi(r.v, y.v, p, a, neg, fp2h_role() != 0);
}
C12381_D void fp2_set_const(fp2h& r, const int32_t (&ca)[NL], const int32_t (&cb)[NL]) {
    fp a, b;
    fp_set_const(a, ca); fp_set_const(b, cb);
    fp_select(r.v, fp2h_role() == 0, a, b);
}
// this lane's half of a whole element / the whole element from the two halves
C12381_D void fp2h_from(fp2h& r, const fp2& x) { fp_select(r.v, fp2h_role() == 0, x.a, x.b); }

#else
// ------------------------------------------------------------------ host emulation: both halves in one object, the per-lane
// routines run once per role (h[0] = real lane, h[1] = imaginary lane)
struct fp2h { fp h[2]; };

inline void fp2_add(fp2h& r, const fp2h& x, const fp2h& y) { fp_add(r.h[0], x.h[0], y.h[0]); fp_add(r.h[1], x.h[1], y.h[1]); }
inline void fp2_sub(fp2h& r, const fp2h& x, const fp2h& y) { fp_sub(r.h[0], x.h[0], y.h[0]); fp_sub(r.h[1], x.h[1], y.h[1]); }
inline void fp2_neg(fp2h& r, const fp2h& x) { fp_neg(r.h[0], x.h[0]); fp_neg(r.h[1], x.h[1]); }
inline void fp2_dbl(fp2h& r, const fp2h& x) { fp_dbl(r.h[0], x.h[0]); fp_dbl(r.h[1], x.h[1]); }
inline void fp2_zero(fp2h& r) { fp_zero(r.h[0]); fp_zero(r.h[1]); }
inline void fp2_one(fp2h& r) { fp_one(r.h[0]); fp_zero(r.h[1]); }
inline void fp2_norm1(fp2h& r, const fp2h& x) { fp_norm1(r.h[0], x.h[0]); fp_norm1(r.h[1], x.h[1]); }
inline void fp2_select(fp2h& r, bool c, const fp2h& x, const fp2h& y) { fp_select(r.h[0], c, x.h[0], y.h[0]); fp_select(r.h[1], c, x.h[1], y.h[1]); }
inline void fp2_mul_small(fp2h& r, const fp2h& x, int32_t k) { fp_mul_small(r.h[0], x.h[0], k); fp_mul_small(r.h[1], x.h[1], k); }
inline void fp2_conj(fp2h& r, const fp2h& x) { fp2h t; fp2h_lane_conj(t.h[0], x.h[0], false); fp2h_lane_conj(t.h[1], x.h[1], true); r = t; }
inline void fp2_mul_ip(fp2h& r, const fp2h& x) { fp2h t; fp2h_lane_mul_ip(t.h[0], x.h[0], x.h[1], false); fp2h_lane_mul_ip(t.h[1], x.h[1], x.h[0], true); r = t; }
inline bool fp2_is_zero(const fp2h& x) { return fp_is_zero(x.h[0]) & fp_is_zero(x.h[1]); }
inline void fp2_mul(fp2h& r, const fp2h& x, const fp2h& y) {
    fp2h t;
    fp2h_lane_mul(t.h[0], x.h[0], x.h[1], y.h[0], y.h[1], false);
    fp2h_lane_mul(t.h[1], x.h[1], x.h[0], y.h[1], y.h[0], true);
    r = t;
}
template <bool SUB>
inline void fp2_mul2(fp2h& r, const fp2h& a, const fp2h& b, const fp2h& c, const fp2h& d) {
    fp2h t;
    fp2h_lane_mul2<SUB>(t.h[0], a.h[0], a.h[1], b.h[0], b.h[1], c.h[0], c.h[1], d.h[0], d.h[1], false);
    fp2h_lane_mul2<SUB>(t.h[1], a.h[1], a.h[0], b.h[1], b.h[0], c.h[1], c.h[0], d.h[1], d.h[0], true);
    r = t;
}
inline void fp2_sqr(fp2h& r, const fp2h& x) { fp2h t; fp2h_lane_sqr(t.h[0], x.h[0], x.h[1], false); fp2h_lane_sqr(t.h[1], x.h[1], x.h[0], true); r = t; }
inline void fp2_mul_fp(fp2h& r, const fp2h& x, const fp& s) { fp_mul(r.h[0], x.h[0], s); fp_mul(r.h[1], x.h[1], s); }
inline void fp2_conj_mul_ci(fp2h& r, const fp2h& x, const fp& c) { fp2h t; fp_mul(t.h[0], x.h[1], c); fp_mul(t.h[1], x.h[0], c); r = t; }
inline void fp2_conj_mul_neg_i(fp2h& r, const fp2h& x) { fp2h t; fp_neg(t.h[0], x.h[1]); fp_neg(t.h[1], x.h[0]); r = t; }
inline void fp2_conj_mul_a1mi(fp2h& r, const fp2h& y, const fp& a, bool neg) {
    fp2h t;
    fp2h_lane_conj_mul_a1mi(t.h[0], y.h[0], y.h[1], a, neg, false);
    fp2h_lane_conj_mul_a1mi(t.h[1], y.h[1], y.h[0], a, neg, true);
    r = t;
}
inline void fp2_set_const(fp2h& r, const int32_t (&ca)[NL], const int32_t (&cb)[NL]) { fp_set_const(r.h[0], ca); fp_set_const(r.h[1], cb); }
inline void fp2h_from(fp2h& r, const fp2& x) { r.h[0] = x.a; r.h[1] = x.b; }
inline void fp2h_to(fp2& r, const fp2h& x) { r.a = x.h[0]; r.b = x.h[1]; }
#endif

}  // namespace c12381
