// HALF an Fp2 element per lane: lanes 2j and 2j+1 of a wavefront hold the real and the imaginary part of the same
// element (role h = lane & 1).  A G2 point is then 3 x 14 dwords per lane instead of 84 — the register footprint of a
// G1 point — and the complete addition of g2.hpp, which spills ~4.7 KB per lane with whole elements, stays in
// registers.  The operations carry the names of fp2.hpp so the templated point arithmetic of g2.hpp serves both types.
//
//   product      re = ar br - ai bi,  im = ar bi + ai br:  every lane forms X*b + Y*pb (pb = the partner's half of b)
//                with (X, Y) = (a, -pa) on the real lane and (pa, a) on the imaginary lane — two products, ONE
//                reduction per lane (fp_mul2), the same 4 products + 2 reductions per element as fp2_mul
//   square       the same form with b = a (a uniform instruction stream cannot give the two roles different shapes)
//   add/sub/neg  per lane, no communication
//   partner      one DPP move per dword (quad_perm [1,0,3,2]): adjacent lanes swap inside the VALU, no LDS traffic
// Device functions only (the host simulation checks the limb bounds of the whole-element forms; the bounds here are the same:
// a product is two limb products per lane and reduction, as in fp2_mul).
#pragma once
#include "fp2.hpp"

#if defined(__HIPCC__)
#define C12381_D __device__ __forceinline__
namespace c12381 {

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
C12381_D void fp2_conj(fp2h& r, const fp2h& x) { fp n; fp_neg(n, x.v); fp_select(r.v, fp2h_role() == 0, x.v, n); }
// (1 + i) x = (a - b) + (a + b) i — lazy, limb bound doubles
C12381_D void fp2_mul_ip(fp2h& r, const fp2h& x) {
    fp p, s, d;
    fp_partner(p, x.v);
    fp_sub(d, x.v, p);                 // real lane: a - b
    fp_add(s, x.v, p);                 // imaginary lane: b + a
    fp_select(r.v, fp2h_role() == 0, d, s);
}
C12381_D bool fp2_is_zero(const fp2h& x) { return pair_and(fp_is_zero(x.v)); }
// r = x * y.  Operand limb bounds as fp2_mul: LBx * LBy <= 2^58.  Output normalised.
C12381_D void fp2_mul(fp2h& r, const fp2h& x, const fp2h& y) {
    fp px, py, npx, X, Y;
    fp_partner(px, x.v);
    fp_partner(py, y.v);
    fp_raw_neg(npx, px);
    const bool im = fp2h_role() != 0;
    fp_select(X, im, px, x.v);
    fp_select(Y, im, x.v, npx);
    fp_mul2<false>(r.v, X, y.v, Y, py);
}
C12381_D void fp2_sqr(fp2h& r, const fp2h& x) { fp2_mul(r, x, x); }
C12381_D void fp2_mul_fp(fp2h& r, const fp2h& x, const fp& s) { fp_mul(r.v, x.v, s); }
C12381_D void fp2_set_const(fp2h& r, const int32_t (&ca)[NL], const int32_t (&cb)[NL]) {
    fp a, b;
    fp_set_const(a, ca); fp_set_const(b, cb);
    fp_select(r.v, fp2h_role() == 0, a, b);
}
// this lane's half of a whole element / the whole element from the two halves
C12381_D void fp2h_from(fp2h& r, const fp2& x) { fp_select(r.v, fp2h_role() == 0, x.a, x.b); }

}  // namespace c12381
#endif
