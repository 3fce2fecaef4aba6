// HIP kernels + C ABI of the batched BLS12-381 backend for MI355X (gfx950).
// Public interface and reference citations: include/c12381_hip.h.
//
// Kernel inventory (one element per lane everywhere):
//   fp_op_kernel / fp_mulchain_kernel   Fp test + VALU-roofline hook
//   g1_mul_kernel      bytes -> on-curve check -> GLV windowed [k]P -> projective SoA in HBM
//   g1_add_kernel      complete addition of two affine inputs -> projective SoA
//   g1_finish_kernel   Montgomery's simultaneous inversion over a strided chunk per lane,
//                      affine conversion, canonical encoding (49 B / 96 B)
//   g1_reduce_kernel   tree sum of projective points (MSM combine)
//   g2_mul_kernel      bytes -> on-twist check -> windowed [k]Q -> affine -> 97 B / 192 B
//   g2_add_kernel      complete addition of two affine G2 inputs
//   pair_kernel        Miller loop + final exponentiation -> 576-byte GT
//   pair_eq_kernel     e(a1,a2) == e(b1,b2): two Miller loops, ONE final exponentiation, is-unity
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/c12381_hip.h"
#include "codec.hpp"
#include "fp.hpp"
#include "g1.hpp"
#include "g2.hpp"
#include "pairing.hpp"
#include "pairing3.hpp"
#include "msm.hpp"

using namespace c12381;

namespace {

constexpr int BLOCK = 256;
constexpr size_t G1_CHUNK = (size_t)1 << 17;     // elements per scalar-mul launch = resident lanes at 2 waves/SIMD; table slab 176 MiB (fits the 256 MiB Infinity Cache)
constexpr size_t G2_CHUNK = (size_t)1 << 17;     // G2 table slab = 352 MiB (2688-byte record per lane)
constexpr int FINISH_M = 16;                     // elements per lane in the simultaneous inversion

// ------------------------------------------------------------------ device helpers
__device__ __forceinline__ void load_raw48(uint32_t* w, const uint8_t* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
#pragma unroll
    for (int i = 0; i < 3; ++i) { uint4 v = q[i]; w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w; }
}
__device__ __forceinline__ void store_raw48(uint8_t* p, const uint32_t* w) {
    uint4* q = reinterpret_cast<uint4*>(p);
#pragma unroll
    for (int i = 0; i < 3; ++i) q[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}
__device__ __forceinline__ void load_raw32(uint32_t* w, const uint8_t* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
#pragma unroll
    for (int i = 0; i < 2; ++i) { uint4 v = q[i]; w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w; }
}
// y^2 == x^3 + 4  (ECP_set ecp_BLS12381.cpp:232, ECP_rhs :279)
__device__ __forceinline__ bool g1_on_curve(const fp& x, const fp& y) {
    fp x2, x3, y2, four, rhs;
    fp_sqr(x2, x); fp_mul(x3, x2, x);
    fp_set_const(four, FP_FOUR);
    fp_add(rhs, x3, four);
    fp_sqr(y2, y);
    return fp_equal(y2, rhs);
}
// parse a 96-byte affine point; all-zero = infinity
__device__ __forceinline__ void g1_parse96(fp& x, fp& y, bool& inf, bool& ok, const uint8_t* p) {
    uint32_t raw[24];
    load_raw48(raw, p); load_raw48(raw + 12, p + 48);
    inf = raw_all_zero(raw, 24);
    fp_from_raw48(x, raw); fp_from_raw48(y, raw + 12);
    ok = inf || g1_on_curve(x, y);
}

// ------------------------------------------------------------------ Fp kernels
__global__ void __launch_bounds__(BLOCK, 2) fp_op_kernel(int op, size_t n, const uint8_t* a, const uint8_t* b, uint8_t* out) {
    const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    uint32_t raw[12];
    fp x, y, r;
    load_raw48(raw, a + 48 * i); fp_from_raw48(x, raw);
    if (b) { load_raw48(raw, b + 48 * i); fp_from_raw48(y, raw); } else { fp_zero(y); }
    switch (op) {
        case 0: fp_mul(r, x, y); break;
        case 1: fp_add(r, x, y); break;
        case 2: fp_sub(r, x, y); break;
        case 3: fp_sqr(r, x); break;
        case 4: fp_neg(r, x); break;
        default: fp_inv(r, x); break;
    }
    fp_to_raw48(raw, r);
    store_raw48(out + 48 * i, raw);
}

__global__ void __launch_bounds__(BLOCK, 2) fp_mulchain_kernel(size_t n, int iters, const uint8_t* a, const uint8_t* b, uint8_t* out) {
    const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    uint32_t raw[12];
    fp x, y;
    load_raw48(raw, a + 48 * i); fp_from_raw48(x, raw);
    load_raw48(raw, b + 48 * i); fp_from_raw48(y, raw);
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        fp z;
        fp_mul(z, x, y);
        x = y; y = z;                 // x_{n+2} = x_n * x_{n+1}: both operands stay live
    }
    fp_to_raw48(raw, y);
    store_raw48(out + 48 * i, raw);
}

// ------------------------------------------------------------------ G1 kernels
// proj layout: coordinate-major, limb-major SoA: proj[(c*NL + limb) * stride + element]
// pt_stride = 96 for per-lane points, 0 to broadcast one point to every lane (fixed-base columns of BBS+)
__global__ void __launch_bounds__(BLOCK, 2) g1_mul_kernel(size_t n, const uint8_t* pts, size_t pt_stride, const uint8_t* scalars, int32_t* tab,
                                                       int32_t* proj, size_t proj_stride, size_t proj_off, int* bad_flag) {
    const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    fp px, py;
    bool inf, ok;
    g1_parse96(px, py, inf, ok, pts + pt_stride * i);
    uint32_t raw[8], k[8];
    load_raw32(raw, scalars + 32 * i);
    scalar_from_raw32(k, raw);
    g1p acc;
    g1_scalar_mul(acc, px, py, inf || !ok, k, tab + i * (size_t)G1_TAB_DWORDS);
    if (!ok) {
        *bad_flag = 1;
        // poison: Z = 0, X = 1 marks "invalid" for the finish kernel
        fp_one(acc.x); fp_zero(acc.y); fp_zero(acc.z);
    }
    g1p o;
    g1_norm1(o, acc);
    soa_store_g1(proj, proj_stride, proj_off + i, o);
}

__global__ void __launch_bounds__(BLOCK, 2) g1_add_kernel(size_t n, const uint8_t* a, const uint8_t* b, int32_t* proj, size_t proj_stride,
                                                       int* bad_flag) {
    const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    g1p p, q, inf_pt;
    bool ia, oa, ib, ob;
    g1_parse96(p.x, p.y, ia, oa, a + 96 * i); fp_one(p.z);
    g1_parse96(q.x, q.y, ib, ob, b + 96 * i); fp_one(q.z);
    g1_set_inf(inf_pt);
    fp_select(p.x, ia, inf_pt.x, p.x); fp_select(p.y, ia, inf_pt.y, p.y); fp_select(p.z, ia, inf_pt.z, p.z);
    fp_select(q.x, ib, inf_pt.x, q.x); fp_select(q.y, ib, inf_pt.y, q.y); fp_select(q.z, ib, inf_pt.z, q.z);
    g1_add(p, q);
    if (!(oa && ob)) { *bad_flag = 1; fp_one(p.x); fp_zero(p.y); fp_zero(p.z); }
    g1p o;
    g1_norm1(o, p);
    soa_store_g1(proj, proj_stride, i, o);
}

// Simultaneous inversion (Montgomery's trick) + affine + encode.  Lane t owns elements
// t, t+T, t+2T, ... so every global access is coalesced across the wavefront.
__global__ void __launch_bounds__(BLOCK, 2) g1_finish_kernel(size_t n, const int32_t* proj, size_t stride, int32_t* pref, uint8_t* out,
                                                          int fmt, size_t T) {
    const size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (t >= T || t >= n) return;
    const int32_t* zbase = proj + (size_t)2 * NL * stride;
    fp run;
    fp_one(run);
    size_t last = t;
#pragma unroll 1
    for (size_t e = t; e < n; e += T) {
        fp z, one;
        soa_load_fp(z, zbase, stride, e);
        fp_one(one);
        const bool inf = fp_is_zero(z);
        fp_select(z, inf, one, z);
        fp_mul(run, run, z);
        soa_store_fp(pref, stride, e, run);
        last = e;
    }
    fp inv;
    fp_inv(inv, run);
#pragma unroll 1
    for (size_t e = last;; e -= T) {
        g1p p;
        soa_load_g1(p, proj, stride, e);
        fp one, prev, zinv;
        fp_one(one);
        const bool inf = fp_is_zero(p.z);
        fp_select(p.z, inf, one, p.z);
        if (e >= T + t) soa_load_fp(prev, pref, stride, e - T); else prev = one;
        fp_mul(zinv, inv, prev);
        fp_mul(inv, inv, p.z);
        fp ax, ay;
        g1_to_affine(ax, ay, p, zinv);
        uint32_t rx[12], ry[12];
        fp_to_raw48(rx, ax);
        uint8_t* o = out + (size_t)fmt * e;
        // X = 1 (Montgomery), Z = 0 marks an invalid input; X = 0, Z = 0 is the point at infinity
        const bool invalid = inf && !fp_is_zero(p.x);
        if (fmt == 96) {
            fp_to_raw48(ry, ay);
            if (inf) {
#pragma unroll
                for (int j = 0; j < 12; ++j) { rx[j] = invalid ? 0xffffffffu : 0u; ry[j] = invalid ? 0xffffffffu : 0u; }
            }
            store_raw48(o, rx); store_raw48(o + 48, ry);
        } else {
            uint8_t tag = (uint8_t)(0x02 | fp_sign(ay));
            if (inf) {
                tag = invalid ? 0xff : 0x00;
#pragma unroll
                for (int j = 0; j < 12; ++j) rx[j] = invalid ? 0xffffffffu : 0u;
            }
            o[0] = tag;
#pragma unroll
            for (int j = 0; j < 12; ++j) {
                const uint32_t v = rx[j];
                o[1 + 4 * j] = (uint8_t)v; o[2 + 4 * j] = (uint8_t)(v >> 8); o[3 + 4 * j] = (uint8_t)(v >> 16); o[4 + 4 * j] = (uint8_t)(v >> 24);
            }
        }
        if (e < T + t) break;
    }
}

// One reduction level: out[j] = sum over i = j, j+m, j+2m, ... < n of in[i]   (projective, complete adds)
__global__ void __launch_bounds__(BLOCK, 2) g1_reduce_kernel(size_t n, const int32_t* in, size_t in_stride, size_t m, int32_t* outp,
                                                          size_t out_stride) {
    const size_t j = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (j >= m) return;
    g1p acc;
    g1_set_inf(acc);
#pragma unroll 1
    for (size_t i = j; i < n; i += m) {
        g1p q;
        soa_load_g1(q, in, in_stride, i);
        g1_add(acc, q);
        g1p nn;
        g1_norm1(nn, acc);
        acc = nn;
    }
    soa_store_g1(outp, out_stride, j, acc);
}

// ------------------------------------------------------------------ G2 / pairing kernels
__device__ __forceinline__ void fp2_load_raw96(fp2& r, const uint8_t* p) {       // b || a
    uint32_t raw[24];
    load_raw48(raw, p); load_raw48(raw + 12, p + 48);
    fp_from_raw48(r.b, raw); fp_from_raw48(r.a, raw + 12);
}
__device__ __forceinline__ void fp2_store_raw96(uint8_t* p, const fp2& x) {
    uint32_t raw[12];
    fp_to_raw48(raw, x.b); store_raw48(p, raw);
    fp_to_raw48(raw, x.a); store_raw48(p + 48, raw);
}
// y^2 == x^3 + 4(1+i)  (ECP2_set ecp2_BLS12381.cpp:299, ECP2_rhs :270-296)
__device__ __noinline__ bool g2_on_curve(const fp2& x, const fp2& y) {
    fp2 x2, x3, y2, b, d;
    fp2_sqr(x2, x); fp2_mul(x3, x2, x);
    fp_set_const(b.a, FP_FOUR); fp_set_const(b.b, FP_FOUR);        // 4(1+i) = 4 + 4i
    fp2_add(x3, x3, b);
    fp2_sqr(y2, y);
    fp2_sub(d, y2, x3);
    return fp2_is_zero(d);
}
__device__ __forceinline__ void g2_parse192(fp2& x, fp2& y, bool& inf, bool& ok, const uint8_t* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 12; ++i) { uint4 v = q[i]; o |= v.x | v.y | v.z | v.w; }
    inf = o == 0;
    fp2_load_raw96(x, p); fp2_load_raw96(y, p + 96);
    ok = inf || g2_on_curve(x, y);
}
// affine + canonical encoding of one projective G2 point (per-lane inversion)
__device__ __noinline__ void g2_store_affine(uint8_t* o, const g2p& acc, int fmt, bool invalid) {
    const bool inf = fp2_is_zero(acc.z);
    fp2 zn, zi, ax, ay, one;
    fp2_one(one);
    fp2_norm1(zn, acc.z);
    fp2_select(zn, inf, one, zn);
    fp2_inv(zi, zn);
    fp2_mul(ax, acc.x, zi); fp2_mul(ay, acc.y, zi);
    if (inf || invalid) {
        const uint32_t fill = invalid ? 0xffffffffu : 0u;
        if (fmt == 192) { uint4* q = reinterpret_cast<uint4*>(o); for (int i = 0; i < 12; ++i) q[i] = make_uint4(fill, fill, fill, fill); }
        else { for (int i = 0; i < 97; ++i) o[i] = (uint8_t)fill; }
        return;
    }
    if (fmt == 192) { fp2_store_raw96(o, ax); fp2_store_raw96(o + 96, ay); }
    else {
        o[0] = (uint8_t)(0x02 | fp2_sign(ay));
        uint32_t raw[24];
        fp_to_raw48(raw, ax.b); fp_to_raw48(raw + 12, ax.a);
        for (int j = 0; j < 24; ++j) { const uint32_t v = raw[j]; o[1 + 4 * j] = (uint8_t)v; o[2 + 4 * j] = (uint8_t)(v >> 8); o[3 + 4 * j] = (uint8_t)(v >> 16); o[4 + 4 * j] = (uint8_t)(v >> 24); }
    }
}

__global__ void __launch_bounds__(BLOCK, 2) g2_mul_kernel(size_t n, const uint8_t* pts, size_t pt_stride, const uint8_t* scalars, int32_t* tab,
                                                       size_t tab_stride, uint8_t* out, int fmt, int* bad_flag) {
    const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    fp2 qx, qy;
    bool inf, ok;
    g2_parse192(qx, qy, inf, ok, pts + pt_stride * i);
    uint32_t raw[8], k[8];
    load_raw32(raw, scalars + 32 * i);
    scalar_from_raw32(k, raw);
    g2p acc;
    g2_scalar_mul(acc, qx, qy, inf || !ok, k, tab + i * (size_t)G2_TAB_DWORDS);
    if (!ok) *bad_flag = 1;
    g2_store_affine(out + (size_t)fmt * i, acc, fmt, !ok);
}

__global__ void __launch_bounds__(BLOCK, 2) g2_add_kernel(size_t n, const uint8_t* a, size_t a_stride, const uint8_t* b, uint8_t* out, int fmt,
                                                       int* bad_flag) {
    const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    g2p p, q, inf_pt;
    bool ia, oa, ib, ob;
    g2_parse192(p.x, p.y, ia, oa, a + a_stride * i); fp2_one(p.z);
    g2_parse192(q.x, q.y, ib, ob, b + 192 * i); fp2_one(q.z);
    g2_set_inf(inf_pt);
    fp2_select(p.x, ia, inf_pt.x, p.x); fp2_select(p.y, ia, inf_pt.y, p.y); fp2_select(p.z, ia, inf_pt.z, p.z);
    fp2_select(q.x, ib, inf_pt.x, q.x); fp2_select(q.y, ib, inf_pt.y, q.y); fp2_select(q.z, ib, inf_pt.z, q.z);
    g2_add(p, q);
    const bool ok = oa && ob;
    if (!ok) *bad_flag = 1;
    g2_store_affine(out + (size_t)fmt * i, p, fmt, !ok);
}

__device__ __forceinline__ void gt_store576(uint8_t* o, const fp12& f, bool invalid) {
#pragma unroll 1
    for (int j = 0; j < 12; ++j) {
        uint32_t raw[12];
        fp_to_raw48(raw, fp12_coord(f, j));
        if (invalid) { for (int t = 0; t < 12; ++t) raw[t] = 0xffffffffu; }
        store_raw48(o + 48 * j, raw);
    }
}
__device__ __noinline__ void pair_inputs(fp& px, fp& py, bool& pinf, fp2& qx, fp2& qy, bool& qinf, bool& ok, const uint8_t* g1, const uint8_t* g2) {
    bool ok1, ok2;
    g1_parse96(px, py, pinf, ok1, g1);
    g2_parse192(qx, qy, qinf, ok2, g2);
    ok = ok1 && ok2;
}

__global__ void __launch_bounds__(BLOCK, 2) pair_kernel(size_t n, const uint8_t* g1, const uint8_t* g2, uint8_t* gt, int* bad_flag) {
    const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    fp px, py; fp2 qx, qy; bool pinf, qinf, ok;
    pair_inputs(px, py, pinf, qx, qy, qinf, ok, g1 + 96 * i, g2 + 192 * i);
    if (!ok) { *bad_flag = 1; pinf = true; qinf = true; }
    fp12 f;
    miller_loop(f, px, py, pinf, qx, qy, qinf);
    final_exp(f);
    gt_store576(gt + 576 * i, f, !ok);
}

__global__ void __launch_bounds__(BLOCK, 2) pair_eq_kernel(size_t n, const uint8_t* a1, const uint8_t* a2, const uint8_t* b1, const uint8_t* b2,
                                                        size_t b2_stride, uint8_t* out, int* bad_flag) {
    const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    fp px, py; fp2 qx, qy; bool pinf, qinf, ok, okb;
    fp12 f, g, t;
    pair_inputs(px, py, pinf, qx, qy, qinf, ok, a1 + 96 * i, a2 + 192 * i);
    if (!ok) { pinf = true; qinf = true; }
    miller_loop(f, px, py, pinf, qx, qy, qinf);
    pair_inputs(px, py, pinf, qx, qy, qinf, okb, b1 + 96 * i, b2 + b2_stride * i);
    if (!okb) { pinf = true; qinf = true; }
    miller_loop(g, px, py, pinf, qx, qy, qinf);
    fp12_conj(t, g);
    fp12_mul(g, f, t);
    final_exp(g);
    const bool valid = ok && okb;
    if (!valid) *bad_flag = 1;
    out[i] = valid ? (fp12_is_one(g) ? 1 : 0) : 0xff;
}

// proj[i] += P for one affine point P broadcast to every lane (BBS+: the constant g1 term)
__global__ void __launch_bounds__(BLOCK, 2) g1_add_const_kernel(size_t n, int32_t* proj, size_t stride, const uint8_t* pt96, int* bad_flag) {
    const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    g1p q, inf_pt, acc;
    bool inf, ok;
    g1_parse96(q.x, q.y, inf, ok, pt96); fp_one(q.z);
    g1_set_inf(inf_pt);
    fp_select(q.x, inf, inf_pt.x, q.x); fp_select(q.y, inf, inf_pt.y, q.y); fp_select(q.z, inf, inf_pt.z, q.z);
    if (!ok) *bad_flag = 1;
    soa_load_g1(acc, proj, stride, i);
    g1_add(acc, q);
    g1p o;
    g1_norm1(o, acc);
    soa_store_g1(proj, stride, i, o);
}

// ------------------------------------------------------------------ three-lanes-per-pairing kernels (pairing3.hpp)
constexpr int TRI_PER_WAVE = 21;
__device__ __forceinline__ void tri_setup(tri& t, size_t& idx, bool& active, size_t n) {
    const unsigned lane = threadIdx.x & 63u;
    const size_t wave = ((size_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
    const unsigned trip = lane / 3u;
    t.role = lane == 63u ? 0 : (int)(lane - 3u * trip);
    t.base = lane == 63u ? 63 : (int)(3u * trip);
    const size_t i = wave * TRI_PER_WAVE + (lane == 63u ? TRI_PER_WAVE - 1 : trip);
    active = lane < 63u && i < n;
    idx = i < n ? i : n - 1;                    // inactive lanes shadow the last element: same instruction stream
}
__device__ __forceinline__ void gt_store_coeff(uint8_t* o576, const fp4& x, int role) {
    uint8_t* o = o576 + (role == 0 ? 384 : (role == 1 ? 192 : 0));      // FP12_toOctet: c | b | a
    uint32_t raw[12];
    fp_to_raw48(raw, x.b.b); store_raw48(o, raw);
    fp_to_raw48(raw, x.b.a); store_raw48(o + 48, raw);
    fp_to_raw48(raw, x.a.b); store_raw48(o + 96, raw);
    fp_to_raw48(raw, x.a.a); store_raw48(o + 144, raw);
}
__global__ void __launch_bounds__(BLOCK, 2) pair3_kernel(size_t n, const uint8_t* g1, const uint8_t* g2, uint8_t* gt, int* bad_flag) {
    if ((((size_t)blockIdx.x * BLOCK + threadIdx.x) >> 6) * TRI_PER_WAVE >= n) return;      // whole wavefront idle
    tri t; size_t i; bool active;
    tri_setup(t, i, active, n);
    fp px, py; fp2 qx, qy; bool pinf, qinf, ok;
    pair_inputs(px, py, pinf, qx, qy, qinf, ok, g1 + 96 * i, g2 + 192 * i);
    if (!ok) { if (active) *bad_flag = 1; pinf = true; qinf = true; }
    fp4 F;
    miller3_loop(F, px, py, pinf, qx, qy, qinf, t);
    f12t_final_exp(F, t);
    if (active) {
        if (!ok) { uint4* q = reinterpret_cast<uint4*>(gt + 576 * i + (t.role == 0 ? 384 : (t.role == 1 ? 192 : 0))); for (int j = 0; j < 12; ++j) q[j] = make_uint4(~0u, ~0u, ~0u, ~0u); }
        else gt_store_coeff(gt + 576 * i, F, t.role);
    }
}
__global__ void __launch_bounds__(BLOCK, 2) pair3_eq_kernel(size_t n, const uint8_t* a1, const uint8_t* a2, const uint8_t* b1, const uint8_t* b2,
                                                         size_t b2_stride, uint8_t* out, int* bad_flag) {
    if ((((size_t)blockIdx.x * BLOCK + threadIdx.x) >> 6) * TRI_PER_WAVE >= n) return;
    tri t; size_t i; bool active;
    tri_setup(t, i, active, n);
    fp px, py; fp2 qx, qy; bool pinf, qinf, ok, okb;
    fp4 F, G, Gc;
    pair_inputs(px, py, pinf, qx, qy, qinf, ok, a1 + 96 * i, a2 + 192 * i);
    if (!ok) { pinf = true; qinf = true; }
    miller3_loop(F, px, py, pinf, qx, qy, qinf, t);
    pair_inputs(px, py, pinf, qx, qy, qinf, okb, b1 + 96 * i, b2 + b2_stride * i);
    if (!okb) { pinf = true; qinf = true; }
    miller3_loop(G, px, py, pinf, qx, qy, qinf, t);
    f12t_conj(Gc, G, t);
    f12t_mul(F, F, Gc, t);
    f12t_final_exp(F, t);
    const bool one = f12t_is_one(F, t);
    const bool valid = ok && okb;
    if (active && t.role == 0) {
        if (!valid) *bad_flag = 1;
        out[i] = valid ? (one ? 1 : 0) : 0xff;
    }
}

// ------------------------------------------------------------------ bucket-method MSM kernels (msm.hpp)
__global__ void __launch_bounds__(BLOCK, 2) msm_prep_kernel(size_t n, const uint8_t* pts, const uint8_t* scalars, int c, int W, int32_t* pts2,
                                                         uint32_t* keys, uint32_t* vals, int* bad_flag) {
    const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    uint32_t rp[24], rs[8];
    load_raw48(rp, pts + 96 * i); load_raw48(rp + 12, pts + 96 * i + 48);
    load_raw32(rs, scalars + 32 * i);
    if (!msm_prep_one(i, n, rp, rs, c, W, pts2, keys, vals)) *bad_flag = 1;
}
__global__ void __launch_bounds__(BLOCK, 2) msm_ranges_kernel(size_t E, const uint32_t* keys, uint32_t* lo, uint32_t* hi) {
    const size_t j = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (j >= E) return;
    const uint32_t k = keys[j];
    if (j == 0 || keys[j - 1] != k) lo[k] = (uint32_t)j;
    if (j + 1 == E || keys[j + 1] != k) hi[k] = (uint32_t)(j + 1);
}
__global__ void __launch_bounds__(BLOCK, 2) msm_bucket_kernel(size_t nbk, const uint32_t* lo, const uint32_t* hi, const uint32_t* vals,
                                                           const int32_t* pts2, int32_t* bk) {
    const size_t b = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (b >= nbk) return;
    g1p acc, nn;
    msm_bucket_one(acc, lo[b], hi[b], vals, pts2);
    g1_norm1(nn, acc);
    tab_store_g1(bk + b * G1_ENT_DWORDS, nn);
}
__global__ void __launch_bounds__(BLOCK, 2) msm_wreduce_kernel(int W, uint32_t nb, uint32_t chunks, const int32_t* bk, int32_t* out, size_t out_stride) {
    const size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (t >= (size_t)W * chunks) return;
    const uint32_t w = (uint32_t)(t / chunks), ch = (uint32_t)(t % chunks);
    g1p part;
    msm_wreduce_one(part, bk + (size_t)w * nb * G1_ENT_DWORDS, ch * MSM_CHUNK, nb);
    soa_store_g1(out, out_stride, (size_t)ch * W + w, part);
}
__global__ void __launch_bounds__(64, 1) msm_horner_kernel(const int32_t* rw, size_t stride, int W, int c, int32_t* out, size_t out_stride) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    g1p acc;
    msm_horner(acc, rw, stride, W, c);
    soa_store_g1(out, out_stride, 0, acc);
}

// ------------------------------------------------------------------ decode / split pairing / GT kernels
// ECP_fromOctet ecp_BLS12381.cpp:495-545 for 49-byte input (tags 02/03; a leading 00 is infinity as in
// g1_point.hpp:89-93); status 1 ok / 0 reject; rejected and infinity lanes give 96 zero bytes.
__global__ void __launch_bounds__(BLOCK, 2) g1_decompress_kernel(size_t n, const uint8_t* in, uint8_t* out, uint8_t* status) {
    const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const uint8_t* sp = in + 49 * i;
    const uint8_t tag = sp[0];
    uint32_t raw[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) raw[j] = (uint32_t)sp[1 + 4 * j] | ((uint32_t)sp[2 + 4 * j] << 8) | ((uint32_t)sp[3 + 4 * j] << 16) | ((uint32_t)sp[4 + 4 * j] << 24);
    fp x, y;
    fp_from_raw48(x, raw);
    const bool ok_tag = tag == 2 || tag == 3;
    const bool ok = g1_set_x(y, x, tag & 1) && ok_tag;
    uint32_t rx[12], ry[12];
    fp_to_raw48(rx, x); fp_to_raw48(ry, y);
    if (!ok) {
#pragma unroll
        for (int j = 0; j < 12; ++j) { rx[j] = 0; ry[j] = 0; }
    }
    store_raw48(out + 96 * i, rx); store_raw48(out + 96 * i + 48, ry);
    status[i] = tag == 0 ? 1 : (ok ? 1 : 0);
}
// ECP2_fromOctet ecp2_BLS12381.cpp:225-266 for 97-byte input: any tag other than 04 is "compressed, sign = tag & 1"
__global__ void __launch_bounds__(BLOCK, 2) g2_decompress_kernel(size_t n, const uint8_t* in, uint8_t* out, uint8_t* status) {
    const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const uint8_t* sp = in + 97 * i;
    const uint8_t tag = sp[0];
    uint32_t raw[24];
#pragma unroll
    for (int j = 0; j < 24; ++j) raw[j] = (uint32_t)sp[1 + 4 * j] | ((uint32_t)sp[2 + 4 * j] << 8) | ((uint32_t)sp[3 + 4 * j] << 16) | ((uint32_t)sp[4 + 4 * j] << 24);
    fp2 x, y;
    fp_from_raw48(x.b, raw); fp_from_raw48(x.a, raw + 12);
    const bool ok = g2_set_x(y, x, tag & 1) && tag != 0 && tag != 4;
    uint8_t* o = out + 192 * i;
    if (ok) { fp2_store_raw96(o, x); fp2_store_raw96(o + 96, y); }
    else { uint4* q = reinterpret_cast<uint4*>(o); for (int j = 0; j < 12; ++j) q[j] = make_uint4(0, 0, 0, 0); }
    status[i] = tag == 0 ? 1 : (ok ? 1 : 0);
}

__device__ __noinline__ void gt_load576(fp12& f, const uint8_t* p) {
#pragma unroll 1
    for (int j = 0; j < 12; ++j) {
        uint32_t raw[12];
        load_raw48(raw, p + 48 * j);
        fp_from_raw48(fp12_coord_mut(f, j), raw);
    }
}
// pair_ate alone: the Miller value as FP12_toOctet bytes (the same field element as the reference's)
__global__ void __launch_bounds__(BLOCK, 2) miller_kernel(size_t n, const uint8_t* g1, const uint8_t* g2, uint8_t* out, int* bad_flag) {
    const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    fp px, py; fp2 qx, qy; bool pinf, qinf, ok;
    pair_inputs(px, py, pinf, qx, qy, qinf, ok, g1 + 96 * i, g2 + 192 * i);
    if (!ok) { *bad_flag = 1; pinf = true; qinf = true; }
    fp12 f;
    miller_loop(f, px, py, pinf, qx, qy, qinf);
    gt_store576(out + 576 * i, f, !ok);
}
// op 0: a*b (FP12_mul), 1: conj(a), 2: a^e (FP12_pow, e = 32-byte exponent used as given), 3: final exponentiation
__global__ void __launch_bounds__(BLOCK, 2) gt_op_kernel(int op, size_t n, const uint8_t* a, const uint8_t* b, uint8_t* out) {
    const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    fp12 x, r;
    gt_load576(x, a + 576 * i);
    if (op == 0) { fp12 y; gt_load576(y, b + 576 * i); fp12_mul(r, x, y); }
    else if (op == 1) { fp12_conj(r, x); }
    else if (op == 2) { uint32_t raw[8], e[8]; load_raw32(raw, b + 32 * i); scalar_from_raw32(e, raw); fp12_pow_generic(r, x, e); }
    else { r = x; final_exp(r); }
    gt_store576(out + 576 * i, r, false);
}
// FP12_isunity per element
__global__ void __launch_bounds__(BLOCK, 2) gt_is_unity_kernel(size_t n, const uint8_t* a, uint8_t* out) {
    const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    fp12 x;
    gt_load576(x, a + 576 * i);
    out[i] = fp12_is_one(x) ? 1 : 0;
}

}  // namespace

// ====================================================================== host side
struct c12381_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    char err[256] = {0};
    enum { WS_TAB, WS_PROJ, WS_PREF, WS_IN0, WS_IN1, WS_OUT, WS_RED0, WS_RED1, WS_BBS_Q, WS_BBS_B, WS_BBS_IN,
           WS_MSM_PTS, WS_MSM_K0, WS_MSM_K1, WS_MSM_V0, WS_MSM_V1, WS_MSM_TMP, WS_MSM_RNG, WS_MSM_BK, WS_COUNT };
    void* ws[WS_COUNT] = {nullptr};
    size_t ws_bytes[WS_COUNT] = {0};
    int* d_flag = nullptr;
    int* h_flag = nullptr;          // pinned
    // optional per-kernel timing (HIP events on the context's stream), see c12381_profile()
    bool profiling = false;
    struct ev_pair { hipEvent_t a, b; int kind; };
    std::vector<ev_pair> events;
};

namespace {

int fail(c12381_ctx* c, hipError_t e, const char* what) {
    std::snprintf(c->err, sizeof c->err, "%s: %s", what, hipGetErrorString(e));
    return C12381_E_HIP;
}
#define HIPCK(c, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail((c), e_, #call); } while (0)

int ensure(c12381_ctx* c, int slot, size_t bytes) {
    if (c->ws_bytes[slot] >= bytes) return 0;
    if (c->ws[slot]) { HIPCK(c, hipFree(c->ws[slot])); c->ws[slot] = nullptr; c->ws_bytes[slot] = 0; }
    hipError_t e = hipMalloc(&c->ws[slot], bytes);
    if (e != hipSuccess) { std::snprintf(c->err, sizeof c->err, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); return C12381_E_NOMEM; }
    c->ws_bytes[slot] = bytes;
    return 0;
}
inline unsigned grid_for(size_t n) { return (unsigned)((n + BLOCK - 1) / BLOCK); }
inline size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

int bind(c12381_ctx* c) {
    if (!c) return C12381_E_ARG;
    HIPCK(c, hipSetDevice(c->device));
    return 0;
}

// HIP-event bracket around a dominant-kernel launch (kind: 0 = g1_mul_kernel, 1 = g1_finish_kernel, ...)
struct timed {
    c12381_ctx* c; int idx = -1;
    timed(c12381_ctx* c_, int kind) : c(c_) {
        if (!c->profiling) return;
        c12381_ctx::ev_pair p; p.kind = kind;
        if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) return;
        (void)hipEventRecord(p.a, c->stream);
        c->events.push_back(p); idx = (int)c->events.size() - 1;
    }
    ~timed() { if (idx >= 0) (void)hipEventRecord(c->events[idx].b, c->stream); }
};

// scalar multiplication of n elements into the projective SoA workspace (stride = padded n)
// (results land at proj[proj_off + i]; pt_stride 96 = per-lane points, 0 = one broadcast point)
int g1_mul_to_proj(c12381_ctx* c, size_t n, const uint8_t* d_pts, const uint8_t* d_sc, size_t stride, size_t pt_stride = 96,
                   size_t proj_off = 0) {
    const size_t chunk = n < G1_CHUNK ? round_up(n, 64) : G1_CHUNK;
    int rc;
    if ((rc = ensure(c, c12381_ctx::WS_TAB, (size_t)G1_TAB_DWORDS * chunk * 4))) return rc;
    if ((rc = ensure(c, c12381_ctx::WS_PROJ, (size_t)3 * NL * stride * 4))) return rc;
    for (size_t off = 0; off < n; off += chunk) {
        const size_t m = n - off < chunk ? n - off : chunk;
        timed tm(c, 0);
        hipLaunchKernelGGL(g1_mul_kernel, dim3(grid_for(m)), dim3(BLOCK), 0, c->stream, m, d_pts + pt_stride * off, pt_stride, d_sc + 32 * off,
                           (int32_t*)c->ws[c12381_ctx::WS_TAB], (int32_t*)c->ws[c12381_ctx::WS_PROJ], stride, proj_off + off, c->d_flag);
        HIPCK(c, hipGetLastError());
    }
    return 0;
}
int g1_finish(c12381_ctx* c, size_t n, const int32_t* proj, size_t stride, uint8_t* d_out, int fmt) {
    int rc;
    if ((rc = ensure(c, c12381_ctx::WS_PREF, (size_t)NL * stride * 4))) return rc;
    size_t T = round_up((n + FINISH_M - 1) / FINISH_M, 64);
    if (T > n) T = n;
    timed tm(c, 1);
    hipLaunchKernelGGL(g1_finish_kernel, dim3(grid_for(T)), dim3(BLOCK), 0, c->stream, n, proj, stride,
                       (int32_t*)c->ws[c12381_ctx::WS_PREF], d_out, fmt, T);
    HIPCK(c, hipGetLastError());
    return 0;
}
int read_flag(c12381_ctx* c) {
    HIPCK(c, hipMemcpyAsync(c->h_flag, c->d_flag, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipMemsetAsync(c->d_flag, 0, sizeof(int), c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    return *c->h_flag ? C12381_E_POINT : 0;
}
// stage host buffers: copies up to three inputs in, runs body, copies output back
struct staged {
    uint8_t *in0 = nullptr, *in1 = nullptr, *out = nullptr;
};
int stage_in(c12381_ctx* c, staged& s, const void* h0, size_t b0, const void* h1, size_t b1, size_t bout) {
    int rc;
    if ((rc = ensure(c, c12381_ctx::WS_IN0, round_up(b0 ? b0 : 16, 256)))) return rc;
    if ((rc = ensure(c, c12381_ctx::WS_IN1, round_up(b1 ? b1 : 16, 256)))) return rc;
    if ((rc = ensure(c, c12381_ctx::WS_OUT, round_up(bout ? bout : 16, 256)))) return rc;
    s.in0 = (uint8_t*)c->ws[c12381_ctx::WS_IN0]; s.in1 = (uint8_t*)c->ws[c12381_ctx::WS_IN1]; s.out = (uint8_t*)c->ws[c12381_ctx::WS_OUT];
    if (h0 && b0) HIPCK(c, hipMemcpyAsync(s.in0, h0, b0, hipMemcpyHostToDevice, c->stream));
    if (h1 && b1) HIPCK(c, hipMemcpyAsync(s.in1, h1, b1, hipMemcpyHostToDevice, c->stream));
    return 0;
}
int stage_out(c12381_ctx* c, const staged& s, void* hout, size_t bout) {
    HIPCK(c, hipMemcpyAsync(hout, s.out, bout, hipMemcpyDeviceToHost, c->stream));
    return 0;
}

// Bucket-method MSM (msm.hpp): prep -> radix sort -> bucket sums -> window reduction -> Horner -> affine.
int g1_msm_pippenger(c12381_ctx* c, size_t n, const uint8_t* pts, const uint8_t* sc, uint8_t* out, int fmt) {
    const int cb = msm_window_bits(n), W = msm_windows(cb);
    const size_t E = (size_t)2 * n * W, nb = (size_t)1 << cb, nbk = nb * W;
    int rc;
    if ((rc = ensure(c, c12381_ctx::WS_MSM_PTS, (size_t)2 * n * MSM_PT_DWORDS * 4))) return rc;
    if ((rc = ensure(c, c12381_ctx::WS_MSM_K0, E * 4))) return rc;
    if ((rc = ensure(c, c12381_ctx::WS_MSM_K1, E * 4))) return rc;
    if ((rc = ensure(c, c12381_ctx::WS_MSM_V0, E * 4))) return rc;
    if ((rc = ensure(c, c12381_ctx::WS_MSM_V1, E * 4))) return rc;
    if ((rc = ensure(c, c12381_ctx::WS_MSM_RNG, (nbk + 1) * 8))) return rc;
    if ((rc = ensure(c, c12381_ctx::WS_MSM_BK, nbk * G1_ENT_DWORDS * 4))) return rc;
    int32_t* pts2 = (int32_t*)c->ws[c12381_ctx::WS_MSM_PTS];
    uint32_t *k0 = (uint32_t*)c->ws[c12381_ctx::WS_MSM_K0], *k1 = (uint32_t*)c->ws[c12381_ctx::WS_MSM_K1];
    uint32_t *v0 = (uint32_t*)c->ws[c12381_ctx::WS_MSM_V0], *v1 = (uint32_t*)c->ws[c12381_ctx::WS_MSM_V1];
    uint32_t* lo = (uint32_t*)c->ws[c12381_ctx::WS_MSM_RNG];
    uint32_t* hi = lo + nbk + 1;
    int32_t* bk = (int32_t*)c->ws[c12381_ctx::WS_MSM_BK];
    hipLaunchKernelGGL(msm_prep_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, c->stream, n, pts, sc, cb, W, pts2, k0, v0, c->d_flag);
    HIPCK(c, hipGetLastError());
    int end_bit = cb;
    while ((1 << (end_bit - cb)) <= W) ++end_bit;              // keys < (W + 1) << cb
    size_t tmp_bytes = 0;
    HIPCK(c, hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, k0, k1, v0, v1, (int)E, 0, end_bit, c->stream));
    if ((rc = ensure(c, c12381_ctx::WS_MSM_TMP, tmp_bytes + 256))) return rc;
    HIPCK(c, hipcub::DeviceRadixSort::SortPairs(c->ws[c12381_ctx::WS_MSM_TMP], tmp_bytes, k0, k1, v0, v1, (int)E, 0, end_bit, c->stream));
    HIPCK(c, hipMemsetAsync(lo, 0, (nbk + 1) * 8, c->stream));
    hipLaunchKernelGGL(msm_ranges_kernel, dim3(grid_for(E)), dim3(BLOCK), 0, c->stream, E, k1, lo, hi);
    HIPCK(c, hipGetLastError());
    {
        timed tm(c, 5);
        hipLaunchKernelGGL(msm_bucket_kernel, dim3(grid_for(nbk)), dim3(BLOCK), 0, c->stream, nbk, lo, hi, v1, pts2, bk);
        HIPCK(c, hipGetLastError());
    }
    const uint32_t chunks = (uint32_t)((nb + MSM_CHUNK - 1) / MSM_CHUNK);
    size_t cur_n = (size_t)W * chunks, cur_stride = round_up(cur_n, 64);
    if ((rc = ensure(c, c12381_ctx::WS_RED0, (size_t)3 * NL * cur_stride * 4))) return rc;
    hipLaunchKernelGGL(msm_wreduce_kernel, dim3(grid_for(cur_n)), dim3(BLOCK), 0, c->stream, W, (uint32_t)nb, chunks, bk,
                       (int32_t*)c->ws[c12381_ctx::WS_RED0], cur_stride);
    HIPCK(c, hipGetLastError());
    // per-window sums: element index = chunk * W + w, so reducing modulo (W * r) keeps windows apart
    const int32_t* cur = (const int32_t*)c->ws[c12381_ctx::WS_RED0];
    int slot = c12381_ctx::WS_RED1;
    while (cur_n > (size_t)W) {
        size_t groups = cur_n / W;                           // points per window still to be summed
        size_t r = groups > 32 ? (groups + 31) / 32 : 1;     // keep r partial sums per window
        const size_t m = (size_t)W * r, m_stride = round_up(m, 64);
        if ((rc = ensure(c, slot, (size_t)3 * NL * m_stride * 4))) return rc;
        hipLaunchKernelGGL(g1_reduce_kernel, dim3(grid_for(m)), dim3(BLOCK), 0, c->stream, cur_n, cur, cur_stride, m, (int32_t*)c->ws[slot], m_stride);
        HIPCK(c, hipGetLastError());
        cur = (const int32_t*)c->ws[slot]; cur_n = m; cur_stride = m_stride;
        slot = slot == c12381_ctx::WS_RED0 ? c12381_ctx::WS_RED1 : c12381_ctx::WS_RED0;
    }
    if ((rc = ensure(c, c12381_ctx::WS_PROJ, (size_t)3 * NL * 64 * 4))) return rc;
    hipLaunchKernelGGL(msm_horner_kernel, dim3(1), dim3(64), 0, c->stream, cur, cur_stride, W, cb, (int32_t*)c->ws[c12381_ctx::WS_PROJ], (size_t)64);
    HIPCK(c, hipGetLastError());
    return g1_finish(c, 1, (const int32_t*)c->ws[c12381_ctx::WS_PROJ], 64, out, fmt);
}
// C12381_MSM=naive forces the n-scalar-muls + tree-sum path (A/B measurements); default: buckets from 2^12 terms
static bool msm_use_buckets(size_t n) {
    static const int mode = [] { const char* e = std::getenv("C12381_MSM"); return e ? (e[0] == 'n' ? 1 : (e[0] == 'b' ? 2 : 0)) : 0; }();
    if (mode == 1) return false;
    if (mode == 2) return n >= 2;
    return n >= 4096;
}
}  // namespace

extern "C" {

int c12381_version(void) { return (0 << 16) | 1; }

int c12381_create(int device, c12381_ctx** out) {
    if (!out) return C12381_E_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) return C12381_E_HIP;
    c12381_ctx* c = new (std::nothrow) c12381_ctx;
    if (!c) return C12381_E_NOMEM;
    c->device = device;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return C12381_E_HIP; }
    c->own_stream = true;
    if (hipMalloc((void**)&c->d_flag, sizeof(int)) != hipSuccess || hipHostMalloc((void**)&c->h_flag, sizeof(int)) != hipSuccess ||
        hipMemset(c->d_flag, 0, sizeof(int)) != hipSuccess) { c12381_destroy(c); return C12381_E_HIP; }
    *out = c;
    return 0;
}

void c12381_destroy(c12381_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (auto& p : c->events) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    for (int i = 0; i < c12381_ctx::WS_COUNT; ++i) if (c->ws[i]) (void)hipFree(c->ws[i]);
    if (c->d_flag) (void)hipFree(c->d_flag);
    if (c->h_flag) (void)hipHostFree(c->h_flag);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char* c12381_last_error(const c12381_ctx* c) { return c ? c->err : "null context"; }

int c12381_set_stream(c12381_ctx* c, void* hip_stream) {
    int rc = bind(c); if (rc) return rc;
    HIPCK(c, hipStreamSynchronize(c->stream));
    if (c->own_stream) { HIPCK(c, hipStreamDestroy(c->stream)); c->own_stream = false; }
    if (hip_stream) { c->stream = (hipStream_t)hip_stream; }
    else { HIPCK(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)); c->own_stream = true; }
    return 0;
}

int c12381_sync(c12381_ctx* c) {
    int rc = bind(c); if (rc) return rc;
    return read_flag(c);
}

int c12381_profile(c12381_ctx* c, int enable) {
    int rc = bind(c); if (rc) return rc;
    HIPCK(c, hipStreamSynchronize(c->stream));
    for (auto& p : c->events) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    c->events.clear();
    c->profiling = enable != 0;
    return 0;
}
int c12381_profile_read(c12381_ctx* c, int kind, double* total_ms, uint64_t* launches) {
    int rc = bind(c); if (rc) return rc;
    if (!total_ms || !launches) return C12381_E_ARG;
    HIPCK(c, hipStreamSynchronize(c->stream));
    double ms = 0; uint64_t cnt = 0;
    for (auto& p : c->events) {
        if (p.kind != kind) continue;
        float t = 0;
        HIPCK(c, hipEventElapsedTime(&t, p.a, p.b));
        ms += t; ++cnt;
    }
    *total_ms = ms; *launches = cnt;
    return 0;
}

// ---------------------------------------------------------------- Fp
int c12381_fp_op_batch_dev(c12381_ctx* c, int op, size_t n, const uint8_t* a, const uint8_t* b, uint8_t* out) {
    int rc = bind(c); if (rc) return rc;
    if (op < 0 || op > 5 || !a || !out || (op <= 2 && !b)) return C12381_E_ARG;
    if (n == 0) return 0;
    hipLaunchKernelGGL(fp_op_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, c->stream, op, n, a, op <= 2 ? b : nullptr, out);
    HIPCK(c, hipGetLastError());
    return 0;
}
int c12381_fp_op_batch(c12381_ctx* c, int op, size_t n, const uint8_t* a, const uint8_t* b, uint8_t* out) {
    int rc = bind(c); if (rc) return rc;
    if (op < 0 || op > 5 || !a || !out || (op <= 2 && !b)) return C12381_E_ARG;
    if (n == 0) return 0;
    staged s;
    if ((rc = stage_in(c, s, a, 48 * n, op <= 2 ? b : nullptr, op <= 2 ? 48 * n : 0, 48 * n))) return rc;
    if ((rc = c12381_fp_op_batch_dev(c, op, n, s.in0, s.in1, s.out))) return rc;
    if ((rc = stage_out(c, s, out, 48 * n))) return rc;
    HIPCK(c, hipStreamSynchronize(c->stream));
    return 0;
}
int c12381_fp_mulchain_dev(c12381_ctx* c, size_t n, int iters, const uint8_t* a, const uint8_t* b, uint8_t* out) {
    int rc = bind(c); if (rc) return rc;
    if (!a || !b || !out || iters < 0) return C12381_E_ARG;
    if (n == 0) return 0;
    hipLaunchKernelGGL(fp_mulchain_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, c->stream, n, iters, a, b, out);
    HIPCK(c, hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------- G1
int c12381_g1_mul_batch_dev(c12381_ctx* c, size_t n, const uint8_t* pts, const uint8_t* sc, uint8_t* out, int fmt) {
    int rc = bind(c); if (rc) return rc;
    if (!pts || !sc || !out || (fmt != 49 && fmt != 96)) return C12381_E_ARG;
    if (n == 0) return 0;
    const size_t stride = round_up(n, 64);
    if ((rc = g1_mul_to_proj(c, n, pts, sc, stride))) return rc;
    return g1_finish(c, n, (const int32_t*)c->ws[c12381_ctx::WS_PROJ], stride, out, fmt);
}
int c12381_g1_mul_batch(c12381_ctx* c, size_t n, const uint8_t* pts, const uint8_t* sc, uint8_t* out, int fmt) {
    int rc = bind(c); if (rc) return rc;
    if (!pts || !sc || !out || (fmt != 49 && fmt != 96)) return C12381_E_ARG;
    if (n == 0) return 0;
    staged s;
    if ((rc = stage_in(c, s, pts, 96 * n, sc, 32 * n, (size_t)fmt * n))) return rc;
    if ((rc = c12381_g1_mul_batch_dev(c, n, s.in0, s.in1, s.out, fmt))) return rc;
    if ((rc = stage_out(c, s, out, (size_t)fmt * n))) return rc;
    return read_flag(c);
}
int c12381_g1_add_batch(c12381_ctx* c, size_t n, const uint8_t* a, const uint8_t* b, uint8_t* out, int fmt) {
    int rc = bind(c); if (rc) return rc;
    if (!a || !b || !out || (fmt != 49 && fmt != 96)) return C12381_E_ARG;
    if (n == 0) return 0;
    staged s;
    if ((rc = stage_in(c, s, a, 96 * n, b, 96 * n, (size_t)fmt * n))) return rc;
    const size_t stride = round_up(n, 64);
    if ((rc = ensure(c, c12381_ctx::WS_PROJ, (size_t)3 * NL * stride * 4))) return rc;
    hipLaunchKernelGGL(g1_add_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, c->stream, n, s.in0, s.in1, (int32_t*)c->ws[c12381_ctx::WS_PROJ], stride,
                       c->d_flag);
    HIPCK(c, hipGetLastError());
    if ((rc = g1_finish(c, n, (const int32_t*)c->ws[c12381_ctx::WS_PROJ], stride, s.out, fmt))) return rc;
    if ((rc = stage_out(c, s, out, (size_t)fmt * n))) return rc;
    return read_flag(c);
}

// MSM, round-1 algorithm: n independent GLV scalar multiplications followed by a tree sum of the
// projective results (the reference's Π is also O(n) full scalar-muls, g1_point.hpp:389-401); only
// the final point is canonical.  A bucket method is a later optimisation behind the same entry.
int c12381_g1_msm_dev(c12381_ctx* c, size_t n, const uint8_t* pts, const uint8_t* sc, uint8_t* out, int fmt) {
    int rc = bind(c); if (rc) return rc;
    if (!out || (n && (!pts || !sc)) || (fmt != 49 && fmt != 96)) return C12381_E_ARG;
    if (n == 0) { HIPCK(c, hipMemsetAsync(out, 0, fmt, c->stream)); return 0; }
    if (msm_use_buckets(n)) return g1_msm_pippenger(c, n, pts, sc, out, fmt);
    const size_t stride = round_up(n, 64);
    if ((rc = g1_mul_to_proj(c, n, pts, sc, stride))) return rc;
    const int32_t* cur = (const int32_t*)c->ws[c12381_ctx::WS_PROJ];
    size_t cur_n = n, cur_stride = stride;
    int slot = c12381_ctx::WS_RED0;
    while (cur_n > 1) {
        size_t m = cur_n > 4096 ? round_up(cur_n / 32, 64) : (cur_n > 64 ? 64 : 1);
        const size_t m_stride = round_up(m, 64);
        if ((rc = ensure(c, slot, (size_t)3 * NL * m_stride * 4))) return rc;
        hipLaunchKernelGGL(g1_reduce_kernel, dim3(grid_for(m)), dim3(BLOCK), 0, c->stream, cur_n, cur, cur_stride, m, (int32_t*)c->ws[slot], m_stride);
        HIPCK(c, hipGetLastError());
        cur = (const int32_t*)c->ws[slot]; cur_n = m; cur_stride = m_stride;
        slot = slot == c12381_ctx::WS_RED0 ? c12381_ctx::WS_RED1 : c12381_ctx::WS_RED0;
    }
    return g1_finish(c, 1, cur, cur_stride, out, fmt);
}
int c12381_g1_msm(c12381_ctx* c, size_t n, const uint8_t* pts, const uint8_t* sc, uint8_t* out, int fmt) {
    int rc = bind(c); if (rc) return rc;
    if (!out || (n && (!pts || !sc)) || (fmt != 49 && fmt != 96)) return C12381_E_ARG;
    staged s;
    if ((rc = stage_in(c, s, pts, 96 * n, sc, 32 * n, (size_t)fmt))) return rc;
    if ((rc = c12381_g1_msm_dev(c, n, s.in0, s.in1, s.out, fmt))) return rc;
    if ((rc = stage_out(c, s, out, (size_t)fmt))) return rc;
    return read_flag(c);
}

// ---------------------------------------------------------------- G2
static int g2_mul_dev_strided(c12381_ctx* c, size_t n, const uint8_t* pts, size_t pt_stride, const uint8_t* sc, uint8_t* out, int fmt);
int c12381_g2_mul_batch_dev(c12381_ctx* c, size_t n, const uint8_t* pts, const uint8_t* sc, uint8_t* out, int fmt) {
    return g2_mul_dev_strided(c, n, pts, 192, sc, out, fmt);
}
static int g2_mul_dev_strided(c12381_ctx* c, size_t n, const uint8_t* pts, size_t pt_stride, const uint8_t* sc, uint8_t* out, int fmt) {
    int rc = bind(c); if (rc) return rc;
    if (!pts || !sc || !out || (fmt != 97 && fmt != 192)) return C12381_E_ARG;
    if (n == 0) return 0;
    const size_t chunk = n < G2_CHUNK ? round_up(n, 64) : G2_CHUNK;
    if ((rc = ensure(c, c12381_ctx::WS_TAB, (size_t)G2_TAB_DWORDS * chunk * 4))) return rc;
    for (size_t off = 0; off < n; off += chunk) {
        const size_t m = n - off < chunk ? n - off : chunk;
        timed tm(c, 2);
        hipLaunchKernelGGL(g2_mul_kernel, dim3(grid_for(m)), dim3(BLOCK), 0, c->stream, m, pts + pt_stride * off, pt_stride, sc + 32 * off,
                           (int32_t*)c->ws[c12381_ctx::WS_TAB], chunk, out + (size_t)fmt * off, fmt, c->d_flag);
        HIPCK(c, hipGetLastError());
    }
    return 0;
}
int c12381_g2_mul_batch(c12381_ctx* c, size_t n, const uint8_t* pts, const uint8_t* sc, uint8_t* out, int fmt) {
    int rc = bind(c); if (rc) return rc;
    if (!pts || !sc || !out || (fmt != 97 && fmt != 192)) return C12381_E_ARG;
    if (n == 0) return 0;
    staged s;
    if ((rc = stage_in(c, s, pts, 192 * n, sc, 32 * n, (size_t)fmt * n))) return rc;
    if ((rc = c12381_g2_mul_batch_dev(c, n, s.in0, s.in1, s.out, fmt))) return rc;
    if ((rc = stage_out(c, s, out, (size_t)fmt * n))) return rc;
    return read_flag(c);
}
int c12381_g2_add_batch(c12381_ctx* c, size_t n, const uint8_t* a, const uint8_t* b, uint8_t* out, int fmt) {
    int rc = bind(c); if (rc) return rc;
    if (!a || !b || !out || (fmt != 97 && fmt != 192)) return C12381_E_ARG;
    if (n == 0) return 0;
    staged s;
    if ((rc = stage_in(c, s, a, 192 * n, b, 192 * n, (size_t)fmt * n))) return rc;
    hipLaunchKernelGGL(g2_add_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, c->stream, n, s.in0, (size_t)192, s.in1, s.out, fmt, c->d_flag);
    HIPCK(c, hipGetLastError());
    if ((rc = stage_out(c, s, out, (size_t)fmt * n))) return rc;
    return read_flag(c);
}

// ---------------------------------------------------------------- pairing
// C12381_PAIR_LANES=1 selects the one-lane-per-pairing kernels (kept for A/B measurements); default is 3.
static int pair_lanes() {
    static const int v = [] { const char* e = std::getenv("C12381_PAIR_LANES"); return (e && e[0] == '1') ? 1 : 3; }();
    return v;
}
static unsigned grid_tri(size_t n) {
    const size_t waves = (n + TRI_PER_WAVE - 1) / TRI_PER_WAVE;
    return (unsigned)((waves * 64 + BLOCK - 1) / BLOCK);
}
int c12381_pair_batch_dev(c12381_ctx* c, size_t n, const uint8_t* g1, const uint8_t* g2, uint8_t* gt) {
    int rc = bind(c); if (rc) return rc;
    if (!g1 || !g2 || !gt) return C12381_E_ARG;
    if (n == 0) return 0;
    timed tm(c, 3);
    if (pair_lanes() == 1) hipLaunchKernelGGL(pair_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, c->stream, n, g1, g2, gt, c->d_flag);
    else hipLaunchKernelGGL(pair3_kernel, dim3(grid_tri(n)), dim3(BLOCK), 0, c->stream, n, g1, g2, gt, c->d_flag);
    HIPCK(c, hipGetLastError());
    return 0;
}
int c12381_pair_batch(c12381_ctx* c, size_t n, const uint8_t* g1, const uint8_t* g2, uint8_t* gt) {
    int rc = bind(c); if (rc) return rc;
    if (!g1 || !g2 || !gt) return C12381_E_ARG;
    if (n == 0) return 0;
    staged s;
    if ((rc = stage_in(c, s, g1, 96 * n, g2, 192 * n, 576 * n))) return rc;
    if ((rc = c12381_pair_batch_dev(c, n, s.in0, s.in1, s.out))) return rc;
    if ((rc = stage_out(c, s, gt, 576 * n))) return rc;
    return read_flag(c);
}
int c12381_pair_eq_batch_dev(c12381_ctx* c, size_t n, const uint8_t* a1, const uint8_t* a2, const uint8_t* b1, const uint8_t* b2, uint8_t* ok) {
    int rc = bind(c); if (rc) return rc;
    if (!a1 || !a2 || !b1 || !b2 || !ok) return C12381_E_ARG;
    if (n == 0) return 0;
    timed tm(c, 4);
    if (pair_lanes() == 1) hipLaunchKernelGGL(pair_eq_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, c->stream, n, a1, a2, b1, b2, (size_t)192, ok, c->d_flag);
    else hipLaunchKernelGGL(pair3_eq_kernel, dim3(grid_tri(n)), dim3(BLOCK), 0, c->stream, n, a1, a2, b1, b2, (size_t)192, ok, c->d_flag);
    HIPCK(c, hipGetLastError());
    return 0;
}
int c12381_pair_eq_batch(c12381_ctx* c, size_t n, const uint8_t* a1, const uint8_t* a2, const uint8_t* b1, const uint8_t* b2, uint8_t* ok) {
    int rc = bind(c); if (rc) return rc;
    if (!a1 || !a2 || !b1 || !b2 || !ok) return C12381_E_ARG;
    if (n == 0) return 0;
    int r2;
    if ((r2 = ensure(c, c12381_ctx::WS_IN0, round_up(96 * n, 256)))) return r2;
    if ((r2 = ensure(c, c12381_ctx::WS_IN1, round_up(192 * n, 256)))) return r2;
    if ((r2 = ensure(c, c12381_ctx::WS_RED0, round_up(96 * n, 256)))) return r2;
    if ((r2 = ensure(c, c12381_ctx::WS_RED1, round_up(192 * n, 256)))) return r2;
    if ((r2 = ensure(c, c12381_ctx::WS_OUT, round_up(n, 256)))) return r2;
    uint8_t* d_a1 = (uint8_t*)c->ws[c12381_ctx::WS_IN0]; uint8_t* d_a2 = (uint8_t*)c->ws[c12381_ctx::WS_IN1];
    uint8_t* d_b1 = (uint8_t*)c->ws[c12381_ctx::WS_RED0]; uint8_t* d_b2 = (uint8_t*)c->ws[c12381_ctx::WS_RED1];
    uint8_t* d_ok = (uint8_t*)c->ws[c12381_ctx::WS_OUT];
    HIPCK(c, hipMemcpyAsync(d_a1, a1, 96 * n, hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipMemcpyAsync(d_a2, a2, 192 * n, hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipMemcpyAsync(d_b1, b1, 96 * n, hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipMemcpyAsync(d_b2, b2, 192 * n, hipMemcpyHostToDevice, c->stream));
    if ((rc = c12381_pair_eq_batch_dev(c, n, d_a1, d_a2, d_b1, d_b2, d_ok))) return rc;
    HIPCK(c, hipMemcpyAsync(ok, d_ok, n, hipMemcpyDeviceToHost, c->stream));
    return read_flag(c);
}

// ---------------------------------------------------------------- decode / split pairing / GT
int c12381_g1_decompress_batch(c12381_ctx* c, size_t n, const uint8_t* in49, uint8_t* out96, uint8_t* status) {
    int rc = bind(c); if (rc) return rc;
    if (!in49 || !out96 || !status) return C12381_E_ARG;
    if (n == 0) return 0;
    staged s;
    if ((rc = stage_in(c, s, in49, 49 * n, nullptr, n, 96 * n))) return rc;
    hipLaunchKernelGGL(g1_decompress_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, c->stream, n, s.in0, s.out, s.in1);
    HIPCK(c, hipGetLastError());
    if ((rc = stage_out(c, s, out96, 96 * n))) return rc;
    HIPCK(c, hipMemcpyAsync(status, s.in1, n, hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    return 0;
}
int c12381_g2_decompress_batch(c12381_ctx* c, size_t n, const uint8_t* in97, uint8_t* out192, uint8_t* status) {
    int rc = bind(c); if (rc) return rc;
    if (!in97 || !out192 || !status) return C12381_E_ARG;
    if (n == 0) return 0;
    staged s;
    if ((rc = stage_in(c, s, in97, 97 * n, nullptr, n, 192 * n))) return rc;
    hipLaunchKernelGGL(g2_decompress_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, c->stream, n, s.in0, s.out, s.in1);
    HIPCK(c, hipGetLastError());
    if ((rc = stage_out(c, s, out192, 192 * n))) return rc;
    HIPCK(c, hipMemcpyAsync(status, s.in1, n, hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    return 0;
}
int c12381_miller_batch(c12381_ctx* c, size_t n, const uint8_t* g1, const uint8_t* g2, uint8_t* out576) {
    int rc = bind(c); if (rc) return rc;
    if (!g1 || !g2 || !out576) return C12381_E_ARG;
    if (n == 0) return 0;
    staged s;
    if ((rc = stage_in(c, s, g1, 96 * n, g2, 192 * n, 576 * n))) return rc;
    hipLaunchKernelGGL(miller_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, c->stream, n, s.in0, s.in1, s.out, c->d_flag);
    HIPCK(c, hipGetLastError());
    if ((rc = stage_out(c, s, out576, 576 * n))) return rc;
    return read_flag(c);
}
int c12381_gt_op_batch(c12381_ctx* c, int op, size_t n, const uint8_t* a576, const uint8_t* b, uint8_t* out576) {
    int rc = bind(c); if (rc) return rc;
    if (op < 0 || op > 3 || !a576 || !out576 || ((op == 0 || op == 2) && !b)) return C12381_E_ARG;
    if (n == 0) return 0;
    staged s;
    const size_t bb = op == 0 ? 576 * n : (op == 2 ? 32 * n : 0);
    if ((rc = stage_in(c, s, a576, 576 * n, bb ? b : nullptr, bb, 576 * n))) return rc;
    hipLaunchKernelGGL(gt_op_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, c->stream, op, n, s.in0, s.in1, s.out);
    HIPCK(c, hipGetLastError());
    if ((rc = stage_out(c, s, out576, 576 * n))) return rc;
    HIPCK(c, hipStreamSynchronize(c->stream));
    return 0;
}
int c12381_fexp_batch(c12381_ctx* c, size_t n, const uint8_t* in576, uint8_t* out576) { return c12381_gt_op_batch(c, 3, n, in576, nullptr, out576); }
int c12381_gt_is_unity_batch(c12381_ctx* c, size_t n, const uint8_t* a576, uint8_t* out) {
    int rc = bind(c); if (rc) return rc;
    if (!a576 || !out) return C12381_E_ARG;
    if (n == 0) return 0;
    staged s;
    if ((rc = stage_in(c, s, a576, 576 * n, nullptr, 0, n))) return rc;
    hipLaunchKernelGGL(gt_is_unity_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, c->stream, n, s.in0, s.out);
    HIPCK(c, hipGetLastError());
    if ((rc = stage_out(c, s, out, n))) return rc;
    HIPCK(c, hipStreamSynchronize(c->stream));
    return 0;
}

// ---------------------------------------------------------------- BBS+ batch verification
// ok[j] = [ e(A_j, w + x_j g2) == e(g1 + r_j h0 + sum_i m_{i,j} h_i, g2) ]   — the verification equation of the
// reference's examples/bbs-plus/src/bbs+.cpp:57-73, evaluated as liner_pair.hpp:339-350 does (two Miller loops,
// one final exponentiation).  Message scalars are message-major: m[i*n + j] belongs to signature j.  All
// pointers are DEVICE pointers; the public parameters are single points.
int c12381_bbs_plus_verify_batch_dev(c12381_ctx* c, size_t n, size_t nmsg, const uint8_t* g1_96, const uint8_t* g2_192, const uint8_t* h0_96,
                                     const uint8_t* h_96, const uint8_t* w_192, const uint8_t* A_96, const uint8_t* x_32, const uint8_t* r_32,
                                     const uint8_t* m_32, uint8_t* ok) {
    int rc = bind(c); if (rc) return rc;
    if (!g1_96 || !g2_192 || !h0_96 || !w_192 || !A_96 || !x_32 || !r_32 || !ok || (nmsg && (!h_96 || !m_32))) return C12381_E_ARG;
    if (n == 0) return 0;
    // Q_j = w + x_j g2
    if ((rc = ensure(c, c12381_ctx::WS_BBS_Q, 192 * n))) return rc;
    if ((rc = ensure(c, c12381_ctx::WS_BBS_B, 192 * n))) return rc;
    uint8_t* d_q = (uint8_t*)c->ws[c12381_ctx::WS_BBS_Q];
    uint8_t* d_b = (uint8_t*)c->ws[c12381_ctx::WS_BBS_B];
    if ((rc = g2_mul_dev_strided(c, n, g2_192, 0, x_32, d_b, 192))) return rc;
    hipLaunchKernelGGL(g2_add_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, c->stream, n, w_192, (size_t)0, d_b, d_q, 192, c->d_flag);
    HIPCK(c, hipGetLastError());
    // B_j = g1 + r_j h0 + sum_i m_ij h_i : (nmsg + 1) fixed-base columns of n scalar multiplications, summed per lane
    const size_t cols = nmsg + 1, total = cols * n, stride = round_up(total, 64);
    if ((rc = g1_mul_to_proj(c, n, h0_96, r_32, stride, 0, 0))) return rc;
    for (size_t i = 0; i < nmsg; ++i)
        if ((rc = g1_mul_to_proj(c, n, h_96 + 96 * i, m_32 + 32 * n * i, stride, 0, (i + 1) * n))) return rc;
    const size_t rstride = round_up(n, 64);
    if ((rc = ensure(c, c12381_ctx::WS_RED0, (size_t)3 * NL * rstride * 4))) return rc;
    int32_t* red = (int32_t*)c->ws[c12381_ctx::WS_RED0];
    hipLaunchKernelGGL(g1_reduce_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, c->stream, total, (const int32_t*)c->ws[c12381_ctx::WS_PROJ], stride, n, red, rstride);
    HIPCK(c, hipGetLastError());
    hipLaunchKernelGGL(g1_add_const_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, c->stream, n, red, rstride, g1_96, c->d_flag);
    HIPCK(c, hipGetLastError());
    if ((rc = g1_finish(c, n, red, rstride, d_b, 96))) return rc;
    timed tm(c, 4);
    if (pair_lanes() == 1) hipLaunchKernelGGL(pair_eq_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, c->stream, n, A_96, d_q, d_b, g2_192, (size_t)0, ok, c->d_flag);
    else hipLaunchKernelGGL(pair3_eq_kernel, dim3(grid_tri(n)), dim3(BLOCK), 0, c->stream, n, A_96, d_q, d_b, g2_192, (size_t)0, ok, c->d_flag);
    HIPCK(c, hipGetLastError());
    return 0;
}
int c12381_bbs_plus_verify_batch(c12381_ctx* c, size_t n, size_t nmsg, const uint8_t* g1_96, const uint8_t* g2_192, const uint8_t* h0_96,
                                 const uint8_t* h_96, const uint8_t* w_192, const uint8_t* A_96, const uint8_t* x_32, const uint8_t* r_32,
                                 const uint8_t* m_32, uint8_t* ok) {
    int rc = bind(c); if (rc) return rc;
    if (!g1_96 || !g2_192 || !h0_96 || !w_192 || !A_96 || !x_32 || !r_32 || !ok || (nmsg && (!h_96 || !m_32))) return C12381_E_ARG;
    if (n == 0) return 0;
    // one staging slab: public parameters, then the per-signature arrays
    const size_t o_g1 = 0, o_g2 = 96, o_h0 = 288, o_w = 384, o_h = 576, o_A = round_up(o_h + 96 * nmsg, 256), o_x = o_A + 96 * n,
                 o_r = o_x + 32 * n, o_m = o_r + 32 * n, o_ok = round_up(o_m + 32 * n * nmsg, 256), bytes = o_ok + round_up(n, 256);
    if ((rc = ensure(c, c12381_ctx::WS_BBS_IN, bytes))) return rc;
    uint8_t* d = (uint8_t*)c->ws[c12381_ctx::WS_BBS_IN];
    HIPCK(c, hipMemcpyAsync(d + o_g1, g1_96, 96, hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipMemcpyAsync(d + o_g2, g2_192, 192, hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipMemcpyAsync(d + o_h0, h0_96, 96, hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipMemcpyAsync(d + o_w, w_192, 192, hipMemcpyHostToDevice, c->stream));
    if (nmsg) HIPCK(c, hipMemcpyAsync(d + o_h, h_96, 96 * nmsg, hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipMemcpyAsync(d + o_A, A_96, 96 * n, hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipMemcpyAsync(d + o_x, x_32, 32 * n, hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipMemcpyAsync(d + o_r, r_32, 32 * n, hipMemcpyHostToDevice, c->stream));
    if (nmsg) HIPCK(c, hipMemcpyAsync(d + o_m, m_32, 32 * n * nmsg, hipMemcpyHostToDevice, c->stream));
    if ((rc = c12381_bbs_plus_verify_batch_dev(c, n, nmsg, d + o_g1, d + o_g2, d + o_h0, d + o_h, d + o_w, d + o_A, d + o_x, d + o_r, d + o_m, d + o_ok))) return rc;
    HIPCK(c, hipMemcpyAsync(ok, d + o_ok, n, hipMemcpyDeviceToHost, c->stream));
    return read_flag(c);
}

}  // extern "C"
