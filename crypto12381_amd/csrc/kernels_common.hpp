// Device-side helpers shared by the kernel translation units: byte loads/stores, input parsing and validation,
// canonical output encodings.  Everything lives in an anonymous namespace: each .hip file gets its own copy.
#pragma once
#include <hip/hip_runtime.h>

#include "codec.hpp"
#include "fp.hpp"
#include "g1.hpp"
#include "g2.hpp"
#include "pairing.hpp"
#include "kernels.hpp"

using namespace c12381;

namespace {

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
// The 49-byte form as the header layer reads it (g1_point.hpp:87-111 in front of ECP_fromOctet ecp_BLS12381.cpp:495-545): a leading
// 0x00 is the point at infinity, 0x02 / 0x03 carry x and the parity of y (ECP_setx: one square root, on-curve by construction, no
// subgroup check), every other tag is rejected (ok = false: the lane is treated like a point that is not on the curve).
// Out of line: the square root must not share the register allocation of the caller's main loop.
__device__ __noinline__ void g1_parse49(fp& x, fp& y, bool& inf, bool& ok, const uint8_t* sp) {
    const uint8_t tag = sp[0];
    uint32_t raw[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) raw[j] = (uint32_t)sp[1 + 4 * j] | ((uint32_t)sp[2 + 4 * j] << 8) | ((uint32_t)sp[3 + 4 * j] << 16) | ((uint32_t)sp[4 + 4 * j] << 24);
    fp_from_raw48(x, raw);
    inf = tag == 0;
    const bool root = g1_set_x(y, x, tag & 1);
    ok = inf || (root && (tag == 2 || tag == 3));
    if (inf) { fp_zero(x); fp_zero(y); }
}
// pt_stride selects the input format of the scalar-multiplication kernels: 96 = affine records, 0 = one affine point for every lane,
// 49 = compressed records (C12381_F_COMPRESSED_IN)
__device__ __forceinline__ void g1_parse_any(fp& x, fp& y, bool& inf, bool& ok, const uint8_t* pts, size_t pt_stride, size_t i) {
    if (pt_stride == 49) g1_parse49(x, y, inf, ok, pts + 49 * i);
    else g1_parse96(x, y, inf, ok, pts + pt_stride * i);
}
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
// The 97-byte form (g2_point.hpp:73-77 in front of ECP2_fromOctet ecp2_BLS12381.cpp:225-266): leading 0x00 = infinity; any tag other
// than 0x04 is "compressed, sign = tag & 1" (ECP2_setx: an Fp2 square root); 0x04 announces the 193-byte form and is rejected here.
__device__ __noinline__ void g2_parse97(fp2& x, fp2& y, bool& inf, bool& ok, const uint8_t* sp) {
    const uint8_t tag = sp[0];
    uint32_t raw[24];
#pragma unroll
    for (int j = 0; j < 24; ++j) raw[j] = (uint32_t)sp[1 + 4 * j] | ((uint32_t)sp[2 + 4 * j] << 8) | ((uint32_t)sp[3 + 4 * j] << 16) | ((uint32_t)sp[4 + 4 * j] << 24);
    fp_from_raw48(x.b, raw); fp_from_raw48(x.a, raw + 12);
    inf = tag == 0;
    const bool root = g2_set_x(y, x, tag & 1);
    ok = inf || (root && tag != 4);
    if (inf) { fp2_zero(x); fp2_zero(y); }
}
__device__ __forceinline__ void g2_parse_any(fp2& x, fp2& y, bool& inf, bool& ok, const uint8_t* pts, size_t pt_stride, size_t i) {
    if (pt_stride == 97) g2_parse97(x, y, inf, ok, pts + 97 * i);
    else g2_parse192(x, y, inf, ok, pts + pt_stride * i);
}
// canonical encoding of one affine G2 point (or the infinity / invalid patterns)
__device__ __noinline__ void g2_store_xy(uint8_t* o, const fp2& ax, const fp2& ay, int fmt, bool inf, bool invalid) {
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
// affine + canonical encoding of one projective G2 point (per-lane inversion; the batch entry points use g2_finish_kernel)
__device__ __noinline__ void g2_store_affine(uint8_t* o, const g2p& acc, int fmt, bool invalid) {
    const bool inf = fp2_is_zero(acc.z);
    fp2 zn, zi, ax, ay, one;
    fp2_one(one);
    fp2_norm1(zn, acc.z);
    fp2_select(zn, inf, one, zn);
    fp2_inv(zi, zn);
    fp2_mul(ax, acc.x, zi); fp2_mul(ay, acc.y, zi);
    g2_store_xy(o, ax, ay, fmt, inf, invalid);
}
// result of one lane for g2_finish_kernel: projective SoA; an invalid input is marked X = 1, Y = Z = 0
__device__ __forceinline__ void g2_store_proj(int32_t* proj, size_t stride, size_t i, const g2p& acc, bool invalid) {
    g2p o;
    g2_norm1(o, acc);
    if (invalid) { fp2_one(o.x); fp2_zero(o.y); fp2_zero(o.z); }
    soa_store_g2(proj, stride, i, o);
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
__device__ __noinline__ void gt_load576(fp12& f, const uint8_t* p) {
#pragma unroll 1
    for (int j = 0; j < 12; ++j) {
        uint32_t raw[12];
        load_raw48(raw, p + 48 * j);
        fp_from_raw48(fp12_coord_mut(f, j), raw);
    }
}

}  // namespace
