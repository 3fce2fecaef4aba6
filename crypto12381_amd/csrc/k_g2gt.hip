// Kernels of the G2 / GT family, one element per lane:
//   g2_mul_kernel      bytes -> on-twist check -> windowed [k]Q -> affine -> 97 B / 192 B
//   g2_add_kernel      complete addition of two affine G2 inputs
//   g2_decompress_kernel
//   (experiments builds only) pair_kernel / pair_eq_kernel / miller_kernel / gt_op_kernel / gt_is_unity_kernel: the one-lane pairing
#include "kernels_common.hpp"

using namespace c12381;

namespace c12381 {

__global__ void __launch_bounds__(BLOCK, 2) g2_mul_kernel(size_t n, const uint8_t* pts, size_t pt_stride, const uint8_t* scalars, int32_t* tab,
                                                       size_t tab_stride, uint8_t* out, int fmt, int* bad_flag, const int32_t* skip_if,
                                                       int32_t* proj, size_t proj_stride, size_t proj_off, int in_g2) {
    if (skip_if && skip_if[HDR_VALID] != 0) return;          // served by a valid fixed-base table (k_fixed.hip)
    const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    fp2 qx, qy;
    bool inf, ok;
    g2_parse_any(qx, qy, inf, ok, pts, pt_stride, i);
    uint32_t raw[8], k[8];
    load_raw32(raw, scalars + 32 * i);
    scalar_from_raw32(k, raw);
    g2p acc;
    g2_scalar_mul(acc, qx, qy, inf || !ok, k, tab + i * (size_t)G2_TAB_DWORDS, in_g2 != 0);
    if (!ok) *bad_flag = 1;
    if (proj) g2_store_proj(proj, proj_stride, proj_off + i, acc, !ok);      // kernel-uniform: affine conversion by g2_finish_kernel
    else g2_store_affine(out + (size_t)fmt * i, acc, fmt, !ok);
}

// Simultaneous inversion (Montgomery's trick in Fp2) + affine + encode, as g1_finish_kernel: lane t owns elements
// t, t + T, t + 2T, ... — one Fp2 inversion per lane instead of one per element (an inversion is ~11 % of a G2
// scalar multiplication).  X = 1, Z = 0 marks an invalid input (all-0xff output), X = 0, Z = 0 is infinity.
__global__ void __launch_bounds__(BLOCK, 2) g2_finish_kernel(size_t n, const int32_t* proj, size_t stride, int32_t* pref, uint8_t* out, int fmt, size_t T) {
    const size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (t >= T || t >= n) return;
    const int32_t* zbase = proj + (size_t)4 * NL * stride;
    fp2 run, one;
    fp2_one(one);
    run = one;
    size_t last = t;
#pragma unroll 1
    for (size_t e = t; e < n; e += T) {
        fp2 z;
        soa_load_fp2(z, zbase, stride, e);
        const bool inf = fp2_is_zero(z);
        fp2_select(z, inf, one, z);
        fp2_mul(run, run, z);
        soa_store_fp2(pref, stride, e, run);
        last = e;
    }
    fp2 inv;
    fp2_inv(inv, run);
#pragma unroll 1
    for (size_t e = last;; e -= T) {
        g2p p;
        soa_load_g2(p, proj, stride, e);
        const bool inf = fp2_is_zero(p.z);
        const bool invalid = inf && !fp2_is_zero(p.x);
        fp2_select(p.z, inf, one, p.z);
        fp2 prev, zinv, ax, ay, ninv;
        if (e >= T + t) soa_load_fp2(prev, pref, stride, e - T); else prev = one;
        fp2_mul(zinv, inv, prev);
        fp2_mul(ninv, inv, p.z);
        inv = ninv;
        fp2_mul(ax, p.x, zinv); fp2_mul(ay, p.y, zinv);
        g2_store_xy(out + (size_t)fmt * e, ax, ay, fmt, inf, invalid);
        if (e < T + t) break;
    }
}

// ---- product of G2 points (g2_point.hpp:225-236: the header's product over G2Point is a chain of add(point2&, point2&);
// with eager exponents, Π q_i^{x_i} is n multiply() calls followed by that chain).  g2_lift_kernel turns affine inputs into the
// projective SoA of the scalar-multiplication kernels; g2_reduce_kernel is one level of the tree sum (lane j adds elements
// j, j + m, ...); an element that is not on the twist marks the whole product invalid (X = 1, Z = 0, as g2_finish_kernel reads it).
__global__ void __launch_bounds__(BLOCK, 2) g2_lift_kernel(size_t n, const uint8_t* pts, int32_t* proj, size_t stride, int* bad_flag) {
    const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    g2p p, inf_pt;
    bool inf, ok;
    g2_parse192(p.x, p.y, inf, ok, pts + 192 * i);
    fp2_one(p.z);
    g2_set_inf(inf_pt);
    fp2_select(p.x, inf, inf_pt.x, p.x); fp2_select(p.y, inf, inf_pt.y, p.y); fp2_select(p.z, inf, inf_pt.z, p.z);
    if (!ok) *bad_flag = 1;
    g2_store_proj(proj, stride, i, p, !ok);
}
__global__ void __launch_bounds__(BLOCK, 2) g2_reduce_kernel(size_t n, const int32_t* in, size_t in_stride, size_t m, int32_t* outp, size_t out_stride) {
    const size_t j = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (j >= m) return;
    g2p acc;
    g2_set_inf(acc);
    bool bad = false;
#pragma unroll 1
    for (size_t i = j; i < n; i += m) {
        g2p q, nn;
        soa_load_g2(q, in, in_stride, i);
        bad = bad || (fp2_is_zero(q.z) && !fp2_is_zero(q.x));
        g2_add(acc, q);
        g2_norm1(nn, acc);
        acc = nn;
    }
    g2_store_proj(outp, out_stride, j, acc, bad);
}

__global__ void __launch_bounds__(BLOCK, 2) g2_add_kernel(size_t n, const uint8_t* a, size_t a_stride, const uint8_t* b, uint8_t* out, int fmt,
                                                       int* bad_flag, const int32_t* skip_if) {
    if (skip_if && skip_if[HDR_VALID] != 0) return;
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

#ifdef C12381_EXPERIMENTS
// Diagnostic: the clock the chip holds while another kernel runs.  ONE lane samples (s_memtime = shader cycles, s_memrealtime = 100 MHz)
// every `gap` sleeps of 127 x 64 cycles and is launched on the context's side stream beside the kernel under study
// (c12381_exp_clock_probe, tools/clock_probe.py): clock = d(memtime) / d(memrealtime) x 100 MHz (MI355X_MICROARCH.md, DVFS item 6).
__global__ void __launch_bounds__(BLOCK, 2) clock_probe_kernel(unsigned long long* out, int n, int gap) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    for (int i = 0; i < n; ++i) {
        out[2 * i] = __builtin_amdgcn_s_memtime();
        out[2 * i + 1] = __builtin_amdgcn_s_memrealtime();
        for (int j = 0; j < gap; ++j) __builtin_amdgcn_s_sleep(127);
    }
}

// The first design — one pairing per lane (pairing.hpp) — superseded by the three-lane kernels of k_pair3.hip; kept in experiments
// builds as an independent implementation for whole-batch cross-checks (tests/test_gpu_variants.py, C12381_PAIR_LANES=1).
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

#endif  // C12381_EXPERIMENTS

// ECP2_fromOctet ecp2_BLS12381.cpp:225-266 for 97-byte input: any tag other than 04 is "compressed, sign = tag & 1"
// mark_invalid: as g1_decompress_kernel — a rejected lane becomes the off-twist record x = 0, y = 1
__global__ void __launch_bounds__(BLOCK, 2) g2_decompress_kernel(size_t n, const uint8_t* in, uint8_t* out, uint8_t* status, int mark_invalid) {
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
    else {
        uint4* q = reinterpret_cast<uint4*>(o);
        for (int j = 0; j < 12; ++j) q[j] = make_uint4(0, 0, 0, 0);
        if (mark_invalid && tag != 0) q[11] = make_uint4(0, 0, 0, 0x01000000u);       // y.a = 1 (layout x.b | x.a | y.b | y.a)
    }
    if (status) status[i] = tag == 0 ? 1 : (ok ? 1 : 0);
}

#ifdef C12381_EXPERIMENTS
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

#endif  // C12381_EXPERIMENTS

}  // namespace c12381
