// Kernels of the Fp / G1 family (one element per lane):
//   fp_op_kernel / fp_mulchain_kernel   Fp test + VALU-roofline hook
//   g1_mul_kernel      bytes -> on-curve check -> GLV windowed [k]P -> projective SoA in HBM
//   g1_add_kernel      complete addition of two affine inputs -> projective SoA
//   g1_finish_kernel   Montgomery's simultaneous inversion over a strided chunk per lane,
//                      affine conversion, canonical encoding (49 B / 96 B)
//   g1_reduce_kernel   tree sum of projective points (MSM combine)
//   msm_*_kernel       bucket-method MSM stages (msm.hpp): prep, ranges, sizes (+ overflow bookkeeping), bucket, overflow,
//                      overflow_combine, wreduce, small_term, horner
//   g1_decompress_kernel   49-byte -> 96-byte decoding with the reference's acceptance rules
#include "kernels_common.hpp"
#include "msm.hpp"

using namespace c12381;

namespace c12381 {

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

// proj layout: coordinate-major, limb-major SoA: proj[(c*NL + limb) * stride + element]
// pt_stride = 96 for per-lane points, 0 to broadcast one point to every lane (fixed-base columns of BBS+)
__global__ void __launch_bounds__(BLOCK, G1_OCC) g1_mul_kernel(size_t n, const uint8_t* pts, size_t pt_stride, const uint8_t* scalars, int32_t* tab,
                                                       int32_t* proj, size_t proj_stride, size_t proj_off, int* bad_flag, const int32_t* skip_if,
                                                       int small_term) {
    if (skip_if && skip_if[HDR_VALID] != 0) return;          // this column is served by a valid fixed-base table (k_fixed.hip)
    const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    fp px, py;
    bool inf, ok;
    g1_parse_any(px, py, inf, ok, pts, pt_stride, i);
    uint32_t raw[8], k[8];
    load_raw32(raw, scalars + 32 * i);
    scalar_from_raw32(k, raw);
    g1p acc;
    g1_scalar_mul(acc, px, py, inf || !ok, k, tab + i * (size_t)G1_TAB_DWORDS);
    // the reference's [r]phi(P) term of scalars below x^2 (g1.hpp): no lane of a batch of random scalars, and then one that keeps its
    // wavefront for the length of a second scalar multiplication.  In a launch of several machine rounds that wavefront's workgroup
    // simply leaves later while others take the free slots; a separate fix-up launch behind the kernel (rounds 1-3) costs its full
    // single-wavefront latency, 1.3 ms, whenever one lane of 2^20 owes the term.
    if (small_term && ok && !inf && scalar_below_x2(k)) {
        g1p base, nn;
        base.x = px; base.y = py; fp_one(base.z);
        g1_norm1(nn, acc);
        g1_glv_small_scalar_term(nn, base);
        acc = nn;
    }
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

// affine inputs -> the projective SoA of the reductions (the plain sum of points: product(type_identity<G1Point>, r) g1_point.hpp — a chain
// of add(point1&, point1&) — and the combine step of a sharded product).  A point that is not on the curve is reported and left out.
__global__ void __launch_bounds__(BLOCK, 2) g1_lift_kernel(size_t n, const uint8_t* pts, int32_t* proj, size_t stride, int* bad_flag) {
    const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    g1p p, inf_pt, o;
    bool inf, ok;
    g1_parse96(p.x, p.y, inf, ok, pts + 96 * i);
    fp_one(p.z);
    g1_set_inf(inf_pt);
    const bool drop = inf || !ok;
    fp_select(p.x, drop, inf_pt.x, p.x); fp_select(p.y, drop, inf_pt.y, p.y); fp_select(p.z, drop, inf_pt.z, p.z);
    if (!ok) *bad_flag = 1;
    g1_norm1(o, p);
    soa_store_g1(proj, stride, i, o);
}

// [k mod r]P by plain double-and-add on the complete formulas, no endomorphism: the TRUE multiple for every curve point — the term
// of the boundary's sum_of_products (-> ECP_muln ecp_BLS12381.cpp:1112-1148), which multiply()'s GLV form equals only on G1.
// Uniform schedule: 255 doublings, every addition executed with the point or infinity selected by the bit.
__global__ void __launch_bounds__(BLOCK, 2) g1_mul_plain_kernel(size_t n, const uint8_t* pts, const uint8_t* scalars, int32_t* proj, size_t proj_stride, int* bad_flag) {
    const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    g1p base, acc, inf_pt;
    bool inf, ok;
    g1_parse96(base.x, base.y, inf, ok, pts + 96 * i);
    fp_one(base.z);
    g1_set_inf(inf_pt);
    if (inf || !ok) base = inf_pt;
    uint32_t raw[8], k[8];
    load_raw32(raw, scalars + 32 * i);
    scalar_from_raw32(k, raw);
    scalar_mod_r(k);
    acc = inf_pt;
#pragma unroll 1
    for (int b = 254; b >= 0; --b) {
        g1_dbl(acc);
        g1p nn, q;
        g1_norm1(nn, acc);
        const bool bit = ((k[b >> 5] >> (b & 31)) & 1u) != 0;
        fp_select(q.x, bit, base.x, inf_pt.x); fp_select(q.y, bit, base.y, inf_pt.y); fp_select(q.z, bit, base.z, inf_pt.z);
        g1_add(nn, q);
        g1_norm1(acc, nn);
    }
    if (!ok) { *bad_flag = 1; fp_one(acc.x); fp_zero(acc.y); fp_zero(acc.z); }
    soa_store_g1(proj, proj_stride, i, acc);
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

// acc[i] <- other[off + i] - acc[i] on projective SoA arrays (BBS+ with fixed G2 arguments: x A - B); runs only when
// run_if[HDR_VALID] != 0
__global__ void __launch_bounds__(BLOCK, 2) g1_rsub_kernel(size_t n, int32_t* acc, size_t acc_stride, const int32_t* other, size_t other_stride,
                                                        size_t other_off, const int32_t* run_if) {
    if (run_if[HDR_VALID] == 0) return;
    const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    g1p a, b, o;
    soa_load_g1(a, acc, acc_stride, i);
    soa_load_g1(b, other, other_stride, other_off + i);
    fp ny;
    fp_neg(ny, a.y);
    fp_norm1(a.y, ny);
    g1_add(b, a);
    g1_norm1(o, b);
    soa_store_g1(acc, acc_stride, i, o);
}

// in_fmt: 96 = affine records, 49 = compressed records (C12381_F_COMPRESSED_IN: decoded here, one square root per term; a rejected
// encoding becomes the off-curve record (0, 1), which msm_prep_one reports and leaves out of the product like any invalid point)
template <class K>
static __device__ __forceinline__ void msm_prep_body(size_t n, const uint8_t* pts, int in_fmt, const uint8_t* scalars, int c, int W, int32_t* pts2,
                                                     K* keys, uint32_t* vals, int* bad_flag) {
    const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    uint32_t rp[24], rs[8];
    if (in_fmt == 49) {
        fp x, y; bool inf, ok;
        g1_parse49(x, y, inf, ok, pts + 49 * i);
        if (inf || !ok) { for (int j = 0; j < 24; ++j) rp[j] = 0; if (!ok) rp[23] = 0x01000000u; }      // infinity | (0, 1)
        else { fp_to_raw48(rp, x); fp_to_raw48(rp + 12, y); }
    } else { load_raw48(rp, pts + 96 * i); load_raw48(rp + 12, pts + 96 * i + 48); }
    load_raw32(rs, scalars + 32 * i);
    if (!msm_prep_one<K>(i, n, rp, rs, c, W, pts2, keys, vals)) *bad_flag = 1;
}
__global__ void __launch_bounds__(BLOCK, 2) msm_prep_kernel(size_t n, const uint8_t* pts, int in_fmt, const uint8_t* scalars, int c, int W, int32_t* pts2,
                                                         uint32_t* keys, uint32_t* vals, int* bad_flag) {
    msm_prep_body<uint32_t>(n, pts, in_fmt, scalars, c, W, pts2, keys, vals, bad_flag);
}
// large products: 16-bit digit keys, no value array (msm_entry_value)
__global__ void __launch_bounds__(BLOCK, 2) msm_prep16_kernel(size_t n, const uint8_t* pts, int in_fmt, const uint8_t* scalars, int c, int W, int32_t* pts2,
                                                           uint16_t* keys, int* bad_flag) {
    msm_prep_body<uint16_t>(n, pts, in_fmt, scalars, c, W, pts2, keys, nullptr, bad_flag);
}
__global__ void __launch_bounds__(BLOCK, 2) msm_ranges_kernel(size_t E, const uint32_t* keys, int c, int W, uint32_t* lo, uint32_t* hi) {
    const size_t j = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (j >= E) return;
    msm_ranges_one(j, E, keys, c, W, lo, hi);
}
// digit-only keys: blockIdx.y = the window segment (W = the small-scalar segment of n entries)
// MSM_RANGES_PER_THREAD consecutive entries per thread, all their keys requested before the first is looked at: with one entry per thread the
// kernel was bound by the latency of one 2-byte load per wavefront of work (220 us for 2^26 entries)
__global__ void __launch_bounds__(BLOCK, 2) msm_ranges16_kernel(size_t n, const uint16_t* keys, int c, int W, uint32_t* lo, uint32_t* hi) {
    constexpr int T = MSM_RANGES_PER_THREAD;
    const uint32_t w = blockIdx.y;
    const uint32_t len = (uint32_t)(w < (uint32_t)W ? 2 * n : n);
    const uint32_t x0 = (blockIdx.x * BLOCK + threadIdx.x) * (uint32_t)T;
    if (x0 >= len) return;
    const size_t seg = (size_t)2 * w * n;
    uint32_t k[T + 2];                                    // keys of x0 - 1 .. x0 + T; 0x10000 = no such entry (never equal to a key)
#pragma unroll
    for (int j = 0; j < T + 2; ++j) {
        const uint32_t x = x0 + (uint32_t)j;              // entry x - 1
        k[j] = (x >= 1 && x - 1 < len) ? (uint32_t)keys[seg + x - 1] : 0x10000u;
    }
#pragma unroll
    for (int j = 0; j < T; ++j) {
        const uint32_t x = x0 + (uint32_t)j, d = k[j + 1];
        if (x >= len || d == 0) continue;
        const uint32_t b = w < (uint32_t)W ? ((w << c) | d) : ((uint32_t)W << c);
        if (k[j] != d) lo[b] = (uint32_t)(seg + x);
        if (k[j + 2] != d) hi[b] = (uint32_t)(seg + x + 1);
    }
}
// the last bucket (nbk - 1: the small-scalar bucket) counts as empty while it has at most `early_max` entries: msm_small_early_kernel sums it
__global__ void __launch_bounds__(BLOCK, 2) msm_sizes_kernel(size_t nbk, const uint32_t* lo, const uint32_t* hi, uint32_t* key, uint32_t* ident,
                                                          uint32_t cap, uint32_t* cnt, uint2* seg, uint4* big, uint32_t early_max) {
    const size_t b = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (b >= nbk) return;
    uint32_t len = hi[b] - lo[b];
    if (b == nbk - 1 && len <= early_max) len = 0;
    key[b] = cap - (len < cap ? len : cap);              // 0 = longest: ascending order of this key is decreasing run length; < 2^bits(cap)
    ident[b] = (uint32_t)b;
    if (len > cap) {
        const uint32_t sl = cap / 2, ns = (len - cap + sl - 1) / sl;
        const uint32_t base = atomicAdd(&cnt[0], ns);
        const uint32_t q = atomicAdd(&cnt[1], 1u);
        big[q] = make_uint4((uint32_t)b, base, ns, 0u);
        for (uint32_t s = 0; s < ns; ++s) seg[base + s] = make_uint2((uint32_t)b, s);
    }
}
__global__ void __launch_bounds__(BLOCK, MSM_OCC) msm_bucket_kernel(size_t nbk, const uint32_t* lo, const uint32_t* hi, const uint32_t* vals,
                                                           const int32_t* pts2, int32_t* bk, const uint32_t* order, uint32_t cap, uint32_t early_max) {
    const size_t slot = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (slot >= nbk) return;
    const size_t b = order[slot];
    g1p acc, nn;
    const size_t l = lo[b];
    size_t h = hi[b];
    if (b == nbk - 1 && h - l <= early_max) h = l;                 // summed by msm_small_early_kernel (see msm_sizes_kernel)
    msm_bucket_one(acc, l, h - l > cap ? l + cap : h, vals, pts2);
    g1_norm1(nn, acc);
    tab_store_g1(bk + b * G1_ENT_DWORDS, nn);
}
// partial sum of overflow segment q (entries [lo + cap + s seg, lo + cap + (s + 1) seg) of its run); the grid covers the
// capacity of the segment list, wavefronts beyond the registered count leave at once.  The 64 segments of a wavefront are
// consecutive list entries — for a long run: of the same bucket — so the wavefront adds them up before storing
// (segmented suffix sums by bucket over six shuffle steps): only the first lane of each run of equal buckets writes, at its
// own list position.  A bucket's partial sums then sit at its first list position and at every later multiple of 64.
__global__ void __launch_bounds__(BLOCK, 2) msm_overflow_kernel(const uint32_t* cnt, const uint2* seg, const uint32_t* lo, const uint32_t* hi,
                                                             const uint32_t* vals, const int32_t* pts2, int32_t* part, uint32_t cap) {
    const size_t q = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t total = cnt[0];
    if (q - lane >= total) return;                                 // the whole wavefront is beyond the list
    const bool valid = q < total;
    uint32_t bucket = 0xffffffffu;
    g1p acc, nn, t, cand;
    g1_set_inf(acc);
    if (valid) {
        const uint2 e = seg[q];
        bucket = e.x;
        const size_t sl = cap / 2, start = (size_t)lo[e.x] + cap + (size_t)e.y * sl, h = hi[e.x];
        msm_bucket_one(acc, start, h - start > sl ? start + sl : h, vals, pts2);
    }
    g1_norm1(nn, acc); acc = nn;
#pragma unroll 1
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t ob = (uint32_t)__shfl_down((int)bucket, off, 64);
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            t.x.l[j] = __shfl_down(acc.x.l[j], off, 64); t.y.l[j] = __shfl_down(acc.y.l[j], off, 64); t.z.l[j] = __shfl_down(acc.z.l[j], off, 64);
        }
        cand = acc;
        g1_add(cand, t);
        g1_norm1(nn, cand);
        const bool take = lane + (uint32_t)off < 64u && ob == bucket;
        fp_select(acc.x, take, nn.x, acc.x); fp_select(acc.y, take, nn.y, acc.y); fp_select(acc.z, take, nn.z, acc.z);
    }
    const uint32_t left = (uint32_t)__shfl_up((int)bucket, 1, 64);
    if (valid && (lane == 0 || left != bucket)) tab_store_g1(part + q * G1_ENT_DWORDS, acc);
}
// bucket b += its overflow partial sums (at list position `base` and at every multiple of 64 inside (base, base + ns), see
// above): one wavefront per cut bucket, lanes take them strided, then six shuffle-and-add steps; wavefronts stride over the
// list of cut buckets
__global__ void __launch_bounds__(BLOCK, 2) msm_overflow_combine_kernel(const uint32_t* cnt, const uint4* big, const int32_t* part, int32_t* bk) {
    const uint32_t lane = threadIdx.x & 63u;
    const size_t wave = ((size_t)blockIdx.x * BLOCK + threadIdx.x) >> 6, nwaves = ((size_t)gridDim.x * BLOCK) >> 6;
    const uint32_t nbig = cnt[1];
#pragma unroll 1
    for (size_t q = wave; q < nbig; q += nwaves) {                          // wave-uniform
        const uint4 e = big[q];
        const uint32_t base = e.y, end = e.y + e.z, first = (base / 64u + 1u) * 64u;
        const uint32_t heads = 1u + (end > first ? (end - 1u - first) / 64u + 1u : 0u);
        g1p acc, t, nn;
        g1_set_inf(acc);
#pragma unroll 1
        for (uint32_t j0 = 0; j0 < heads; j0 += 64) {
            const uint32_t j = j0 + lane;
            if (j < heads) {
                const uint32_t pos = j == 0 ? base : first + 64u * (j - 1u);
                tab_load_g1(t, part + (size_t)pos * G1_ENT_DWORDS);
                g1_add(acc, t);
                g1_norm1(nn, acc); acc = nn;
            }
        }
        if (lane == 0) {
            tab_load_g1(t, bk + (size_t)e.x * G1_ENT_DWORDS);
            g1_add(acc, t);
            g1_norm1(nn, acc); acc = nn;
        }
#pragma unroll 1
        for (int off = 32; off >= 1; off >>= 1) {
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                t.x.l[i] = __shfl_down(acc.x.l[i], off, 64); t.y.l[i] = __shfl_down(acc.y.l[i], off, 64); t.z.l[i] = __shfl_down(acc.z.l[i], off, 64);
            }
            g1_add(acc, t);
            g1_norm1(nn, acc); acc = nn;
        }
        if (lane == 0) tab_store_g1(bk + (size_t)e.x * G1_ENT_DWORDS, acc);
    }
}

__global__ void __launch_bounds__(BLOCK, 2) msm_wreduce_kernel(int W, uint32_t nb, uint32_t chunks, const int32_t* bk, int32_t* out, size_t out_stride) {
    const size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (t >= (size_t)W * chunks) return;
    const uint32_t w = (uint32_t)(t / chunks), ch = (uint32_t)(t % chunks);
    g1p part;
    msm_wreduce_one(part, bk + (size_t)w * nb * G1_ENT_DWORDS, ch * MSM_CHUNK, nb);
    soa_store_g1(out, out_stride, (size_t)ch * W + w, part);
}

// ECP_fromOctet ecp_BLS12381.cpp:495-545 for 49-byte input (tags 02/03; a leading 00 is infinity as in
// g1_point.hpp:89-93); status 1 ok / 0 reject; rejected and infinity lanes give 96 zero bytes.
// mark_invalid != 0 (internal use: compressed inputs of the pairing entry points): a rejected lane becomes the off-curve record (0, 1)
// instead of zeros, so the kernel that consumes `out` reports it like any point that is not on the curve (0xff lane, C12381_E_POINT)
__global__ void __launch_bounds__(BLOCK, 2) g1_decompress_kernel(size_t n, const uint8_t* in, uint8_t* out, uint8_t* status, int mark_invalid) {
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
        if (mark_invalid && tag != 0) ry[11] = 0x01000000u;
    }
    store_raw48(out + 96 * i, rx); store_raw48(out + 96 * i + 48, ry);
    if (status) status[i] = tag == 0 ? 1 : (ok ? 1 : 0);
}

// One level of the per-window sums of the MSM, 64 points per wavefront: element index = group * W + w; wavefront (g, w) loads
// the 64 points (64 g + lane) * W + w, adds them in six shuffle steps and stores element g * W + w of the next level —
// 4096 partial sums per window become one in two levels of depth 6 instead of three levels of up to 32 dependent additions
__global__ void __launch_bounds__(BLOCK, 2) g1_wave_reduce_kernel(size_t groups, int W, const int32_t* in, size_t in_stride, int32_t* outp, size_t out_stride) {
    const size_t wave = ((size_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63u;
    const size_t out_groups = (groups + 63) / 64;
    if (wave >= out_groups * (size_t)W) return;                               // wave-uniform
    const size_t g = wave / (size_t)W, w = wave % (size_t)W;
    const size_t src = g * 64 + lane;
    g1p acc, t, nn;
    g1_set_inf(acc);
    if (src < groups) soa_load_g1(acc, in, in_stride, src * (size_t)W + w);
#pragma unroll 1
    for (int off = 32; off >= 1; off >>= 1) {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            t.x.l[i] = __shfl_down(acc.x.l[i], off, 64); t.y.l[i] = __shfl_down(acc.y.l[i], off, 64); t.z.l[i] = __shfl_down(acc.z.l[i], off, 64);
        }
        g1_add(acc, t);
        g1_norm1(nn, acc); acc = nn;
    }
    if (lane == 0) soa_store_g1(outp, out_stride, g * (size_t)W + w, acc);
}

// A doubling spread over FOUR lanes (the Horner chain is 112 dependent doublings in a single lane otherwise: ~1.1 ms of an
// MSM, a third of a small one).  The point is replicated in the lanes of a quad; the eight products of g1_dbl form two rounds
// of four independent ones — Y^2, YZ, Z^2, XY, then t1*z8, u*y3, t2*z8, u*xy — so lane r of the quad computes product r
// of each round and the quad exchanges the results by DPP broadcasts (quad_perm [k,k,k,k]); additions are replicated.
// Same values as g1_dbl (Y3 is the sum of two separately reduced products instead of one lazily reduced sum).
__device__ __forceinline__ void quad_bcast(fp& r, const fp& v, int k) {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        int32_t w;
        switch (k) {                                               // the control word must be a compile-time constant
            case 0: w = __builtin_amdgcn_mov_dpp(v.l[i], 0x00, 0xF, 0xF, true); break;
            case 1: w = __builtin_amdgcn_mov_dpp(v.l[i], 0x55, 0xF, 0xF, true); break;
            case 2: w = __builtin_amdgcn_mov_dpp(v.l[i], 0xAA, 0xF, 0xF, true); break;
            default: w = __builtin_amdgcn_mov_dpp(v.l[i], 0xFF, 0xF, 0xF, true); break;
        }
        r.l[i] = w;
    }
}
__device__ __forceinline__ void g1_dbl_quad(g1p& p, int r) {
    fp a, b, prod, t0, t1, t2, xy, z8, u, y3s;
    // round 1: r0 Y*Y, r1 Y*Z, r2 Z*Z, r3 X*Y
    fp_select(a, r == 2, p.z, p.y); fp_select(a, r == 3, p.x, a);
    fp_select(b, r == 1 || r == 2, p.z, p.y);
    fp_mul(prod, a, b);
    quad_bcast(t0, prod, 0); quad_bcast(t1, prod, 1); quad_bcast(t2, prod, 2); quad_bcast(xy, prod, 3);
    fp_mul_small(z8, t0, 8);                             // 8 Y^2
    fp_mul_small(t2, t2, 12);                            // 3b Z^2
    fp_add(y3s, t0, t2);
    fp_dbl(u, t2); fp_add(u, u, t2);                     // 9b Z^2
    fp_sub(u, t0, u);
    fp_norm1(u, u);
    // round 2: r0 t1*z8 (Z3), r1 u*y3s, r2 t2*z8, r3 u*xy
    fp_select(a, r == 0, t1, u); fp_select(a, r == 2, t2, a);
    fp_select(b, r == 1, y3s, z8); fp_select(b, r == 3, xy, b);
    fp_mul(prod, a, b);
    fp ya, yb, xh;
    quad_bcast(p.z, prod, 0); quad_bcast(ya, prod, 1); quad_bcast(yb, prod, 2); quad_bcast(xh, prod, 3);
    fp_add(p.y, ya, yb);
    fp_dbl(p.x, xh);
}
// p <- [|x|]p on a quad (the point replicated in its four lanes): the 63 doublings of the double-and-add chain run as g1_dbl_quad,
// the 5 additions are replicated
__device__ __forceinline__ void g1_mul_absx_quad(g1p& p, int r) {
    g1p base, acc, nn;
    g1_norm1(base, p);
    acc = base;
#pragma unroll 1
    for (int i = 62; i >= 0; --i) {
        g1_dbl_quad(acc, r);
        if ((BLS_X_W[i >> 5] >> (i & 31)) & 1u) { g1_norm1(nn, acc); g1_add(nn, base); acc = nn; }
    }
    p = acc;
}
// term = [r]phi(S) for S = bucket `sbucket`, the sum of the points whose scalar is below x^2 (msm.hpp) — g1_glv_small_scalar_term's
// sequence on ONE QUAD: a membership test of 126 doublings (phi(phi(S)) = [-x^2]phi(S) iff S is in G1) when any scalar was small,
// the 255-bit multiple only for S outside G1, an immediate return for an empty bucket.  A single lane took 1.57 ms for the test —
// longer than the window reductions it hides behind on the side stream; the quad takes a third of that.  One wavefront of 64 with the
// whole register file (like msm_horner_kernel): under the 256-register budget of the other kernels its seven live points spilled (101
// registers) and the test took 0.87 ms — it then ran on beside the first fold of the window sums and slowed that one from 80 to 360 us.
__device__ __forceinline__ void msm_small_term_quad(g1p& term, const g1p& S, int r) {
    g1_set_inf(term);
    if (!g1_is_inf(S)) {                                       // quad-uniform
        fp beta;
        fp_set_const(beta, FP_BETA_A);
        g1p q, s1, t;
        fp_mul(q.x, S.x, beta); q.y = S.y; q.z = S.z;          // phi(S)
        g1_norm1(q, q);
        s1 = q;
        g1_mul_absx_quad(s1, r); g1_mul_absx_quad(s1, r);      // [x^2]phi(S)
        g1_norm1(s1, s1);
        fp_mul(t.x, q.x, beta); t.y = q.y; t.z = q.z;          // phi(phi(S))
        g1_norm1(t, t);
        g1p u = s1;
        g1_add(u, t);
        if (!g1_is_inf(u)) {                                   // S outside G1: [r]Q = [x^4]Q - [x^2]Q + Q for Q = phi(S)
            g1p s2 = s1;
            g1_mul_absx_quad(s2, r); g1_mul_absx_quad(s2, r);
            fp_neg(t.y, s1.y); t.x = s1.x; t.z = s1.z;
            g1_norm1(t, t);
            g1_norm1(s2, s2);
            g1_add(s2, t);
            g1_norm1(s2, s2);
            g1_add(s2, q);
            term = s2;
        }
    }
}
// after the bucket kernel: the term for a small-scalar bucket too long for msm_small_early_kernel (which has otherwise done the work: *done != 0)
__global__ void __launch_bounds__(64, 1) msm_small_term_kernel(const int32_t* sbucket, int32_t* term_out, const uint32_t* done) {
    if (threadIdx.x >= 4 || blockIdx.x != 0 || *done != 0u) return;
    const int r = (int)(threadIdx.x & 3u);
    g1p S, term, nn;
    tab_load_g1(S, sbucket);
    msm_small_term_quad(term, S, r);
    g1_norm1(nn, term);
    if (r == 0) tab_store_g1(term_out, nn);
}
// The same BEFORE the bucket kernel, on the side stream, for a small-scalar bucket of at most `early_max` entries (the usual case: none, or
// the odd scalar 0 < k < x^2 of a batch): one wavefront adds the entries up (lane-strided, then six shuffle steps) and its first quad runs
// the membership test while the bucket kernel works — the 0.9 ms of the test are hidden instead of standing between the window reductions
// and the Horner chain (and slowing the first fold they ran beside from 80 to 300 us).  msm_sizes_kernel / msm_bucket_kernel treat that
// bucket as empty under the same condition.  *done = 1 when the term has been written.
__global__ void __launch_bounds__(64, 1) msm_small_early_kernel(const uint32_t* lo, const uint32_t* hi, uint32_t small_bucket, uint32_t early_max,
                                                             const uint32_t* vals, const int32_t* pts2, int32_t* term_out, uint32_t* done) {
    if (blockIdx.x != 0) return;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t l = lo[small_bucket], h = hi[small_bucket];
    if (h - l > early_max) { if (lane == 0) *done = 0u; return; }      // wave-uniform
    g1p acc, t, nn;
    g1_set_inf(acc);
    if (h > l) {
#pragma unroll 1
        for (uint32_t j = l + lane; j < h; j += 64u) {
            fp x, y;
            msm_load_pt(x, y, pts2 + (size_t)vals[j] * MSM_PT_STRIDE);
            g1_add_affine(acc, x, y);
        }
        g1_norm1(nn, acc); acc = nn;
#pragma unroll 1
        for (int off = 32; off >= 1; off >>= 1) {
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                t.x.l[i] = __shfl_down(acc.x.l[i], off, 64); t.y.l[i] = __shfl_down(acc.y.l[i], off, 64); t.z.l[i] = __shfl_down(acc.z.l[i], off, 64);
            }
            g1_add(acc, t);
            g1_norm1(nn, acc); acc = nn;
        }
#pragma unroll
        for (int i = 0; i < NL; ++i) {                                 // the sum sits in lane 0: replicate it over the first quad
            acc.x.l[i] = __shfl(acc.x.l[i], 0, 64); acc.y.l[i] = __shfl(acc.y.l[i], 0, 64); acc.z.l[i] = __shfl(acc.z.l[i], 0, 64);
        }
    }
    if (lane >= 4u) return;
    g1p term;
    msm_small_term_quad(term, acc, (int)lane);
    g1_norm1(nn, term);
    if (lane == 0) { tab_store_g1(term_out, nn); *done = 1u; }
}

// R = sum_w 2^(c w) R_w + term: the Horner chain on one quad (lanes 0..3 of a wavefront), doublings spread over its lanes
__global__ void __launch_bounds__(64, 1) msm_horner_kernel(const int32_t* rw, size_t stride, int W, int c, int32_t* out, size_t out_stride,
                                                        const int32_t* term_in) {
    if (threadIdx.x >= 4 || blockIdx.x != 0) return;
    const int r = (int)(threadIdx.x & 3u);
    g1p acc, q, nn, term;
    soa_load_g1(acc, rw, stride, (size_t)(W - 1));
#pragma unroll 1
    for (int w = W - 2; w >= 0; --w) {
#pragma unroll 1
        for (int b = 0; b < c; ++b) g1_dbl_quad(acc, r);
        soa_load_g1(q, rw, stride, (size_t)w);
        g1_norm1(nn, acc);
        g1_add(nn, q);
        acc = nn;
    }
    g1_norm1(nn, acc);
    tab_load_g1(term, term_in);
    g1_add(nn, term);
    g1_norm1(acc, nn);
    if (r == 0) soa_store_g1(out, out_stride, 0, acc);
}

}  // namespace c12381
