// Kernels either side of the point arithmetic, one element per lane:
//   g1_from_hash_kernel   64-byte digest -> point of G1, or field element -> point of E (h2c.hpp); projective SoA
//                         for g1_finish_kernel
//   zp_op_kernel          scalar-field mul / add / sub / neg on canonical 32-byte values (fr.hpp)
//   zp_batch_inv_kernel   simultaneous inversion (one exponentiation per 16 elements), optionally of x[i] + gamma
//   zp_from_hash_kernel   64-byte digest -> scalar mod r
//   zp_fold_kernel        strided partial sums of a[i] * b[i] (or of a[i]) mod r: inner products and sums
#include "kernels_common.hpp"
#include "fr.hpp"
#include "h2c.hpp"

using namespace c12381;

namespace {

__device__ __forceinline__ void store_raw32(uint8_t* p, const uint32_t* w) {
    uint4* q = reinterpret_cast<uint4*>(p);
#pragma unroll
    for (int i = 0; i < 2; ++i) q[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}
__device__ __forceinline__ void fr_load32(fr& r, const uint8_t* p) {
    uint32_t raw[8], k[8];
    load_raw32(raw, p);
    scalar_from_raw32(k, raw);
    fr_from_words(r, k);
}
__device__ __forceinline__ void fr_store32(uint8_t* p, const fr& a) {
    uint32_t k[8], raw[8];
    fr_to_words(k, a);
#pragma unroll
    for (int i = 0; i < 8; ++i) raw[i] = bswap32(k[7 - i]);
    store_raw32(p, raw);
}

}  // namespace

namespace c12381 {

// mode 0: 64-byte digests -> from_hash (map + cofactor);  mode 1: 48-byte field elements -> map_to_point only;
// mode 2: 96-byte affine points -> multiply_cofactor only
__global__ void __launch_bounds__(BLOCK, 2) g1_from_hash_kernel(size_t n, const uint8_t* in, int mode, int32_t* proj, size_t proj_stride,
                                                             int* bad_flag) {
    const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    g1p p, o;
    if (mode == 0) {
        uint32_t raw[16];
        load_raw32(raw, in + 64 * i);
        load_raw32(raw + 8, in + 64 * i + 32);
        g1_from_digest(p, raw);
    } else if (mode == 1) {
        uint32_t raw[12];
        fp u;
        load_raw48(raw, in + 48 * i);
        fp_from_raw48(u, raw);
        g1_map_to_point(p, u);
    } else {
        bool inf, ok;
        g1_parse96(p.x, p.y, inf, ok, in + 96 * i);
        fp_one(p.z);
        if (inf) g1_set_inf(p);
        g1_clear_cofactor(p);
        if (!ok) { *bad_flag = 1; fp_one(p.x); fp_zero(p.y); fp_zero(p.z); }      // poison, as g1_mul_kernel does
    }
    g1_norm1(o, p);
    soa_store_g1(proj, proj_stride, i, o);
}

// op: 0 mul, 1 add, 2 sub, 3 neg, 4 inverse (0 -> 0)
__global__ void __launch_bounds__(BLOCK, 2) zp_op_kernel(int op, size_t n, const uint8_t* a, const uint8_t* b, uint8_t* out) {
    const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    fr x, y, r;
    fr_load32(x, a + 32 * i);
    if (b) fr_load32(y, b + 32 * i); else y = x;
    switch (op) {
        case 0: fr_mul(r, x, y); break;
        case 1: fr_add(r, x, y); break;
        case 2: fr_sub(r, x, y); break;
        case 3: fr_neg(r, x); break;
        default: fr_inv(r, x); break;              // the entry points send inversions to zp_batch_inv_kernel
    }
    fr_store32(out + 32 * i, r);
}

// out[i] = 1 / (x[i] + gamma) mod r  (gamma == nullptr: 1 / x[i]); inverse(0) = 0 (BIG_invmodp; unit-tests/zp_number.cpp:76).
// Simultaneous inversion: lane t owns the ZP_INV_RUN elements t, t + T, t + 2T, ... (coalesced), multiplies them up
// (zeros replaced by 1), inverts the product ONCE (a^(r-2), 319 multiplications) and unwinds — 3 multiplications per
// element plus 1/16 of an exponentiation instead of a whole one.  `pref`: 8 words per element, running products.
// out may alias x (an element is read before it is written, by the same lane).
__global__ void __launch_bounds__(BLOCK, 2) zp_batch_inv_kernel(size_t n, size_t T, const uint8_t* x, const uint8_t* gamma, uint8_t* out, uint32_t* pref) {
    const size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (t >= T) return;
    fr g, one, run;
    fr_set_words(one, FR_R1);
    if (gamma) fr_load32(g, gamma);
    run = one;
#pragma unroll 1
    for (int j = 0; j < ZP_INV_RUN; ++j) {
        const size_t i = t + (size_t)j * T;
        if (i >= n) break;
        fr a;
        fr_load32(a, x + 32 * i);
        if (gamma) fr_add(a, a, g);
        if (fr_is_zero(a)) a = one;
        fr_mul(run, run, a);
        uint4* p = reinterpret_cast<uint4*>(pref + 8 * i);
        p[0] = make_uint4(run.w[0], run.w[1], run.w[2], run.w[3]);
        p[1] = make_uint4(run.w[4], run.w[5], run.w[6], run.w[7]);
    }
    fr inv;
    fr_inv(inv, run);
#pragma unroll 1
    for (int j = ZP_INV_RUN - 1; j >= 0; --j) {
        const size_t i = t + (size_t)j * T;
        if (i >= n) continue;
        fr a, prev, r;
        fr_load32(a, x + 32 * i);
        if (gamma) fr_add(a, a, g);
        const bool z = fr_is_zero(a);
        if (z) a = one;
        if (j > 0) {
            const uint4* p = reinterpret_cast<const uint4*>(pref + 8 * (i - T));
            const uint4 lo = p[0], hi = p[1];
            prev.w[0] = lo.x; prev.w[1] = lo.y; prev.w[2] = lo.z; prev.w[3] = lo.w; prev.w[4] = hi.x; prev.w[5] = hi.y; prev.w[6] = hi.z; prev.w[7] = hi.w;
        } else prev = one;
        fr_mul(r, inv, prev);
        fr_mul(inv, inv, a);
        if (z) { for (int k = 0; k < 8; ++k) r.w[k] = 0; }
        fr_store32(out + 32 * i, r);
    }
}

__global__ void __launch_bounds__(BLOCK, 2) zp_from_hash_kernel(size_t n, const uint8_t* digests, uint8_t* out) {
    const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    uint32_t raw[16], w[16];
    load_raw32(raw, digests + 64 * i);
    load_raw32(raw + 8, digests + 64 * i + 32);
#pragma unroll
    for (int j = 0; j < 16; ++j) w[j] = bswap32(raw[j]);
    fr r;
    fr_from_digest_words(r, w);
    fr_store32(out + 32 * i, r);
}

// out[t] = sum over i = t (mod T) of a[i] * b[i] (b == nullptr: of a[i]), canonical bytes
__global__ void __launch_bounds__(BLOCK, 2) zp_fold_kernel(size_t n, const uint8_t* a, const uint8_t* b, size_t T, uint8_t* out) {
    const size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (t >= T) return;
    fr acc;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc.w[j] = 0;
#pragma unroll 1
    for (size_t i = t; i < n; i += T) {
        fr x;
        fr_load32(x, a + 32 * i);
        if (b) { fr y; fr_load32(y, b + 32 * i); fr_mul(x, x, y); }
        fr_add(acc, acc, x);
    }
    fr_store32(out + 32 * t, acc);
}

// several folds in one launch (blockIdx.y = column): column y of the first stage multiplies a[i] by 1 (y = 0), r[i] (y = 1)
// or m[(y - 2) n + i] — the nmsg + 2 inner products of the aggregate BBS+ verification; later stages sum column y of the
// previous stage's partial sums (a + y * a_col_stride).  out[T * y + t]
__global__ void __launch_bounds__(BLOCK, 2) zp_fold_cols_kernel(size_t n, const uint8_t* a, size_t a_col_stride, const uint8_t* r32, const uint8_t* m32,
                                                             int first, size_t T, uint8_t* out) {
    const size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (t >= T) return;
    const size_t y = blockIdx.y;
    const uint8_t* ap = a + a_col_stride * y;
    const uint8_t* bp = !first || y == 0 ? nullptr : (y == 1 ? r32 : m32 + 32 * n * (y - 2));
    fr acc;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc.w[j] = 0;
#pragma unroll 1
    for (size_t i = t; i < n; i += T) {
        fr x;
        fr_load32(x, ap + 32 * i);
        if (bp) { fr v; fr_load32(v, bp + 32 * i); fr_mul(x, x, v); }
        fr_add(acc, acc, x);
    }
    fr_store32(out + 32 * (T * y + t), acc);
}


// ------------------------------------------------------------------ BBS+ wire formats (the caller side of config 5)
// examples/bbs-plus/src/bbs+.cpp:57-73 starts from serialized values: parse<G1, G2, G1>(pp.g1_g2_h0) (49 + 97 + 49 bytes),
// parse<G1>(pp.h) (49 bytes each), parse<G2>(pk) (97 bytes), encode_to<Zp>(message), parse<G1, Zp, Zp>(signature) (49 + 48 + 48).
// bbs_wire_pub_kernel gathers the public points into the strides the decompression kernels read; bbs_wire_prep_kernel does the
// per-signature part: A -> 49-byte array, x and r -> 32-byte scalars with parse<Zp>'s range check (48 big-endian bytes below r,
// zp_number.hpp:226-236), message bytes -> encode_to<Zp> units (zp_number.hpp:1011-1037: 31-byte units behind a 0x01 byte, a short
// last unit left-aligned), message-major.  status[j] = 0: the reference would throw for signature j.
__global__ void __launch_bounds__(BLOCK, 2) bbs_wire_pub_kernel(size_t nblk, const uint8_t* g1_g2_h0, const uint8_t* h49, const uint8_t* pk97,
                                                             uint8_t* g1s49, uint8_t* g2s97) {
    const size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    const size_t n1 = (2 + nblk) * 49, n2 = 2 * 97;
    if (t < n1) {
        const size_t e = t / 49, b = t % 49;
        g1s49[t] = e == 0 ? g1_g2_h0[b] : (e == 1 ? g1_g2_h0[146 + b] : h49[49 * (e - 2) + b]);
    } else if (t < n1 + n2) {
        const size_t u = t - n1, e = u / 97, b = u % 97;
        g2s97[u] = e == 0 ? g1_g2_h0[49 + b] : pk97[b];
    }
}
__device__ __forceinline__ bool wire_zp(uint8_t* out32, const uint8_t* b48) {
    uint32_t hi = 0, w[8];
#pragma unroll
    for (int i = 0; i < 16; ++i) hi |= b48[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) {                     // w[0] = least significant word
        const uint8_t* q = b48 + 16 + 4 * (7 - i);
        w[i] = ((uint32_t)q[0] << 24) | ((uint32_t)q[1] << 16) | ((uint32_t)q[2] << 8) | (uint32_t)q[3];
    }
    uint64_t bw = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { const uint64_t t = (uint64_t)w[i] - ORDER_R[i] - bw; bw = (t >> 32) & 1; }
#pragma unroll
    for (int i = 0; i < 32; ++i) out32[i] = b48[16 + i];
    return hi == 0 && bw == 1;                         // value < r
}
__global__ void __launch_bounds__(BLOCK, 2) bbs_wire_prep_kernel(size_t n, size_t msg_len, size_t nblk, const uint8_t* sig145, const uint8_t* msgs,
                                                              uint8_t* a49, uint8_t* x32, uint8_t* r32, uint8_t* m32, uint8_t* status) {
    const size_t j = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (j >= n) return;
    const uint8_t* s = sig145 + 145 * j;
#pragma unroll 1
    for (int i = 0; i < 49; ++i) a49[49 * j + i] = s[i];
    const bool okx = wire_zp(x32 + 32 * j, s + 49);
    const bool okr = wire_zp(r32 + 32 * j, s + 97);
    status[j] = (okx && okr) ? 1 : 0;
    const uint8_t* msg = msgs + msg_len * j;
#pragma unroll 1
    for (size_t i = 0; i < nblk; ++i) {
        uint8_t* o = m32 + 32 * (i * n + j);
        const size_t len = (i + 1) * 31 <= msg_len ? 31 : msg_len - i * 31;
        o[0] = 1;
#pragma unroll 1
        for (size_t b = 0; b < 31; ++b) o[1 + b] = b < len ? msg[31 * i + b] : 0;
    }
}
// ok[j] <- 0xff where the reference would have thrown: malformed x / r, A not decodable, or public material not decodable
__global__ void __launch_bounds__(BLOCK, 2) bbs_wire_finish_kernel(size_t n, size_t npub1, const uint8_t* st_sig, const uint8_t* st_a, const uint8_t* st_pub1,
                                                                const uint8_t* st_pub2, uint8_t* ok, int* bad_flag) {
    const size_t j = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (j >= n) return;
    bool pub = st_pub2[0] != 0 && st_pub2[1] != 0;
#pragma unroll 1
    for (size_t i = 0; i < npub1; ++i) pub = pub && st_pub1[i] != 0;
    if (!pub) { ok[j] = 0xff; if (j == 0) *bad_flag = 1; }
    else if (!st_sig[j] || !st_a[j]) ok[j] = 0xff;
}

}  // namespace c12381
