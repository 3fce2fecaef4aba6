// Scalar-field (Zp of the reference, p = the group order r) batch helpers: the signing-side arithmetic of
// include/crypto12381/zp_number.hpp (operator* :295-380 via multiply/split, inverse :420-425 -> BIG_invmodp,
// from_hash :540-548, sum/inner products :549-615) on canonical 32-byte big-endian values.
// Representation: 8 x 32-bit words, Montgomery radix 2^256, saturated CIOS — this path is a handful of
// multiplications per element, nowhere near the point arithmetic in cost, so it uses the simplest exact form.
#pragma once
#include "fp.hpp"

namespace c12381 {

struct fr { uint32_t w[8]; };      // little-endian words, value < r, Montgomery form unless stated

C12381_HD bool fr_geq_r(const uint32_t (&a)[8]) {
    for (int i = 7; i >= 0; --i) {
        if (a[i] > ORDER_R[i]) return true;
        if (a[i] < ORDER_R[i]) return false;
    }
    return true;
}
C12381_HD void fr_sub_r(uint32_t (&a)[8]) {
    int64_t b = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        b += (int64_t)a[i] - (int64_t)ORDER_R[i];
        a[i] = (uint32_t)b;
        b >>= 32;
    }
}
C12381_HD void fr_add(fr& r, const fr& a, const fr& b) {
    uint64_t c = 0;
    uint32_t t[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { c += (uint64_t)a.w[i] + b.w[i]; t[i] = (uint32_t)c; c >>= 32; }
    if (c || fr_geq_r(t)) fr_sub_r(t);         // a + b < 2r < 2^256: a carry out cannot occur, kept for clarity
#pragma unroll
    for (int i = 0; i < 8; ++i) r.w[i] = t[i];
}
C12381_HD bool fr_is_zero(const fr& a) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) o |= a.w[i];
    return o == 0;
}
C12381_HD void fr_neg(fr& r, const fr& a) {
    const bool z = fr_is_zero(a);
    int64_t b = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        b += (int64_t)ORDER_R[i] - (int64_t)a.w[i];
        r.w[i] = z ? 0u : (uint32_t)b;
        b >>= 32;
    }
}
C12381_HD void fr_sub(fr& r, const fr& a, const fr& b) {
    fr nb;
    fr_neg(nb, b);
    fr_add(r, a, nb);
}
// Montgomery product a * b / 2^256 mod r (CIOS, operands < r)
C12381_HD void fr_mul(fr& r, const fr& a, const fr& b) {
    uint32_t t[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) t[i] = 0;
#pragma unroll 1
    for (int i = 0; i < 8; ++i) {
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            c += (uint64_t)a.w[j] * b.w[i] + t[j];
            t[j] = (uint32_t)c;
            c >>= 32;
        }
        c += t[8];
        t[8] = (uint32_t)c;
        t[9] = (uint32_t)(c >> 32);
        const uint32_t m = t[0] * FR_N0;
        c = (uint64_t)m * ORDER_R[0] + t[0];
        c >>= 32;
#pragma unroll
        for (int j = 1; j < 8; ++j) {
            c += (uint64_t)m * ORDER_R[j] + t[j];
            t[j - 1] = (uint32_t)c;
            c >>= 32;
        }
        c += t[8];
        t[7] = (uint32_t)c;
        t[8] = t[9] + (uint32_t)(c >> 32);
    }
    uint32_t o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = t[i];
    if (t[8] || fr_geq_r(o)) fr_sub_r(o);
#pragma unroll
    for (int i = 0; i < 8; ++i) r.w[i] = o[i];
}
C12381_HD void fr_set_words(fr& r, const uint32_t (&c)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) r.w[i] = c[i];
}
// any 256-bit integer (little-endian words) -> Montgomery form of its residue mod r
C12381_HD void fr_from_words(fr& r, const uint32_t (&k)[8]) {
    uint32_t t[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = k[i];
    // 2^256 < 3r: at most two subtractions
    if (fr_geq_r(t)) fr_sub_r(t);
    if (fr_geq_r(t)) fr_sub_r(t);
    fr x, r2;
    fr_set_words(x, t);
    fr_set_words(r2, FR_R2);
    fr_mul(r, x, r2);
}
// Montgomery form -> canonical little-endian words
C12381_HD void fr_to_words(uint32_t (&k)[8], const fr& a) {
    fr one, t;
#pragma unroll
    for (int i = 0; i < 8; ++i) one.w[i] = i == 0 ? 1u : 0u;
    fr_mul(t, a, one);
#pragma unroll
    for (int i = 0; i < 8; ++i) k[i] = t.w[i];
}
// a^(r-2): the modular inverse, 0 -> 0 like BIG_invmodp (big_B384_58.cpp:1827; unit-tests/zp_number.cpp:76)
C12381_HDN void fr_inv(fr& r, const fr& a) {
    fr tab[16];
    fr_set_words(tab[0], FR_R1);
    tab[1] = a;
#pragma unroll 1
    for (int i = 2; i < 16; ++i) fr_mul(tab[i], tab[i - 1], a);
    fr acc;
    fr_set_words(acc, FR_R1);
#pragma unroll 1
    for (int wi = 63; wi >= 0; --wi) {
        fr_mul(acc, acc, acc); fr_mul(acc, acc, acc); fr_mul(acc, acc, acc); fr_mul(acc, acc, acc);
        const uint32_t d = (EXP_R_MINUS_2[wi / 8] >> (4 * (wi % 8))) & 15u;
        fr_mul(acc, acc, tab[d]);
    }
    r = acc;
}
// 64-byte big-endian digest (16 numeric words, w[0] most significant) mod r -> Montgomery form
// (Zp from_hash, zp_number.hpp:540-548: fixed_time_mod of the 512-bit integer)
C12381_HD void fr_from_digest_words(fr& r, const uint32_t (&w)[16]) {
    uint32_t hi[8], lo[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { hi[i] = w[7 - i]; lo[i] = w[15 - i]; }
    fr h, l, c;
    fr_from_words(h, hi);
    fr_from_words(l, lo);
    fr_set_words(c, FR_R2);            // Montgomery form of 2^256
    fr_mul(h, h, c);
    fr_add(r, h, l);
}

}  // namespace c12381
