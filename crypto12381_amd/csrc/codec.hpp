// Canonical byte encodings <-> device number format (BIG_toBytes/BIG_fromBytes
// big_B384_58.cpp:171-197, ECP_toOctet ecp_BLS12381.cpp:445-488 of the reference).
// All functions work on 32-bit words already fetched from memory ("raw" = the 4 bytes as they
// lie in the big-endian encoding, so a byte swap gives the numeric word).
#pragma once
#include "fp.hpp"

namespace c12381 {

C12381_HD uint32_t bswap32(uint32_t v) { return __builtin_bswap32(v); }

// 48 big-endian bytes given as 12 raw words -> Montgomery-form fp
C12381_HD void fp_from_raw48(fp& r, const uint32_t* raw) {
    uint32_t w[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) w[i] = bswap32(raw[i]);
    fp_from_words_be(r, w);
}
// Montgomery-form fp -> 12 raw words of the canonical 48-byte big-endian encoding
C12381_HD void fp_to_raw48(uint32_t* raw, const fp& a) {
    uint32_t w[12];
    fp_to_words_be(w, a);
#pragma unroll
    for (int i = 0; i < 12; ++i) raw[i] = bswap32(w[i]);
}
// 32 big-endian bytes (8 raw words) -> 8 little-endian numeric words
C12381_HD void scalar_from_raw32(uint32_t (&k)[8], const uint32_t* raw) {
#pragma unroll
    for (int i = 0; i < 8; ++i) k[i] = bswap32(raw[7 - i]);
}
C12381_HD bool raw_all_zero(const uint32_t* raw, int nwords) {
    uint32_t o = 0;
    for (int i = 0; i < nwords; ++i) o |= raw[i];
    return o == 0;
}

}  // namespace c12381
