// Drop-in definitions of crypto12381's G1 / G2 / GT boundary functions on top of libc12381_hip.
//
// This is the reference-side binding of INTEGRATION.md as real code: it defines, with the reference's own
// C++ signatures (namespace crypto12381::detail::miracl_core, declared in the reference's
// include/crypto12381/miracl_core_interface.hpp:90-204), the functions that
// src/miracl_core_interface.cpp:109-289 forwards to MIRACL — and forwards them to the C ABI of
// include/c12381_hip.h instead.  It is compiled INSIDE the reference tree (it includes the reference's
// header from there; nothing of the reference is vendored here).  The hash / big / random functions of the
// seam (miracl_core_interface.hpp:16-64) are host-side scalar glue and stay with the reference's own file.
//
// The seam's PODs are opaque to the headers, so they are used as byte containers:
//   point1 (192 B): bytes 0..95  = x||y big-endian canonical, all zero = infinity
//   point2 (384 B): bytes 0..191 = x.b||x.a||y.b||y.a, all zero = infinity
//   fp12   (776 B): bytes 0..575 = FP12_toOctet encoding
// Every call is a size-1 batch on the GPU: functionally exact, latency-bound — callers that hold many
// operands should use the *_batch entry points directly.
#include <crypto12381/miracl_core_interface.hpp>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "c12381_hip.h"

namespace {

using crypto12381::detail::chunk_t;
namespace mc = crypto12381::detail::miracl_core;

// one context per host thread, created on first use and destroyed when the thread exits
struct thread_ctx {
    c12381_ctx* p = nullptr;
    thread_ctx() {
        if (c12381_create(0, &p) != 0) {                    // no CPU fallback
            std::fprintf(stderr, "crypto12381 HIP backend: no usable HIP device (c12381_create failed)\n");
            std::abort();
        }
    }
    ~thread_ctx() { if (p) c12381_destroy(p); }
    thread_ctx(const thread_ctx&) = delete;
    thread_ctx& operator=(const thread_ctx&) = delete;
};
c12381_ctx* ctx() {
    thread_local thread_ctx t;
    return t.p;
}
inline uint8_t* raw(mc::point1& p) { return reinterpret_cast<uint8_t*>(&p); }
inline uint8_t* raw(mc::point2& p) { return reinterpret_cast<uint8_t*>(&p); }
inline uint8_t* raw(mc::fp12& p) { return reinterpret_cast<uint8_t*>(&p); }
inline const uint8_t* raw(const mc::point1& p) { return reinterpret_cast<const uint8_t*>(&p); }
inline const uint8_t* raw(const mc::point2& p) { return reinterpret_cast<const uint8_t*>(&p); }
inline bool all_zero(const uint8_t* p, size_t n) { uint8_t t = 0; for (size_t i = 0; i < n; ++i) t |= p[i]; return t == 0; }

// p and r as big-endian bytes (public constants)
const uint8_t P_BE[48] = {0x1a,0x01,0x11,0xea,0x39,0x7f,0xe6,0x9a,0x4b,0x1b,0xa7,0xb6,0x43,0x4b,0xac,0xd7,0x64,0x77,0x4b,0x84,0xf3,0x85,0x12,0xbf,
                          0x67,0x30,0xd2,0xa0,0xf6,0xb0,0xf6,0x24,0x1e,0xab,0xff,0xfe,0xb1,0x53,0xff,0xff,0xb9,0xfe,0xff,0xff,0xff,0xff,0xaa,0xab};
// y -> p - y on 48 big-endian bytes (0 stays 0)
void fp_negate_be(uint8_t* y) {
    if (all_zero(y, 48)) return;
    int borrow = 0;
    for (int i = 47; i >= 0; --i) { int d = (int)P_BE[i] - (int)y[i] - borrow; borrow = d < 0; y[i] = (uint8_t)(d + (borrow ? 256 : 0)); }
}
// `big` is transparent: 7 little-endian limbs of 58 bits, normalised by the headers before every boundary call
// (zp_number.hpp:656-659).  Value < 2^406; reduced mod 2^256 * ... -> we pass the low 32 bytes when the value
// fits and otherwise reduce mod r on the host (the boundary's multiply reduces mod r anyway).
const uint64_t R_LE[4] = {0xffffffff00000001ull, 0x53bda402fffe5bfeull, 0x3339d80809a1d805ull, 0x73eda753299d7d48ull};
void scalar32(uint8_t out[32], const chunk_t (&k)[7]) {
    uint64_t w[8] = {0};                               // 512-bit little-endian; normalised limbs occupy disjoint bits
    for (int i = 0; i < 7; ++i) {
        const int wi = (58 * i) / 64, bo = (58 * i) % 64;
        const unsigned __int128 v = (unsigned __int128)((uint64_t)k[i] & 0x3ffffffffffffffull) << bo;
        w[wi] |= (uint64_t)v;
        w[wi + 1] |= (uint64_t)(v >> 64);
    }
    if (w[4] | w[5] | w[6] | w[7]) {                   // >= 2^256: binary long division by r
        uint64_t rem[5] = {0, 0, 0, 0, 0};
        for (int bit = 511; bit >= 0; --bit) {
            for (int i = 4; i >= 1; --i) rem[i] = (rem[i] << 1) | (rem[i - 1] >> 63);
            rem[0] = (rem[0] << 1) | ((w[bit / 64] >> (bit % 64)) & 1);
            uint64_t d[5]; unsigned __int128 bw = 0;
            for (int i = 0; i < 5; ++i) { unsigned __int128 t = (unsigned __int128)rem[i] - (i < 4 ? R_LE[i] : 0) - (uint64_t)bw; d[i] = (uint64_t)t; bw = (t >> 64) & 1; }
            if (!bw) std::memcpy(rem, d, sizeof rem);
        }
        std::memcpy(w, rem, 32);
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) out[31 - (8 * i + j)] = (uint8_t)(w[i] >> (8 * j));
}
// normalised big (value < 2^384) -> 48 big-endian bytes
void big48(uint8_t out[48], const chunk_t (&k)[7]) {
    uint64_t w[8] = {0};
    for (int i = 0; i < 7; ++i) {
        const int wi = (58 * i) / 64, bo = (58 * i) % 64;
        const unsigned __int128 v = (unsigned __int128)((uint64_t)k[i] & 0x3ffffffffffffffull) << bo;
        w[wi] |= (uint64_t)v;
        w[wi + 1] |= (uint64_t)(v >> 64);
    }
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 8; ++j) out[47 - (8 * i + j)] = (uint8_t)(w[i] >> (8 * j));
}
// The seam is noexcept and total: it has no way to report a failed call, and a caller that carried on with an unchanged
// or poisoned POD would sign / verify garbage.  Every failure of the backend (HIP error, out of memory, a point that
// is not on the curve — impossible for PODs that came through from_bytes —, an internal time-out) therefore ends the
// process with a message, exactly like a missing device does in ctx().
inline void ck(int rc) {
    if (rc == 0) return;
    std::fprintf(stderr, "crypto12381 HIP backend: call failed with code %d (%s)\n", rc, c12381_last_error(ctx()));
    std::abort();
}

const char* G1_GEN_HEX =
    "17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb"
    "08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1";
const char* G2_GEN_HEX =
    "13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e"
    "024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8"
    "0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be"
    "0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801";
void from_hex(uint8_t* out, const char* h, size_t n) {
    auto v = [](char c) { return c <= '9' ? c - '0' : c - 'a' + 10; };
    for (size_t i = 0; i < n; ++i) out[i] = (uint8_t)(v(h[2 * i]) * 16 + v(h[2 * i + 1]));
}

}  // namespace

namespace crypto12381::detail::miracl_core {

// ------------------------------------------------------------------ G1  (src/miracl_core_interface.cpp:109-182)
int from_bytes(point1& result, bytes_view& bytes) noexcept {                      // was ECP_fromOctet
    std::memset(&result, 0, sizeof result);
    const uint8_t* b = reinterpret_cast<const uint8_t*>(bytes.data);
    if (b[0] == 0x04) {
        uint8_t inf[96] = {0}, out[96];
        if (c12381_g1_add_batch(ctx(), 1, b + 1, inf, out, 96) != 0) return 0;    // on-curve check on the device
        std::memcpy(raw(result), b + 1, 96);
        return 1;
    }
    if (b[0] != 0x02 && b[0] != 0x03) return 0;
    uint8_t st = 0;
    if (c12381_g1_decompress_batch(ctx(), 1, b, raw(result), &st) != 0 || !st) { std::memset(&result, 0, sizeof result); return 0; }
    return 1;
}
void to_bytes(bytes_view& result, point1& point, bool compressed) noexcept {      // was ECP_toOctet
    const uint8_t* r = raw(point);
    uint8_t* o = reinterpret_cast<uint8_t*>(result.data);
    if (compressed) { o[0] = (uint8_t)(0x02 | (r[95] & 1)); std::memcpy(o + 1, r, 48); result.len = 49; }
    else { o[0] = 0x04; std::memcpy(o + 1, r, 96); result.len = 97; }
}
bool is_infinity(const point1& point) noexcept { return all_zero(raw(point), 96); }
void negate(point1& point) noexcept { if (!is_infinity(point)) fp_negate_be(raw(point) + 48); }
void add(point1& object, point1& point) noexcept { ck(c12381_g1_add_batch(ctx(), 1, raw(object), raw(point), raw(object), 96)); }
void sub(point1& object, point1& point) noexcept {
    point1 t = point;
    negate(t);
    add(object, t);
}
int equal(point1& l, point1& r) noexcept { return std::memcmp(raw(l), raw(r), 96) == 0 ? 1 : 0; }
void get_infinity(point1& result) noexcept { std::memset(&result, 0, sizeof result); }
int get_default_generator(point1& result) noexcept {
    std::memset(&result, 0, sizeof result);
    from_hex(raw(result), G1_GEN_HEX, 96);
    return 1;
}
// hash-to-G1 pieces (G1Point::from_hash, g1_point.hpp:219-234).  `fp` is opaque to the headers: bytes 0..47 = the
// canonical value, big-endian.
void residue(fp& result, const big& value) noexcept {                             // was FP_nres
    std::memset(&result, 0, sizeof result);
    big48(reinterpret_cast<uint8_t*>(&result), value);
}
void map_to_point(point1& result, const fp& value) noexcept {                     // was ECP_map2point
    std::memset(&result, 0, sizeof result);
    ck(c12381_g1_map_to_point_batch(ctx(), 1, reinterpret_cast<const uint8_t*>(&value), raw(result)));
}
void multiply_cofactor(point1& object) noexcept {                                 // was ECP_cfp: [1 - x]P
    ck(c12381_g1_clear_cofactor_batch(ctx(), 1, raw(object), raw(object)));
}
void multiply(point1& object, const big& value) noexcept {                        // was PAIR_G1mul
    uint8_t k[32];
    scalar32(k, value);
    ck(c12381_g1_mul_batch(ctx(), 1, raw(object), k, raw(object), 96));
}
void double_multiply(point1& p1, point1& p2, big& v1, big& v2) noexcept {         // was ECP_mul2: p1 = v1 p1 + v2 p2
    uint8_t pts[192], ks[64];
    std::memcpy(pts, raw(p1), 96); std::memcpy(pts + 96, raw(p2), 96);
    scalar32(ks, v1); scalar32(ks + 32, v2);
    ck(c12381_g1_sum_of_products(ctx(), 2, pts, ks, raw(p1), 96));      // ECP_mul2: true multiples, no endomorphism
}
void sum_of_products(point1& result, int n, point1* points, const big* numbers) noexcept {    // was ECP_muln
    std::vector<uint8_t> p((size_t)96 * n), k((size_t)32 * n);
    for (int i = 0; i < n; ++i) { std::memcpy(&p[(size_t)96 * i], raw(points[i]), 96); scalar32(&k[(size_t)32 * i], numbers[i]); }
    std::memset(&result, 0, sizeof result);
    ck(c12381_g1_sum_of_products(ctx(), (size_t)n, p.data(), k.data(), raw(result), 96));   // ECP_muln: true multiples
}

// ------------------------------------------------------------------ G2  (src/miracl_core_interface.cpp:187-236)
int from_bytes(point2& result, bytes_view& bytes) noexcept {                      // was ECP2_fromOctet
    std::memset(&result, 0, sizeof result);
    const uint8_t* b = reinterpret_cast<const uint8_t*>(bytes.data);
    if (b[0] == 0x04) {
        uint8_t inf[192] = {0}, out[192];
        if (c12381_g2_add_batch(ctx(), 1, b + 1, inf, out, 192) != 0) return 0;
        std::memcpy(raw(result), b + 1, 192);
        return 1;
    }
    uint8_t st = 0;
    if (c12381_g2_decompress_batch(ctx(), 1, b, raw(result), &st) != 0 || !st) { std::memset(&result, 0, sizeof result); return 0; }
    return 1;
}
void to_bytes(bytes_view& result, point2& point, bool compressed) noexcept {      // was ECP2_toOctet
    const uint8_t* r = raw(point);
    uint8_t* o = reinterpret_cast<uint8_t*>(result.data);
    if (compressed) {
        // FP2_sign (fp2_BLS12381.cpp:168-181): parity of y.a, or of y.b when y.a == 0; layout y.b || y.a
        const int sign = all_zero(r + 144, 48) ? (r[143] & 1) : (r[191] & 1);
        o[0] = (uint8_t)(0x02 | sign); std::memcpy(o + 1, r, 96); result.len = 97;
    } else { o[0] = 0x04; std::memcpy(o + 1, r, 192); result.len = 193; }
}
bool is_infinity(const point2& point) noexcept { return all_zero(raw(point), 192); }
void multiply(point2& object, const big& value) noexcept {                        // was PAIR_G2mul
    uint8_t k[32];
    scalar32(k, value);
    ck(c12381_g2_mul_batch(ctx(), 1, raw(object), k, raw(object), 192));
}
void negate(point2& point) noexcept { if (!is_infinity(point)) { fp_negate_be(raw(point) + 96); fp_negate_be(raw(point) + 144); } }
void add(point2& object, point2& point) noexcept { ck(c12381_g2_add_batch(ctx(), 1, raw(object), raw(point), raw(object), 192)); }
void sub(point2& object, point2& point) noexcept {
    point2 t = point;
    negate(t);
    add(object, t);
}
int equal(point2& l, point2& r) noexcept { return std::memcmp(raw(l), raw(r), 192) == 0 ? 1 : 0; }
void get_infinity(point2& result) noexcept { std::memset(&result, 0, sizeof result); }
int get_default_generator(point2& result) noexcept {
    std::memset(&result, 0, sizeof result);
    from_hex(raw(result), G2_GEN_HEX, 192);
    return 1;
}

// ------------------------------------------------------------------ GT  (src/miracl_core_interface.cpp:241-289)
void from_bytes(fp12& result, bytes_view& bytes) noexcept { std::memset(&result, 0, sizeof result); std::memcpy(raw(result), bytes.data, 576); }
void to_bytes(bytes_view& result, fp12& value) noexcept { std::memcpy(result.data, raw(value), 576); result.len = 576; }
void conjugate(fp12& result, fp12& value) noexcept { ck(c12381_gt_op_batch(ctx(), 1, 1, raw(value), nullptr, raw(result))); }
void multiply(fp12& result, fp12& value) noexcept { ck(c12381_gt_op_batch(ctx(), 0, 1, raw(result), raw(value), raw(result))); }
// was FP12_pow (fp12_BLS12381.cpp:736-774), which uses the exponent AS GIVEN.  The C ABI carries 32-byte exponents: a value below 2^256 is passed as it
// is; a value of 2^256 or more (a `big` holds up to 406 bits) is reduced mod r by scalar32 first.  Unobservable through the reference's headers: every
// exponent they form is a Zp_number, normalised below r < 2^255 (zp_number.hpp:656-659, :262-267), and for a base in GT (order r) the reduced
// exponent gives the same element anyway; it differs from FP12_pow only for an exponent >= 2^256 applied to a base OUTSIDE the order-r subgroup,
// which no caller of this seam produces.
void pow(fp12& result, fp12& base, const big& exponent) noexcept {
    uint8_t k[32];
    scalar32(k, exponent);
    ck(c12381_gt_op_batch(ctx(), 2, 1, raw(base), k, raw(result)));
}
int equal(fp12& l, fp12& r) noexcept { return std::memcmp(raw(l), raw(r), 576) == 0 ? 1 : 0; }
bool is_unity(fp12& value) noexcept {
    uint8_t ok = 0;
    ck(c12381_gt_is_unity_batch(ctx(), 1, raw(value), &ok));
    return ok == 1;
}
void pair_ate(fp12& result, point2& p2, point1& p1) noexcept {                    // was PAIR_ate
    std::memset(&result, 0, sizeof result);
    ck(c12381_miller_batch(ctx(), 1, raw(p1), raw(p2), raw(result)));
}
void pair_final_exponentiation(fp12& object) noexcept { ck(c12381_fexp_batch(ctx(), 1, raw(object), raw(object))); }   // was PAIR_fexp
void pair_double_ate(fp12& result, point2& p2, point1& p1, point2& q2, point1& q1) noexcept {   // was PAIR_double_ate
    // one joint Miller loop with shared squarings (c12381_pair_product_batch, k = 2), stopped before the final exponentiation
    uint8_t g1[192], g2[384];
    std::memcpy(g1, raw(p1), 96); std::memcpy(g1 + 96, raw(q1), 96);
    std::memcpy(g2, raw(p2), 192); std::memcpy(g2 + 192, raw(q2), 192);
    std::memset(&result, 0, sizeof result);
    ck(c12381_pair_product_batch(ctx(), 1, 2, g1, g2, raw(result), C12381_F_MILLER_ONLY));
}

}  // namespace crypto12381::detail::miracl_core
