// TEST INFRASTRUCTURE ONLY — never linked, imported or called by the product path.
//
// extern "C" batch driver over the REAL reference boundary
// (crypto12381::detail::miracl_core, /root/reference/include/crypto12381/
// miracl_core_interface.hpp:16-204, defined in /root/reference/src/
// miracl_core_interface.cpp:12-289).  This file is our own code; it is compiled
// together with the reference's sources *where they lie* by oracle/Makefile into
// oracle/_ref/libc12381_ref.so (git-ignored, never committed).  Nothing of the
// reference is copied into this repository.
//
// All data crosses this wrapper as canonical big-endian bytes:
//   Fp   48 B            Zp scalar 32 B (any value < 2^256; the reference reduces mod r)
//   G1   96 B  x‖y       (96 zero bytes = point at infinity)   | 49 B compressed
//   G2  192 B  x.b‖x.a‖y.b‖y.a (192 zero bytes = infinity)     | 97 B compressed
//   GT  576 B  MIRACL tower order c‖b‖a (fp12_BLS12381.cpp:923-929)
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>
#include <algorithm>

#include <crypto12381/miracl_core_interface.hpp>
#include <crypto12381/random.hpp>
#include <miracl-core/bls_BLS12381.h>

namespace mc = crypto12381::detail::miracl_core;
using namespace core;
using namespace BLS12381;
using namespace B384_58;

namespace {

bool all_zero(const uint8_t* p, size_t n) {
    for (size_t i = 0; i < n; ++i) if (p[i]) return false;
    return true;
}

void scalar_from32(mc::big& k, const uint8_t* s32) {
    char buf[48];
    std::memset(buf, 0, 16);
    std::memcpy(buf + 16, s32, 32);
    mc::from_bytes(k, buf);
}

// 96-byte affine -> point1 through the boundary decoder (tag 0x04).
int g1_load(mc::point1& P, const uint8_t* a96) {
    if (all_zero(a96, 96)) { mc::get_infinity(P); return 1; }
    char buf[97];
    buf[0] = 0x04;
    std::memcpy(buf + 1, a96, 96);
    mc::bytes_view v{97, 97, buf};
    return mc::from_bytes(P, v);
}
void g1_store(uint8_t* out, mc::point1& P, int fmt) {
    if (mc::is_infinity(P)) { std::memset(out, 0, fmt); return; }
    char buf[97];
    mc::bytes_view v{0, 97, buf};
    if (fmt == 49) { mc::to_bytes(v, P, true); std::memcpy(out, buf, 49); }
    else           { mc::to_bytes(v, P, false); std::memcpy(out, buf + 1, 96); }
}
int g2_load(mc::point2& P, const uint8_t* a192) {
    if (all_zero(a192, 192)) { mc::get_infinity(P); return 1; }
    char buf[193];
    buf[0] = 0x04;
    std::memcpy(buf + 1, a192, 192);
    mc::bytes_view v{193, 193, buf};
    return mc::from_bytes(P, v);
}
void g2_store(uint8_t* out, mc::point2& P, int fmt) {
    if (mc::is_infinity(P)) { std::memset(out, 0, fmt); return; }
    char buf[193];
    mc::bytes_view v{0, 193, buf};
    if (fmt == 97) { mc::to_bytes(v, P, true); std::memcpy(out, buf, 97); }
    else           { mc::to_bytes(v, P, false); std::memcpy(out, buf + 1, 192); }
}

template <class F>
void par_for(size_t n, int nthreads, F&& body) {
    if (nthreads <= 1 || n < 2) { body(0, n); return; }
    size_t T = std::min<size_t>(nthreads, n);
    std::vector<std::thread> th;
    for (size_t t = 0; t < T; ++t) {
        size_t lo = n * t / T, hi = n * (t + 1) / T;
        th.emplace_back([=, &body] { body(lo, hi); });
    }
    for (auto& x : th) x.join();
}

} // namespace

extern "C" {

int ref_g1_generator(uint8_t out[96]) {
    mc::point1 G;
    if (!mc::get_default_generator(G)) return -1;
    g1_store(out, G, 96);
    return 0;
}
int ref_g2_generator(uint8_t out[192]) {
    mc::point2 G;
    if (!mc::get_default_generator(G)) return -1;
    g2_store(out, G, 192);
    return 0;
}

// op: 0 mul, 1 add, 2 sub, 3 sqr(a), 4 neg(a), 5 inv(a), 6 sqrt(a) (status in ok[i]: 1 = QR)
int ref_fp_op_batch(int op, size_t n, const uint8_t* a, const uint8_t* b, uint8_t* out, uint8_t* ok) {
    for (size_t i = 0; i < n; ++i) {
        BIG x, y; FP fa, fb, fr;
        BIG_fromBytes(x, (char*)a + 48 * i);
        FP_nres(&fa, x);
        if (b) { BIG_fromBytes(y, (char*)b + 48 * i); FP_nres(&fb, y); }
        int st = 1;
        switch (op) {
            case 0: FP_mul(&fr, &fa, &fb); break;
            case 1: FP_add(&fr, &fa, &fb); break;
            case 2: FP_sub(&fr, &fa, &fb); break;
            case 3: FP_sqr(&fr, &fa); break;
            case 4: FP_neg(&fr, &fa); break;
            case 5: FP_inv(&fr, &fa, NULL); break;
            case 6: st = FP_qr(&fa, NULL); if (st) FP_sqrt(&fr, &fa, NULL); else FP_zero(&fr); break;
            default: return -1;
        }
        FP_reduce(&fr);
        BIG_zero(x);
        FP_redc(x, &fr);
        BIG_toBytes((char*)out + 48 * i, x);
        if (ok) ok[i] = (uint8_t)st;
    }
    return 0;
}

// multiply(point1&, const big&) — src/miracl_core_interface.cpp:174-177 -> PAIR_G1mul
int ref_g1_mul_batch(size_t n, const uint8_t* pts96, const uint8_t* scalars32, uint8_t* out, int out_fmt, int nthreads) {
    if (out_fmt != 49 && out_fmt != 96) return -1;
    int bad = 0;
    par_for(n, nthreads, [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i) {
            mc::point1 P; mc::big k;
            if (!g1_load(P, pts96 + 96 * i)) { bad = 1; std::memset(out + (size_t)out_fmt * i, 0xff, out_fmt); continue; }
            scalar_from32(k, scalars32 + 32 * i);
            mc::multiply(P, k);
            g1_store(out + (size_t)out_fmt * i, P, out_fmt);
        }
    });
    return bad ? -2 : 0;
}

// add(point1&, point1&) — src/miracl_core_interface.cpp:129-132
int ref_g1_add_batch(size_t n, const uint8_t* a96, const uint8_t* b96, uint8_t* out, int out_fmt) {
    for (size_t i = 0; i < n; ++i) {
        mc::point1 A, B;
        if (!g1_load(A, a96 + 96 * i) || !g1_load(B, b96 + 96 * i)) return -2;
        mc::add(A, B);
        g1_store(out + (size_t)out_fmt * i, A, out_fmt);
    }
    return 0;
}

// negate(point1&) / sub(point1&, point1&) / equal(point1&, point1&) — src/miracl_core_interface.cpp:124-127, 139-147
// (ECP_neg, ECP_sub, ECP_equals).  equal: out[i] = the seam's return value (1 / 0).
int ref_g1_neg_batch(size_t n, const uint8_t* a96, uint8_t* out, int out_fmt) {
    for (size_t i = 0; i < n; ++i) {
        mc::point1 A;
        if (!g1_load(A, a96 + 96 * i)) return -2;
        mc::negate(A);
        g1_store(out + (size_t)out_fmt * i, A, out_fmt);
    }
    return 0;
}
int ref_g1_sub_batch(size_t n, const uint8_t* a96, const uint8_t* b96, uint8_t* out, int out_fmt) {
    for (size_t i = 0; i < n; ++i) {
        mc::point1 A, B;
        if (!g1_load(A, a96 + 96 * i) || !g1_load(B, b96 + 96 * i)) return -2;
        mc::sub(A, B);
        g1_store(out + (size_t)out_fmt * i, A, out_fmt);
    }
    return 0;
}
int ref_g1_equal_batch(size_t n, const uint8_t* a96, const uint8_t* b96, uint8_t* out) {
    for (size_t i = 0; i < n; ++i) {
        mc::point1 A, B;
        if (!g1_load(A, a96 + 96 * i) || !g1_load(B, b96 + 96 * i)) return -2;
        out[i] = (uint8_t)mc::equal(A, B);
    }
    return 0;
}
// the same on values that are NOT fresh from the decoder: A = a + b computed through the seam (projective inside the
// reference), compared with the decoded c — ECP_equals cross-multiplies, the shim compares canonical bytes
int ref_g1_equal_sum_batch(size_t n, const uint8_t* a96, const uint8_t* b96, const uint8_t* c96, uint8_t* out) {
    for (size_t i = 0; i < n; ++i) {
        mc::point1 A, B, C;
        if (!g1_load(A, a96 + 96 * i) || !g1_load(B, b96 + 96 * i) || !g1_load(C, c96 + 96 * i)) return -2;
        mc::add(A, B);
        out[i] = (uint8_t)mc::equal(A, C);
    }
    return 0;
}

// from_bytes(point1&, bytes_view&) on 49-byte compressed input; leading 0x00 => infinity
// as in include/crypto12381/g1_point.hpp:89-93.  status[i] = 1 ok / 0 reject.
int ref_g1_decompress_batch(size_t n, const uint8_t* in49, uint8_t* out96, uint8_t* status) {
    for (size_t i = 0; i < n; ++i) {
        const uint8_t* s = in49 + 49 * i;
        mc::point1 P;
        if (s[0] == 0) { std::memset(out96 + 96 * i, 0, 96); status[i] = 1; continue; }
        char buf[49]; std::memcpy(buf, s, 49);
        mc::bytes_view v{49, 49, buf};
        int ok = mc::from_bytes(P, v);
        status[i] = (uint8_t)ok;
        if (ok) g1_store(out96 + 96 * i, P, 96); else std::memset(out96 + 96 * i, 0, 96);
    }
    return 0;
}
int ref_g1_compress_batch(size_t n, const uint8_t* in96, uint8_t* out49) {
    for (size_t i = 0; i < n; ++i) {
        mc::point1 P;
        if (!g1_load(P, in96 + 96 * i)) return -2;
        g1_store(out49 + 49 * i, P, 49);
    }
    return 0;
}

// double_multiply(p1,p2,v1,v2) — src/miracl_core_interface.cpp:179-182 -> ECP_mul2
int ref_g1_mul2_batch(size_t n, const uint8_t* p96, const uint8_t* q96, const uint8_t* u32, const uint8_t* v32, uint8_t* out, int out_fmt) {
    for (size_t i = 0; i < n; ++i) {
        mc::point1 P, Q; mc::big u, v;
        if (!g1_load(P, p96 + 96 * i) || !g1_load(Q, q96 + 96 * i)) return -2;
        scalar_from32(u, u32 + 32 * i); scalar_from32(v, v32 + 32 * i);
        mc::double_multiply(P, Q, u, v);
        g1_store(out + (size_t)out_fmt * i, P, out_fmt);
    }
    return 0;
}

// Π g_i^{x_i}: the header-level product (include/crypto12381/g1_point.hpp:371-404) is
// pairwise double_multiply + add; only the final point is canonical, so the wrapper
// evaluates it as multiply + add per term (same group element), sharded over threads.
int ref_g1_msm(size_t n, const uint8_t* pts96, const uint8_t* scalars32, uint8_t* out, int out_fmt, int nthreads) {
    size_t T = std::max(1, nthreads);
    std::vector<mc::point1> part(T);
    for (auto& p : part) mc::get_infinity(p);
    int bad = 0;
    std::vector<std::thread> th;
    for (size_t t = 0; t < T; ++t) {
        size_t lo = n * t / T, hi = n * (t + 1) / T;
        th.emplace_back([&, t, lo, hi] {
            for (size_t i = lo; i < hi; ++i) {
                mc::point1 P; mc::big k;
                if (!g1_load(P, pts96 + 96 * i)) { bad = 1; continue; }
                scalar_from32(k, scalars32 + 32 * i);
                mc::multiply(P, k);
                mc::add(part[t], P);
            }
        });
    }
    for (auto& x : th) x.join();
    for (size_t t = 1; t < T; ++t) mc::add(part[0], part[t]);
    g1_store(out, part[0], out_fmt);
    return bad ? -2 : 0;
}

// sum_of_products(point1&, n, point1*, const big*) — src/miracl_core_interface.cpp:134-137 -> ECP_muln
int ref_g1_sum_of_products(int n, const uint8_t* pts96, const uint8_t* scalars32, uint8_t* out, int out_fmt) {
    std::vector<mc::point1> P(n);
    std::vector<int64_t> K((size_t)n * 7);
    for (int i = 0; i < n; ++i) {
        if (!g1_load(P[i], pts96 + 96 * (size_t)i)) return -2;
        mc::big k; scalar_from32(k, scalars32 + 32 * (size_t)i);
        // ECP_muln takes scalars as given (no reduction); reduce mod r here like the headers' Zp values
        BIG r; BIG_rcopy(r, CURVE_Order); BIG_mod(k, r);
        std::memcpy(&K[(size_t)i * 7], k, sizeof(mc::big));
    }
    mc::point1 R;
    mc::sum_of_products(R, n, P.data(), (const mc::big*)K.data());
    g1_store(out, R, out_fmt);
    return 0;
}

// negate(point2&) / sub(point2&, point2&) / equal(point2&, point2&) — src/miracl_core_interface.cpp:207-226 (ECP2_neg, ECP2_sub, ECP2_equals)
int ref_g2_neg_batch(size_t n, const uint8_t* a192, uint8_t* out, int out_fmt) {
    for (size_t i = 0; i < n; ++i) {
        mc::point2 A;
        if (!g2_load(A, a192 + 192 * i)) return -2;
        mc::negate(A);
        g2_store(out + (size_t)out_fmt * i, A, out_fmt);
    }
    return 0;
}
int ref_g2_sub_batch(size_t n, const uint8_t* a192, const uint8_t* b192, uint8_t* out, int out_fmt) {
    for (size_t i = 0; i < n; ++i) {
        mc::point2 A, B;
        if (!g2_load(A, a192 + 192 * i) || !g2_load(B, b192 + 192 * i)) return -2;
        mc::sub(A, B);
        g2_store(out + (size_t)out_fmt * i, A, out_fmt);
    }
    return 0;
}
int ref_g2_equal_batch(size_t n, const uint8_t* a192, const uint8_t* b192, uint8_t* out) {
    for (size_t i = 0; i < n; ++i) {
        mc::point2 A, B;
        if (!g2_load(A, a192 + 192 * i) || !g2_load(B, b192 + 192 * i)) return -2;
        out[i] = (uint8_t)mc::equal(A, B);
    }
    return 0;
}
int ref_g2_equal_sum_batch(size_t n, const uint8_t* a192, const uint8_t* b192, const uint8_t* c192, uint8_t* out) {
    for (size_t i = 0; i < n; ++i) {
        mc::point2 A, B, C;
        if (!g2_load(A, a192 + 192 * i) || !g2_load(B, b192 + 192 * i) || !g2_load(C, c192 + 192 * i)) return -2;
        mc::add(A, B);
        out[i] = (uint8_t)mc::equal(A, C);
    }
    return 0;
}

// multiply(point2&, const big&) — src/miracl_core_interface.cpp:202-205 -> PAIR_G2mul
int ref_g2_mul_batch(size_t n, const uint8_t* pts192, const uint8_t* scalars32, uint8_t* out, int out_fmt, int nthreads) {
    if (out_fmt != 97 && out_fmt != 192) return -1;
    int bad = 0;
    par_for(n, nthreads, [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i) {
            mc::point2 P; mc::big k;
            if (!g2_load(P, pts192 + 192 * i)) { bad = 1; std::memset(out + (size_t)out_fmt * i, 0xff, out_fmt); continue; }
            scalar_from32(k, scalars32 + 32 * i);
            mc::multiply(P, k);
            g2_store(out + (size_t)out_fmt * i, P, out_fmt);
        }
    });
    return bad ? -2 : 0;
}
int ref_g2_add_batch(size_t n, const uint8_t* a192, const uint8_t* b192, uint8_t* out, int out_fmt) {
    for (size_t i = 0; i < n; ++i) {
        mc::point2 A, B;
        if (!g2_load(A, a192 + 192 * i) || !g2_load(B, b192 + 192 * i)) return -2;
        mc::add(A, B);
        g2_store(out + (size_t)out_fmt * i, A, out_fmt);
    }
    return 0;
}
int ref_g2_decompress_batch(size_t n, const uint8_t* in97, uint8_t* out192, uint8_t* status) {
    for (size_t i = 0; i < n; ++i) {
        const uint8_t* s = in97 + 97 * i;
        mc::point2 P;
        if (s[0] == 0) { std::memset(out192 + 192 * i, 0, 192); status[i] = 1; continue; }
        char buf[97]; std::memcpy(buf, s, 97);
        mc::bytes_view v{97, 97, buf};
        int ok = mc::from_bytes(P, v);
        status[i] = (uint8_t)ok;
        if (ok) g2_store(out192 + 192 * i, P, 192); else std::memset(out192 + 192 * i, 0, 192);
    }
    return 0;
}
int ref_g2_compress_batch(size_t n, const uint8_t* in192, uint8_t* out97) {
    for (size_t i = 0; i < n; ++i) {
        mc::point2 P;
        if (!g2_load(P, in192 + 192 * i)) return -2;
        g2_store(out97 + 97 * i, P, 97);
    }
    return 0;
}

// pair_ate + pair_final_exponentiation + to_bytes — src/miracl_core_interface.cpp:276-284, 246-249
int ref_pair_batch(size_t n, const uint8_t* g1_96, const uint8_t* g2_192, uint8_t* gt576, int nthreads) {
    int bad = 0;
    par_for(n, nthreads, [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i) {
            mc::point1 P; mc::point2 Q; mc::fp12 f;
            if (!g1_load(P, g1_96 + 96 * i) || !g2_load(Q, g2_192 + 192 * i)) { bad = 1; continue; }
            mc::pair_ate(f, Q, P);
            mc::pair_final_exponentiation(f);
            mc::bytes_view v{0, 576, (char*)gt576 + 576 * i};
            mc::to_bytes(v, f);
        }
    });
    return bad ? -2 : 0;
}

// pair_ate alone (Miller value; not canonical in general, but a well-defined field element whose
// FP12_toOctet bytes can be compared) and pair_final_exponentiation alone, on 576-byte values
int ref_miller_batch(size_t n, const uint8_t* g1_96, const uint8_t* g2_192, uint8_t* out576) {
    for (size_t i = 0; i < n; ++i) {
        mc::point1 P; mc::point2 Q; mc::fp12 f;
        if (!g1_load(P, g1_96 + 96 * i) || !g2_load(Q, g2_192 + 192 * i)) return -2;
        mc::pair_ate(f, Q, P);
        mc::bytes_view v{0, 576, (char*)out576 + 576 * i};
        mc::to_bytes(v, f);
    }
    return 0;
}
int ref_fexp_batch(size_t n, const uint8_t* in576, uint8_t* out576) {
    for (size_t i = 0; i < n; ++i) {
        mc::fp12 f; char buf[576];
        std::memcpy(buf, in576 + 576 * i, 576);
        mc::bytes_view vi{576, 576, buf};
        mc::from_bytes(f, vi);
        mc::pair_final_exponentiation(f);
        mc::bytes_view vo{0, 576, (char*)out576 + 576 * i};
        mc::to_bytes(vo, f);
    }
    return 0;
}

// e(a1,a2) == e(b1,b2) exactly as include/crypto12381/liner_pair.hpp:339-350:
// two Miller loops, conjugate, multiply, ONE final exponentiation, is_unity.
int ref_pair_eq_batch(size_t n, const uint8_t* a1, const uint8_t* a2, const uint8_t* b1, const uint8_t* b2, uint8_t* ok, int nthreads) {
    int bad = 0;
    par_for(n, nthreads, [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i) {
            mc::point1 P, R; mc::point2 Q, S; mc::fp12 f, g, gc;
            if (!g1_load(P, a1 + 96 * i) || !g2_load(Q, a2 + 192 * i) ||
                !g1_load(R, b1 + 96 * i) || !g2_load(S, b2 + 192 * i)) { bad = 1; continue; }
            mc::pair_ate(f, Q, P);
            mc::pair_ate(g, S, R);
            mc::conjugate(gc, g);
            mc::multiply(f, gc);
            mc::pair_final_exponentiation(f);
            ok[i] = mc::is_unity(f) ? 1 : 0;
        }
    });
    return bad ? -2 : 0;
}

// pair_double_ate + fexp — src/miracl_core_interface.cpp:286-289
int ref_pair2_batch(size_t n, const uint8_t* a1, const uint8_t* a2, const uint8_t* b1, const uint8_t* b2, uint8_t* gt576) {
    for (size_t i = 0; i < n; ++i) {
        mc::point1 P, R; mc::point2 Q, S; mc::fp12 f;
        if (!g1_load(P, a1 + 96 * i) || !g2_load(Q, a2 + 192 * i) ||
            !g1_load(R, b1 + 96 * i) || !g2_load(S, b2 + 192 * i)) return -2;
        mc::pair_double_ate(f, Q, P, S, R);
        mc::pair_final_exponentiation(f);
        mc::bytes_view v{0, 576, (char*)gt576 + 576 * i};
        mc::to_bytes(v, f);
    }
    return 0;
}

// GT ops on canonical bytes: op 0 multiply, 1 conjugate(a), 2 pow(a, scalar32 in b)
int ref_gt_op_batch(int op, size_t n, const uint8_t* a576, const uint8_t* b, uint8_t* out576) {
    for (size_t i = 0; i < n; ++i) {
        mc::fp12 x, y, r;
        char buf[576];
        std::memcpy(buf, a576 + 576 * i, 576);
        mc::bytes_view va{576, 576, buf};
        mc::from_bytes(x, va);
        if (op == 0) {
            char bb[576]; std::memcpy(bb, b + 576 * i, 576);
            mc::bytes_view vb{576, 576, bb};
            mc::from_bytes(y, vb);
            mc::multiply(x, y);
            r = x;
        } else if (op == 1) {
            mc::conjugate(r, x);
        } else if (op == 2) {
            mc::big k; scalar_from32(k, b + 32 * i);
            mc::pow(r, x, k);
        } else return -1;
        mc::bytes_view vo{0, 576, (char*)out576 + 576 * i};
        mc::to_bytes(vo, r);
    }
    return 0;
}

// equal(fp12&, fp12&) — src/miracl_core_interface.cpp:266-269 (FP12_equals); with_product: compare a * b (computed through the
// seam's multiply) with c instead of a with b
int ref_gt_equal_batch(size_t n, const uint8_t* a576, const uint8_t* b576, const uint8_t* c576, uint8_t* out) {
    for (size_t i = 0; i < n; ++i) {
        mc::fp12 x, y, z;
        char ba[576], bb[576], bc[576];
        std::memcpy(ba, a576 + 576 * i, 576); std::memcpy(bb, b576 + 576 * i, 576);
        mc::bytes_view va{576, 576, ba}, vb{576, 576, bb};
        mc::from_bytes(x, va); mc::from_bytes(y, vb);
        if (c576) {
            std::memcpy(bc, c576 + 576 * i, 576);
            mc::bytes_view vc{576, 576, bc};
            mc::from_bytes(z, vc);
            mc::multiply(x, y);
            out[i] = (uint8_t)mc::equal(x, z);
        } else out[i] = (uint8_t)mc::equal(x, y);
    }
    return 0;
}

// The reference's own seeded scalar stream: RandomEngine(seed) + random_in(·, r)
// (src/random.cpp:14-24, src/miracl_core_interface.cpp:65-69) — lets tests reproduce the
// inputs of unit-tests/*.cpp seeds such as "pairing bilinearity seed".
int ref_random_scalars(const char* seed, int seed_len, size_t n, uint8_t* out32) {
    crypto12381::RandomEngine rng{std::span<const char>(seed, (size_t)seed_len)};
    mc::big r; BIG_rcopy(r, CURVE_Order);
    for (size_t i = 0; i < n; ++i) {
        mc::big k; char buf[48];
        mc::random_in(k, r, rng);
        mc::to_bytes(buf, k);
        std::memcpy(out32 + 32 * i, buf + 16, 32);
    }
    return 0;
}

// sha3 KAT hook (unit-tests/miracl_core_interface.cpp:10-33)
int ref_sha3_512(const uint8_t* msg, size_t len, uint8_t out[64]) {
    mc::sha3_state st;
    mc::sha3_init(st, 64);
    for (size_t i = 0; i < len; ++i) mc::sha3_process(st, msg[i]);
    mc::sha3_hash(st, (char*)out);
    return 0;
}

// G1Point::from_hash from the digest on (include/crypto12381/g1_point.hpp:219-234): from_bytes(big2) ->
// fixed_time_mod(x, dbig, p, 512 - 381) -> residue -> map_to_point -> multiply_cofactor -> to_bytes.
int ref_g1_from_hash_batch(size_t n, const uint8_t* digests64, uint8_t* out, int out_fmt) {
    static const uint8_t P_BE[48] = {
        0x1a, 0x01, 0x11, 0xea, 0x39, 0x7f, 0xe6, 0x9a, 0x4b, 0x1b, 0xa7, 0xb6, 0x43, 0x4b, 0xac, 0xd7,
        0x64, 0x77, 0x4b, 0x84, 0xf3, 0x85, 0x12, 0xbf, 0x67, 0x30, 0xd2, 0xa0, 0xf6, 0xb0, 0xf6, 0x24,
        0x1e, 0xab, 0xff, 0xfe, 0xb1, 0x53, 0xff, 0xff, 0xb9, 0xfe, 0xff, 0xff, 0xff, 0xff, 0xaa, 0xab};
    mc::big modulus;
    mc::from_bytes(modulus, (const char*)P_BE);
    for (size_t i = 0; i < n; ++i) {
        mc::big2 dbig;
        mc::from_bytes(dbig, (const char*)digests64 + 64 * i, 64);
        mc::big x;
        mc::fixed_time_mod(x, dbig, modulus, 64 * 8 - 381);
        mc::fp f;
        mc::residue(f, x);
        mc::point1 P;
        mc::map_to_point(P, f);
        mc::multiply_cofactor(P);
        g1_store(out + (size_t)out_fmt * i, P, out_fmt);
    }
    return 0;
}

// residue + map_to_point alone (no cofactor): 48-byte field element -> affine point of E
int ref_g1_map_to_point_batch(size_t n, const uint8_t* u48, uint8_t* out96) {
    for (size_t i = 0; i < n; ++i) {
        mc::big x;
        mc::from_bytes(x, (const char*)u48 + 48 * i);
        mc::fp f;
        mc::residue(f, x);
        mc::point1 P;
        mc::map_to_point(P, f);
        g1_store(out96 + 96 * i, P, 96);
    }
    return 0;
}

// Zp helpers as zp_number.hpp evaluates them through the boundary: operands are reduced mod r first.
// op 0 mul (multiply + mod), 1 add, 2 sub, 3 neg (mod_negate), 4 inverse (mod_inverse, 0 -> 0)
static void zp_load(mc::big& k, const uint8_t* s32, const mc::big& order) {
    scalar_from32(k, s32);
    mc::big2 d;
    char buf[96];
    std::memset(buf, 0, 96);
    std::memcpy(buf + 64, s32, 32);
    mc::from_bytes(d, buf, 96);
    mc::mod(k, d, order);
}
static void zp_store(uint8_t* out32, const mc::big& k) {
    char buf[48];
    mc::to_bytes(buf, k);
    std::memcpy(out32, buf + 16, 32);
}
static void zp_order(mc::big& order) {
    static const uint8_t R_BE[48] = {
        0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
        0x73, 0xed, 0xa7, 0x53, 0x29, 0x9d, 0x7d, 0x48, 0x33, 0x39, 0xd8, 0x08, 0x09, 0xa1, 0xd8, 0x05,
        0x53, 0xbd, 0xa4, 0x02, 0xff, 0xfe, 0x5b, 0xfe, 0xff, 0xff, 0xff, 0xff, 0x00, 0x00, 0x00, 0x01};
    mc::from_bytes(order, (const char*)R_BE);
}
int ref_zp_op_batch(int op, size_t n, const uint8_t* a32, const uint8_t* b32, uint8_t* out32) {
    mc::big order;
    zp_order(order);
    for (size_t i = 0; i < n; ++i) {
        mc::big a, b, r;
        zp_load(a, a32 + 32 * i, order);
        if (b32 && op <= 2) zp_load(b, b32 + 32 * i, order);
        mc::big2 d;
        switch (op) {
            case 0: mc::multiply(d, a, b); mc::mod(r, d, order); break;
            case 1: case 2: {
                if (op == 2) mc::mod_negate(b, b, order);
                for (int j = 0; j < 7; ++j) r[j] = a[j] + b[j];
                mc::normalize(r);
                if (mc::compare(r, order) >= 0) { for (int j = 0; j < 7; ++j) r[j] -= order[j]; mc::normalize(r); }
                break;
            }
            case 3: mc::mod_negate(r, a, order); if (mc::compare(r, order) >= 0) { for (int j = 0; j < 7; ++j) r[j] -= order[j]; mc::normalize(r); } break;
            case 4: mc::mod_inverse(r, a, order); break;
            default: return -1;
        }
        zp_store(out32 + 32 * i, r);
    }
    return 0;
}
// Zp from_hash (zp_number.hpp:540-548)
int ref_zp_from_hash_batch(size_t n, const uint8_t* digests64, uint8_t* out32) {
    mc::big order;
    zp_order(order);
    for (size_t i = 0; i < n; ++i) {
        mc::big2 dbig;
        mc::from_bytes(dbig, (const char*)digests64 + 64 * i, 64);
        mc::big x;
        mc::fixed_time_mod(x, dbig, order, 64 * 8 - 255);
        zp_store(out32 + 32 * i, x);
    }
    return 0;
}


// ---- threaded forms of the split pairing (CPU baselines of bench.py; same loops as ref_miller_batch / ref_fexp_batch)
int ref_miller_batch_t(size_t n, const uint8_t* g1_96, const uint8_t* g2_192, uint8_t* out576, int nthreads) {
    int bad = 0;
    par_for(n, nthreads, [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i) {
            mc::point1 P; mc::point2 Q; mc::fp12 f;
            if (!g1_load(P, g1_96 + 96 * i) || !g2_load(Q, g2_192 + 192 * i)) { bad = 1; continue; }
            mc::pair_ate(f, Q, P);
            mc::bytes_view v{0, 576, (char*)out576 + 576 * i};
            mc::to_bytes(v, f);
        }
    });
    return bad ? -2 : 0;
}
int ref_fexp_batch_t(size_t n, const uint8_t* in576, uint8_t* out576, int nthreads) {
    par_for(n, nthreads, [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i) {
            mc::fp12 f; char buf[576];
            std::memcpy(buf, in576 + 576 * i, 576);
            mc::bytes_view vi{576, 576, buf};
            mc::from_bytes(f, vi);
            mc::pair_final_exponentiation(f);
            mc::bytes_view vo{0, 576, (char*)out576 + 576 * i};
            mc::to_bytes(vo, f);
        }
    });
    return 0;
}

// ---- BBS+ verification, the reference's op sequence (examples/bbs-plus/src/bbs+.cpp:57-73) at boundary level:
//   pair(A, w * (g2^x)) == pair(g1 * (h0^r) * Π[n](h[i]^m[i]), g2)
// g2^x -> multiply(point2&, big); w * . -> add; h0^r, h[i]^m[i] -> multiply(point1&, big); products -> add;
// == -> two pair_ate, conjugate, multiply, one final exponentiation, is_unity (liner_pair.hpp:339-350).
// Messages are message-major: m32[(i * n + j) * 32] is block i of signature j (the layout of c12381_bbs_plus_verify_batch).
static int bbs_verify_one(mc::point1& G1p, mc::point2& G2p, mc::point1& H0, std::vector<mc::point1>& H, mc::point2& W,
                          mc::point1& A, const mc::big& x, const mc::big& r, const std::vector<mc::big>& m) {
    mc::point2 Q = G2p;
    mc::multiply(Q, x);
    mc::point2 Wc = W;
    mc::add(Wc, Q);                                    // w * g2^x
    mc::point1 B = G1p, T = H0;
    mc::multiply(T, r);
    mc::add(B, T);
    for (size_t i = 0; i < m.size(); ++i) {
        mc::point1 Hi = H[i];
        mc::multiply(Hi, m[i]);
        mc::add(B, Hi);
    }
    mc::fp12 f, g, gc;
    mc::point2 G2c = G2p;
    mc::pair_ate(f, Wc, A);
    mc::pair_ate(g, G2c, B);
    mc::conjugate(gc, g);
    mc::multiply(f, gc);
    mc::pair_final_exponentiation(f);
    return mc::is_unity(f) ? 1 : 0;
}
int ref_bbs_plus_verify_batch(size_t n, size_t nmsg, const uint8_t* g1_96, const uint8_t* g2_192, const uint8_t* h0_96, const uint8_t* h_96,
                              const uint8_t* w_192, const uint8_t* A_96, const uint8_t* x32, const uint8_t* r32, const uint8_t* m32,
                              uint8_t* ok, int nthreads) {
    mc::point1 G1p, H0; mc::point2 G2p, W;
    std::vector<mc::point1> H(nmsg);
    if (!g1_load(G1p, g1_96) || !g1_load(H0, h0_96) || !g2_load(G2p, g2_192) || !g2_load(W, w_192)) return -2;
    for (size_t i = 0; i < nmsg; ++i) if (!g1_load(H[i], h_96 + 96 * i)) return -2;
    par_for(n, nthreads, [&](size_t lo, size_t hi) {
        for (size_t j = lo; j < hi; ++j) {
            mc::point1 A;
            if (!g1_load(A, A_96 + 96 * j)) { ok[j] = 0xff; continue; }
            mc::big x, r; std::vector<mc::big> m(nmsg);
            scalar_from32(x, x32 + 32 * j); scalar_from32(r, r32 + 32 * j);
            for (size_t i = 0; i < nmsg; ++i) scalar_from32(m[i], m32 + 32 * (i * n + j));
            mc::point1 g = G1p, h0 = H0; mc::point2 g2 = G2p, w = W;
            std::vector<mc::point1> h = H;
            ok[j] = (uint8_t)bbs_verify_one(g, g2, h0, h, w, A, x, r, m);
        }
    });
    return 0;
}

// ---- the same from the WIRE formats the example exchanges (bbs+.cpp:57-73 with the header layer's decoders restated):
//   pp.g1_g2_h0 = serialize(g1, g2, h0) = 49 + 97 + 49 bytes, pp.h = 49 bytes each, pk = 97 bytes,
//   signature = serialize(A, x, r) = 49 + 48 + 48 bytes, message = raw bytes.
// parse<G1>/parse<G2> (g1_point.hpp:87-111, g2_point.hpp:71-95): leading 0x00 = infinity, otherwise from_bytes, failure throws;
// parse<Zp> (zp_number.hpp:226-236): 48 big-endian bytes, value >= r throws;
// encode_to<Zp> (zp_number.hpp:1011-1037): 31-byte units, byte 16 of the 48-byte field set to 1, a short last unit is
// left-aligned in its 31 bytes; more units than h entries throws "message is too long".
// ok[j] = 1 / 0 = the boolean verify() returns, 0xff = the reference would throw (malformed signature); return -2 when the
// public material itself does not parse or the message is too long (verify() throws for every signature then).
static int wire_g1(mc::point1& P, const uint8_t* b49) {
    if (b49[0] == 0) { mc::get_infinity(P); return 1; }
    char buf[49]; std::memcpy(buf, b49, 49);
    mc::bytes_view v{49, 49, buf};
    return mc::from_bytes(P, v);
}
static int wire_g2(mc::point2& P, const uint8_t* b97) {
    if (b97[0] == 0) { mc::get_infinity(P); return 1; }
    char buf[97]; std::memcpy(buf, b97, 97);
    mc::bytes_view v{97, 97, buf};
    return mc::from_bytes(P, v);
}
static int wire_zp(mc::big& k, const uint8_t* b48) {
    mc::from_bytes(k, (const char*)b48);
    mc::big order; BIG_rcopy(order, CURVE_Order);
    return mc::compare(k, order) < 0;
}
int ref_bbs_plus_verify_wire_batch(size_t n, size_t nh, size_t msg_len, const uint8_t* g1_g2_h0_195, const uint8_t* h49, const uint8_t* pk97,
                                   const uint8_t* sig145, const uint8_t* msgs, uint8_t* ok, int nthreads) {
    mc::point1 G1p, H0; mc::point2 G2p, W;
    const size_t nblk = (msg_len + 30) / 31;
    if (nblk > nh) return -2;
    std::vector<mc::point1> H(nblk);
    if (!wire_g1(G1p, g1_g2_h0_195) || !wire_g2(G2p, g1_g2_h0_195 + 49) || !wire_g1(H0, g1_g2_h0_195 + 146) || !wire_g2(W, pk97)) return -2;
    for (size_t i = 0; i < nblk; ++i) if (!wire_g1(H[i], h49 + 49 * i)) return -2;
    par_for(n, nthreads, [&](size_t lo, size_t hi) {
        for (size_t j = lo; j < hi; ++j) {
            const uint8_t* s = sig145 + 145 * j;
            mc::point1 A; mc::big x, r;
            if (!wire_g1(A, s) || !wire_zp(x, s + 49) || !wire_zp(r, s + 97)) { ok[j] = 0xff; continue; }
            std::vector<mc::big> m(nblk);
            const uint8_t* msg = msgs + msg_len * j;
            for (size_t i = 0; i < nblk; ++i) {
                uint8_t buf[48] = {0};
                buf[16] = 1;
                const size_t len = (i + 1) * 31 <= msg_len ? 31 : msg_len - i * 31;
                std::memcpy(buf + 17, msg + 31 * i, len);
                mc::from_bytes(m[i], (const char*)buf);
            }
            mc::point1 g = G1p, h0 = H0; mc::point2 g2 = G2p, w = W;
            std::vector<mc::point1> h = H;
            ok[j] = (uint8_t)bbs_verify_one(g, g2, h0, h, w, A, x, r, m);
        }
    });
    return 0;
}
// encode_to<Zp> alone (zp_number.hpp:1011-1037): msg_len bytes -> ceil(msg_len / 31) scalars of 32 bytes (the 48-byte field's low 32)
int ref_encode_to_zp(size_t msg_len, const uint8_t* msg, uint8_t* out32) {
    const size_t nblk = (msg_len + 30) / 31;
    for (size_t i = 0; i < nblk; ++i) {
        uint8_t buf[48] = {0};
        buf[16] = 1;
        const size_t len = (i + 1) * 31 <= msg_len ? 31 : msg_len - i * 31;
        std::memcpy(buf + 17, msg + 31 * i, len);
        mc::big k; char o[48];
        mc::from_bytes(k, (const char*)buf);
        mc::to_bytes(o, k);
        std::memcpy(out32 + 32 * i, o + 16, 32);
    }
    return 0;
}

} // extern "C"
