/* TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the BLS12-381 hot path of Adttil/crypto12381 (vendored
 * MIRACL-core behind src/miracl_core_interface.cpp).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this; the product path (crypto12381_amd/csrc)
 * never includes, links or calls it.
 *
 * PARITY PINNED: every entry point is checked byte-for-byte against the compiled reference
 * (oracle/_ref/libc12381_ref.so, built by oracle/Makefile from /root/reference) and against
 * the golden vectors in tests/golden/ that were generated from it (oracle/gen_golden.py).
 *
 * The restatement follows the reference's ALGORITHMS (tower, formulas, loop schedules,
 * encodings) function by function — file:line citations are relative to
 * /root/reference/3rd-party/miracl-core/ unless stated — but not its number format:
 * field elements here are 6 x 64-bit limbs in Montgomery form with R = 2^384, always fully
 * reduced, instead of MIRACL's 7 x 58-bit lazily-reduced limbs.  Constants
 * are derived from the public integers p, r, x (the curve parameter) at start-up.
 */
#include "c12381_oracle.h"
#include <string.h>
#include <stdlib.h>
#include <pthread.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t l[6]; } fp;
typedef struct { fp a, b; } fp2;           /* a + b*i,  i^2 = -1           (fp2_BLS12381.h) */
typedef struct { fp2 a, b; } fp4;          /* a + b*s,  s^2 = 1 + i        (fp4_BLS12381.h) */
typedef struct { fp4 a, b, c; } fp12;      /* a + b*w + c*w^2, w^3 = s     (fp12_BLS12381.h:32-38) */
typedef struct { fp x, y, z; } g1p;        /* homogeneous projective (X:Y:Z), ecp_BLS12381.h:114-123 */
typedef struct { fp2 x, y, z; } g2p;       /* on the M-type twist y^2 = x^3 + 4(1+i), ecp2_BLS12381.h:42-48 */

/* ---- public integers (rom_field_BLS12381.cpp:51 Modulus, rom_curve_BLS12381.cpp:80 CURVE_Order,
 *      :87 CURVE_Bnx) re-expressed in 64-bit limbs ---- */
static const uint64_t P[6] = {0xb9feffffffffaaabULL, 0x1eabfffeb153ffffULL, 0x6730d2a0f6b0f624ULL,
                              0x64774b84f38512bfULL, 0x4b1ba7b6434bacd7ULL, 0x1a0111ea397fe69aULL};
static const uint64_t N0 = 0x89f3fffcfffcfffdULL;            /* -p^-1 mod 2^64 */
static const fp R2 = {{0xf4df1f341c341746ULL, 0x0a76e6a609d104f1ULL, 0x8de5476c4c95b6d5ULL,
                       0x67eb88a9939d83c0ULL, 0x9a793e85b519952dULL, 0x11988fe592cae3aaULL}};
static const fp ONE = {{0x760900000002fffdULL, 0xebf4000bc40c0002ULL, 0x5f48985753c758baULL,
                        0x77ce585370525745ULL, 0x5c071a97a256ec6dULL, 0x15f65ec3fa80e493ULL}};
static const uint64_t RORD[4] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL,
                                 0x73eda753299d7d48ULL};
static const uint64_t BNX = 0xd201000000010000ULL;           /* |x|; SIGN_OF_X = NEGATIVEX (config_curve_BLS12381.h:50) */

/* ------------------------------------------------------------------ Fp (fp_BLS12381.cpp) */
static int fp_is_zero(const fp* a) { uint64_t t = 0; for (int i = 0; i < 6; i++) t |= a->l[i]; return t == 0; }
static int fp_eq(const fp* a, const fp* b) { uint64_t t = 0; for (int i = 0; i < 6; i++) t |= a->l[i] ^ b->l[i]; return t == 0; }
static int raw_geq_p(const uint64_t* t) {
    for (int i = 5; i >= 0; i--) { if (t[i] > P[i]) return 1; if (t[i] < P[i]) return 0; }
    return 1;
}
static void raw_sub_p(uint64_t* t) {
    u128 bw = 0;
    for (int i = 0; i < 6; i++) { u128 d = (u128)t[i] - P[i] - bw; t[i] = (uint64_t)d; bw = (d >> 64) & 1; }
}
/* FP_add :485 (here with immediate reduction instead of the XES excess counter) */
static void fp_add(fp* r, const fp* a, const fp* b) {
    u128 c = 0; uint64_t t[6];
    for (int i = 0; i < 6; i++) { c += (u128)a->l[i] + b->l[i]; t[i] = (uint64_t)c; c >>= 64; }
    if (c || raw_geq_p(t)) raw_sub_p(t);
    memcpy(r->l, t, sizeof t);
}
/* FP_sub :500 */
static void fp_sub(fp* r, const fp* a, const fp* b) {
    u128 bw = 0; uint64_t t[6];
    for (int i = 0; i < 6; i++) { u128 d = (u128)a->l[i] - b->l[i] - bw; t[i] = (uint64_t)d; bw = (d >> 64) & 1; }
    if (bw) { u128 c = 0; for (int i = 0; i < 6; i++) { c += (u128)t[i] + P[i]; t[i] = (uint64_t)c; c >>= 64; } }
    memcpy(r->l, t, sizeof t);
}
/* FP_neg :588 */
static void fp_neg(fp* r, const fp* a) { fp z; memset(&z, 0, sizeof z); fp_sub(r, &z, a); }
/* FP_mul :396 = BIG_mul (big_B384_58.cpp:570) + BIG_monty (:836), here as one CIOS pass */
static void fp_mul(fp* r, const fp* a, const fp* b) {
    uint64_t t[8] = {0};
    for (int i = 0; i < 6; i++) {
        u128 c = 0;
        for (int j = 0; j < 6; j++) { c += (u128)a->l[j] * b->l[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
        c += t[6]; t[6] = (uint64_t)c; t[7] = (uint64_t)(c >> 64);
        uint64_t m = t[0] * N0;
        c = ((u128)m * P[0] + t[0]) >> 64;
        for (int j = 1; j < 6; j++) { c += (u128)m * P[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
        c += t[6]; t[5] = (uint64_t)c; t[6] = t[7] + (uint64_t)(c >> 64);
    }
    if (t[6] || raw_geq_p(t)) raw_sub_p(t);
    memcpy(r->l, t, 6 * sizeof(uint64_t));
}
static void fp_sqr(fp* r, const fp* a) { fp_mul(r, a, a); }                 /* FP_sqr :466 */
static void fp_dbl(fp* r, const fp* a) { fp_add(r, a, a); }
/* FP_imul :420 — multiplication by a small integer */
static void fp_imul(fp* r, const fp* a, int k) {
    fp acc, base = *a; memset(&acc, 0, sizeof acc);
    while (k) { if (k & 1) fp_add(&acc, &acc, &base); fp_dbl(&base, &base); k >>= 1; }
    *r = acc;
}
/* FP_div2 :521 */
static void fp_div2(fp* r, const fp* a) {
    uint64_t t[7]; memcpy(t, a->l, 48); t[6] = 0;
    if (t[0] & 1) { u128 c = 0; for (int i = 0; i < 6; i++) { c += (u128)t[i] + P[i]; t[i] = (uint64_t)c; c >>= 64; } t[6] = (uint64_t)c; }
    for (int i = 0; i < 6; i++) r->l[i] = (t[i] >> 1) | (t[i + 1] << 63);
}
/* FP_pow :631 (plain square-and-multiply; exponent little-endian 64-bit words) */
static void fp_pow(fp* r, const fp* a, const uint64_t* e, int nw) {
    fp acc = ONE, base = *a;
    for (int i = 0; i < nw * 64; i++) {
        if ((e[i / 64] >> (i % 64)) & 1) fp_mul(&acc, &acc, &base);
        fp_sqr(&base, &base);
    }
    *r = acc;
}
static uint64_t E_PM2[6], E_PP1D4[6], E_PM1D6[6], E_PM1D2[6];
/* FP_inv :817 — Fermat: a^(p-2) */
static void fp_inv(fp* r, const fp* a) { fp_pow(r, a, E_PM2, 6); }
/* FP_qr :800-813 — a^((p-1)/2) == 1; note 0 is reported as a NON-residue (0 != 1), as in MIRACL */
static int fp_qr(const fp* a) {
    fp t;
    fp_pow(&t, a, E_PM1D2, 6); return fp_eq(&t, &ONE);
}
/* FP_sqrt :842 — p = 3 mod 4: a^((p+1)/4); caller checks residuosity */
static void fp_sqrt(fp* r, const fp* a) { fp_pow(r, a, E_PP1D4, 6); }
/* FP_nres :223 / BIG_fromBytes big_B384_58.cpp:186: 48 big-endian bytes -> Montgomery form (value taken mod p) */
static void fp_from_bytes(fp* r, const uint8_t* b) {
    fp t;
    for (int i = 0; i < 6; i++) { uint64_t w = 0; for (int j = 0; j < 8; j++) w = (w << 8) | b[(5 - i) * 8 + j]; t.l[i] = w; }
    fp_mul(r, &t, &R2);
}
/* FP_redc :234 + BIG_toBytes big_B384_58.cpp:171 */
static void fp_to_raw(uint64_t* out, const fp* a) { fp one, t; memset(&one, 0, sizeof one); one.l[0] = 1; fp_mul(&t, a, &one); memcpy(out, t.l, 48); }
static void fp_to_bytes(uint8_t* b, const fp* a) {
    uint64_t t[6]; fp_to_raw(t, a);
    for (int i = 0; i < 6; i++) for (int j = 0; j < 8; j++) b[(5 - i) * 8 + j] = (uint8_t)(t[i] >> (56 - 8 * j));
}
/* FP_sign :912-936 (non-BIG_ENDIAN_SIGN branch): parity of the canonical residue */
static int fp_sign(const fp* a) { uint64_t t[6]; fp_to_raw(t, a); return (int)(t[0] & 1); }
static void fp_set_int(fp* r, int v) { fp t; memset(&t, 0, sizeof t); t.l[0] = (uint64_t)v; fp_mul(r, &t, &R2); }

/* ------------------------------------------------------------------ Fp2 (fp2_BLS12381.cpp) */
static int fp2_is_zero(const fp2* a) { return fp_is_zero(&a->a) && fp_is_zero(&a->b); }
static int fp2_eq(const fp2* a, const fp2* b) { return fp_eq(&a->a, &b->a) && fp_eq(&a->b, &b->b); }
static void fp2_add(fp2* r, const fp2* a, const fp2* b) { fp_add(&r->a, &a->a, &b->a); fp_add(&r->b, &a->b, &b->b); }   /* :214 */
static void fp2_sub(fp2* r, const fp2* a, const fp2* b) { fp_sub(&r->a, &a->a, &b->a); fp_sub(&r->b, &a->b, &b->b); }   /* :222 */
static void fp2_neg(fp2* r, const fp2* a) { fp_neg(&r->a, &a->a); fp_neg(&r->b, &a->b); }                                /* :185 */
static void fp2_conj(fp2* r, const fp2* a) { r->a = a->a; fp_neg(&r->b, &a->b); }                                       /* :204 */
/* FP2_mul :266-302 (three products; the reference's lazy reduction changes nothing observable) */
static void fp2_mul(fp2* r, const fp2* x, const fp2* y) {
    fp A, B, C, D, E;
    fp_mul(&A, &x->a, &y->a); fp_mul(&B, &x->b, &y->b);
    fp_add(&C, &x->a, &x->b); fp_add(&D, &y->a, &y->b); fp_mul(&E, &C, &D);
    fp_sub(&r->a, &A, &B);
    fp_sub(&E, &E, &A); fp_sub(&r->b, &E, &B);
}
/* FP2_sqr :241-261: (a+b)(a-b) + 2ab i */
static void fp2_sqr(fp2* r, const fp2* x) {
    fp w1, w2, w3;
    fp_add(&w1, &x->a, &x->b); fp_sub(&w2, &x->a, &x->b); fp_dbl(&w3, &x->a);
    fp_mul(&r->b, &w3, &x->b); fp_mul(&r->a, &w1, &w2);
}
static void fp2_pmul(fp2* r, const fp2* x, const fp* s) { fp_mul(&r->a, &x->a, s); fp_mul(&r->b, &x->b, s); }           /* :231 */
static void fp2_imul(fp2* r, const fp2* x, int k) { fp_imul(&r->a, &x->a, k); fp_imul(&r->b, &x->b, k); }               /* :238 */
/* FP2_inv :334 */
static void fp2_inv(fp2* r, const fp2* x) {
    fp w1, w2;
    fp_sqr(&w1, &x->a); fp_sqr(&w2, &x->b); fp_add(&w1, &w1, &w2); fp_inv(&w1, &w1);
    fp_mul(&r->a, &x->a, &w1); fp_neg(&w1, &w1); fp_mul(&r->b, &x->b, &w1);
}
/* FP2_mul_ip :373-396 with QNRI = 0: multiply by (1 + i) */
static void fp2_mul_ip(fp2* w) { fp2 t = *w; fp_sub(&w->a, &t.a, &t.b); fp_add(&w->b, &t.a, &t.b); }
/* FP2_sign :168-181: parity of a, or of b when a == 0 */
static int fp2_sign(const fp2* w) { return fp_is_zero(&w->a) ? fp_sign(&w->b) : fp_sign(&w->a); }
/* FP2_qr :446 */
static int fp2_qr(const fp2* x) { fp n, t; fp_sqr(&n, &x->a); fp_sqr(&t, &x->b); fp_add(&n, &n, &t); return fp_qr(&n); }
/* FP2_sqrt :460-521 (complex method; returns the root of sign 0 like the reference) */
static void fp2_sqrt(fp2* w, const fp2* u) {
    fp w1, w2, w3, hb; fp2 t = *u;
    if (fp2_is_zero(&t)) { *w = t; return; }
    fp_sqr(&w1, &t.b); fp_sqr(&w2, &t.a); fp_add(&w1, &w1, &w2);
    fp_sqrt(&w1, &w1);                      /* sqrt(a^2 + b^2) */
    fp_add(&w2, &t.a, &w1); fp_div2(&w2, &w2);
    fp_div2(&hb, &t.b);
    int qr = fp_qr(&w2);
    if (!qr) fp_neg(&w2, &w2);              /* -1 is a non-residue: exactly one of +-w2 is a square */
    fp ra, rb;
    fp_sqrt(&ra, &w2);
    fp_inv(&w3, &w2); fp_mul(&w3, &w3, &ra); fp_mul(&rb, &w3, &hb);
    if (qr) { w->a = ra; w->b = rb; } else { w->a = rb; w->b = ra; }
    if (fp2_sign(w)) fp2_neg(w, w);
}
static void fp2_from_bytes(fp2* r, const uint8_t* b) { fp_from_bytes(&r->b, b); fp_from_bytes(&r->a, b + 48); }        /* :89-93, b first */
static void fp2_to_bytes(uint8_t* b, const fp2* x) { fp_to_bytes(b, &x->b); fp_to_bytes(b + 48, &x->a); }               /* :83-87 */

/* ------------------------------------------------------------------ Fp4 (fp4_BLS12381.cpp) */
static void fp4_add(fp4* r, const fp4* a, const fp4* b) { fp2_add(&r->a, &a->a, &b->a); fp2_add(&r->b, &a->b, &b->b); }
static void fp4_sub(fp4* r, const fp4* a, const fp4* b) { fp2_sub(&r->a, &a->a, &b->a); fp2_sub(&r->b, &a->b, &b->b); }
static void fp4_neg(fp4* r, const fp4* a) { fp2_neg(&r->a, &a->a); fp2_neg(&r->b, &a->b); }
static void fp4_conj(fp4* r, const fp4* a) { r->a = a->a; fp2_neg(&r->b, &a->b); }     /* :162 */
static void fp4_nconj(fp4* r, const fp4* a) { fp2_neg(&r->a, &a->a); r->b = a->b; }    /* :171 */
static int fp4_eq(const fp4* a, const fp4* b) { return fp2_eq(&a->a, &b->a) && fp2_eq(&a->b, &b->b); }
/* FP4_mul :274-304 */
static void fp4_mul(fp4* w, const fp4* x, const fp4* y) {
    fp2 t1, t2, t3, t4;
    fp2_mul(&t1, &x->a, &y->a); fp2_mul(&t2, &x->b, &y->b);
    fp2_add(&t3, &y->b, &y->a); fp2_add(&t4, &x->b, &x->a); fp2_mul(&t4, &t4, &t3);
    fp2_sub(&t4, &t4, &t1); fp2_sub(&w->b, &t4, &t2);
    fp2_mul_ip(&t2); fp2_add(&w->a, &t2, &t1);
}
/* FP4_sqr :243-271 */
static void fp4_sqr(fp4* w, const fp4* x) {
    fp2 t1, t2, t3;
    fp2_mul(&t3, &x->a, &x->b);
    t2 = x->b; fp2_add(&t1, &x->a, &x->b); fp2_mul_ip(&t2); fp2_add(&t2, &x->a, &t2);
    fp2 wa; fp2_mul(&wa, &t1, &t2);
    t2 = t3; fp2_mul_ip(&t2); fp2_add(&t2, &t2, &t3);
    fp2_sub(&w->a, &wa, &t2); fp2_add(&w->b, &t3, &t3);
}
/* FP4_inv :326-340 */
static void fp4_inv(fp4* w, const fp4* x) {
    fp2 t1, t2;
    fp2_sqr(&t1, &x->a); fp2_sqr(&t2, &x->b); fp2_mul_ip(&t2); fp2_sub(&t1, &t1, &t2); fp2_inv(&t1, &t1);
    fp2_mul(&w->a, &t1, &x->a); fp2_neg(&t1, &t1); fp2_mul(&w->b, &t1, &x->b);
}
/* FP4_times_i :343-356 (NEGATOWER): multiply by s */
static void fp4_times_i(fp4* w) { fp2 t = w->b; w->b = w->a; fp2_mul_ip(&t); w->a = t; }
/* FP4_frob :359-364 */
static void fp4_frob(fp4* w, const fp2* f) { fp2_conj(&w->a, &w->a); fp2_conj(&w->b, &w->b); fp2_mul(&w->b, f, &w->b); }
static void fp4_pmul(fp4* w, const fp4* x, const fp2* s) { fp2_mul(&w->a, &x->a, s); fp2_mul(&w->b, &x->b, s); }      /* :210 */
static void fp4_from_bytes(fp4* r, const uint8_t* b) { fp2_from_bytes(&r->b, b); fp2_from_bytes(&r->a, b + 96); }      /* :67-71 */
static void fp4_to_bytes(uint8_t* b, const fp4* x) { fp2_to_bytes(b, &x->b); fp2_to_bytes(b + 96, &x->a); }            /* :61-65 */

/* ------------------------------------------------------------------ Fp12 (fp12_BLS12381.cpp) */
static fp2 FROB;     /* (1+i)^((p-1)/6) = Fra + Frb*i  (rom_field_BLS12381.cpp:56-57), derived at init */
static void fp12_one(fp12* w) { memset(w, 0, sizeof *w); w->a.a.a = ONE; }
static int fp12_eq(const fp12* x, const fp12* y) { return fp4_eq(&x->a, &y->a) && fp4_eq(&x->b, &y->b) && fp4_eq(&x->c, &y->c); }  /* :108 */
static int fp12_is_unity(const fp12* x) { fp12 o; fp12_one(&o); return fp12_eq(x, &o); }                                          /* :71 */
/* FP12_conj :117-123 */
static void fp12_conj(fp12* w, const fp12* x) { fp4_conj(&w->a, &x->a); fp4_nconj(&w->b, &x->b); fp4_conj(&w->c, &x->c); }
/* FP12_mul :246-299 (dense x dense; the sparse variants :304-619 compute the same product) */
static void fp12_mul(fp12* w, const fp12* y) {
    fp4 z0, z1, z2, z3, t0, t1;
    fp4_mul(&z0, &w->a, &y->a);
    fp4_mul(&z2, &w->b, &y->b);
    fp4_add(&t0, &w->a, &w->b); fp4_add(&t1, &y->a, &y->b); fp4_mul(&z1, &t0, &t1);
    fp4_add(&t0, &w->b, &w->c); fp4_add(&t1, &y->b, &y->c); fp4_mul(&z3, &t0, &t1);
    fp4_sub(&z1, &z1, &z0); fp4 wb; fp4_sub(&wb, &z1, &z2);          /* xa.yb + xb.ya */
    fp4_sub(&z3, &z3, &z2);                                          /* (xb+xc)(yb+yc) - xb.yb */
    fp4_sub(&z2, &z2, &z0);                                          /* xb.yb - xa.ya */
    fp4_add(&t0, &w->a, &w->c); fp4_add(&t1, &y->a, &y->c); fp4_mul(&t0, &t1, &t0);
    fp4_add(&z2, &z2, &t0);
    fp4_mul(&t0, &w->c, &y->c);                                      /* xc.yc */
    fp4_sub(&w->c, &z2, &t0);
    fp4_sub(&z3, &z3, &t0);                                          /* xb.yc + xc.yb */
    fp4_times_i(&t0); fp4_add(&w->b, &wb, &t0);
    fp4_times_i(&z3); fp4_add(&w->a, &z0, &z3);
}
/* FP12_sqr :190-238 (Chung-Hasan SQR2) */
static void fp12_sqr(fp12* w, const fp12* x) {
    fp4 A, B, C, D, wc;
    fp4_sqr(&A, &x->a);
    fp4_mul(&B, &x->b, &x->c); fp4_add(&B, &B, &B);
    fp4_sqr(&C, &x->c);
    fp4_mul(&D, &x->a, &x->b); fp4_add(&D, &D, &D);
    fp4_add(&wc, &x->a, &x->c); fp4_add(&wc, &x->b, &wc); fp4_sqr(&wc, &wc);
    fp4 wa = A;
    fp4_add(&A, &A, &B); fp4_add(&A, &A, &C); fp4_add(&A, &A, &D); fp4_neg(&A, &A);
    fp4_times_i(&B); fp4_times_i(&C);
    fp4_add(&w->a, &wa, &B); fp4_add(&w->b, &C, &D); fp4_add(&w->c, &wc, &A);
}
/* FP12_usqr :147-186 (Granger-Scott; equals sqr only on the cyclotomic subgroup) */
static void fp12_usqr(fp12* w, const fp12* x) {
    fp4 A, B, C, D, wa, wb, wc;
    A = x->a;
    fp4_sqr(&wa, &x->a); fp4_add(&D, &wa, &wa); fp4_add(&wa, &D, &wa);
    fp4_nconj(&A, &A); fp4_add(&A, &A, &A); fp4_add(&wa, &wa, &A);
    fp4_sqr(&B, &x->c); fp4_times_i(&B); fp4_add(&D, &B, &B); fp4_add(&B, &B, &D);
    fp4_sqr(&C, &x->b); fp4_add(&D, &C, &C); fp4_add(&C, &C, &D);
    fp4_conj(&wb, &x->b); fp4_add(&wb, &wb, &wb);
    fp4_nconj(&wc, &x->c); fp4_add(&wc, &wc, &wc);
    fp4_add(&wb, &B, &wb); fp4_add(&wc, &C, &wc);
    w->a = wa; w->b = wb; w->c = wc;
}
/* FP12_inv :627-664 */
static void fp12_inv(fp12* w, const fp12* x) {
    fp4 f0, f1, f2, f3, t;
    fp4_sqr(&f0, &x->a); fp4_mul(&f1, &x->b, &x->c); fp4_times_i(&f1); fp4_sub(&f0, &f0, &f1);
    fp4_sqr(&f1, &x->c); fp4_times_i(&f1); fp4_mul(&f2, &x->a, &x->b); fp4_sub(&f1, &f1, &f2);
    fp4_sqr(&f2, &x->b); fp4_mul(&f3, &x->a, &x->c); fp4_sub(&f2, &f2, &f3);
    fp4_mul(&f3, &x->b, &f2); fp4_times_i(&f3);
    fp4_mul(&t, &f0, &x->a); fp4_add(&f3, &t, &f3);
    fp4_mul(&t, &f1, &x->c); fp4_times_i(&t); fp4_add(&f3, &t, &f3);
    fp4_inv(&f3, &f3);
    fp4_mul(&w->a, &f0, &f3); fp4_mul(&w->b, &f1, &f3); fp4_mul(&w->c, &f2, &f3);
}
/* FP12_frob :867-880 */
static void fp12_frob(fp12* w, const fp2* f) {
    fp2 f2, f3;
    fp2_sqr(&f2, f); fp2_mul(&f3, &f2, f);
    fp4_frob(&w->a, &f3); fp4_frob(&w->b, &f3); fp4_frob(&w->c, &f3);
    fp4_pmul(&w->b, &w->b, f); fp4_pmul(&w->c, &w->c, &f2);
}
/* small helpers on little-endian multiword integers */
static int bn_bit(const uint64_t* a, int nw, int i) { return i / 64 < nw ? (int)((a[i / 64] >> (i % 64)) & 1) : 0; }
static int bn_nbits(const uint64_t* a, int nw) { for (int i = nw * 64 - 1; i >= 0; i--) if (bn_bit(a, nw, i)) return i + 1; return 0; }
/* FP12_pow :736-774: signed-digit (3e - e) ladder with unitary squarings; exponent used as given */
static void fp12_pow(fp12* r, const fp12* a, const uint64_t* e, int nw) {
    uint64_t e3[6] = {0}; u128 c = 0;
    for (int i = 0; i < nw; i++) { c += (u128)e[i] * 3; e3[i] = (uint64_t)c; c >>= 64; }
    e3[nw] = (uint64_t)c;
    int nb = bn_nbits(e3, nw + 1);
    if (nb == 0) { fp12_one(r); return; }
    fp12 w = *a, sf = *a, sfc; fp12_conj(&sfc, &sf);
    for (int i = nb - 2; i >= 1; i--) {
        fp12_usqr(&w, &w);
        int bt = bn_bit(e3, nw + 1, i) - bn_bit(e, nw, i);
        if (bt == 1) fp12_mul(&w, &sf);
        if (bt == -1) fp12_mul(&w, &sfc);
    }
    *r = w;
}
static void fp12_from_bytes(fp12* g, const uint8_t* b) { fp4_from_bytes(&g->c, b); fp4_from_bytes(&g->b, b + 192); fp4_from_bytes(&g->a, b + 384); }  /* :933-939 */
static void fp12_to_bytes(uint8_t* b, const fp12* g) { fp4_to_bytes(b, &g->c); fp4_to_bytes(b + 192, &g->b); fp4_to_bytes(b + 384, &g->a); }          /* :923-929 */

/* ------------------------------------------------------------------ G1 (ecp_BLS12381.cpp) */
static fp B3_G1;     /* 3*b = 12 */
static void g1_inf(g1p* P) { memset(P, 0, sizeof *P); P->y = ONE; }                                   /* ECP_inf :149 */
static int g1_is_inf(const g1p* P) { return fp_is_zero(&P->x) && fp_is_zero(&P->z); }                 /* ECP_isinf :31 */
/* ECP_dbl :550-588 (a = 0, Renes-Costello-Batina complete doubling) */
static void g1_dbl(g1p* P) {
    fp t0, t1, t2, x3, y3, z3;
    fp_sqr(&t0, &P->y); fp_mul(&t1, &P->y, &P->z); fp_sqr(&t2, &P->z);
    fp_dbl(&z3, &t0); fp_dbl(&z3, &z3); fp_dbl(&z3, &z3);             /* 8 y^2 */
    fp_mul(&t2, &t2, &B3_G1);                                          /* 3b z^2 */
    fp_mul(&x3, &t2, &z3);
    fp_add(&y3, &t0, &t2);
    fp_mul(&z3, &z3, &t1);
    fp_dbl(&t1, &t2); fp_add(&t2, &t2, &t1);
    fp_sub(&t0, &t0, &t2);
    fp_mul(&y3, &y3, &t0); fp_add(&y3, &y3, &x3);
    fp_mul(&t1, &P->x, &P->y);
    fp_mul(&x3, &t0, &t1); fp_dbl(&x3, &x3);
    P->x = x3; P->y = y3; P->z = z3;
}
/* ECP_add :750-812 (a = 0, complete addition) */
static void g1_add(g1p* P, const g1p* Q) {
    fp t0, t1, t2, t3, t4, x3, y3, z3;
    fp_mul(&t0, &P->x, &Q->x); fp_mul(&t1, &P->y, &Q->y); fp_mul(&t2, &P->z, &Q->z);
    fp_add(&t3, &P->x, &P->y); fp_add(&t4, &Q->x, &Q->y); fp_mul(&t3, &t3, &t4);
    fp_add(&t4, &t0, &t1); fp_sub(&t3, &t3, &t4);
    fp_add(&t4, &P->y, &P->z); fp_add(&x3, &Q->y, &Q->z); fp_mul(&t4, &t4, &x3);
    fp_add(&x3, &t1, &t2); fp_sub(&t4, &t4, &x3);
    fp_add(&x3, &P->x, &P->z); fp_add(&y3, &Q->x, &Q->z); fp_mul(&x3, &x3, &y3);
    fp_add(&y3, &t0, &t2); fp_sub(&y3, &x3, &y3);
    fp_dbl(&x3, &t0); fp_add(&t0, &t0, &x3);
    fp_mul(&t2, &t2, &B3_G1);
    fp_add(&z3, &t1, &t2); fp_sub(&t1, &t1, &t2);
    fp_mul(&y3, &y3, &B3_G1);
    fp_mul(&x3, &y3, &t4); fp_mul(&t2, &t3, &t1); fp_sub(&P->x, &t2, &x3);
    fp_mul(&y3, &y3, &t0); fp_mul(&t1, &t1, &z3); fp_add(&P->y, &y3, &t1);
    fp_mul(&t0, &t0, &t3); fp_mul(&z3, &z3, &t4); fp_add(&P->z, &z3, &t0);
}
/* ECP_affine :329-350 */
static void g1_affine(g1p* P) {
    if (g1_is_inf(P)) return;
    fp zi; fp_inv(&zi, &P->z);
    fp_mul(&P->x, &P->x, &zi); fp_mul(&P->y, &P->y, &zi); P->z = ONE;
}
/* ECP_rhs :279: x^3 + 4 */
static void g1_rhs(fp* r, const fp* x) { fp t, four; fp_sqr(&t, x); fp_mul(&t, &t, x); fp_set_int(&four, 4); fp_add(r, &t, &four); }
/* ECP_set :232 — accept (x,y) only if on the curve */
static int g1_set(g1p* P, const fp* x, const fp* y) {
    fp rhs, y2; g1_rhs(&rhs, x); fp_sqr(&y2, y);
    if (!fp_eq(&y2, &rhs)) { g1_inf(P); return 0; }
    P->x = *x; P->y = *y; P->z = ONE; return 1;
}
/* ECP_setx :302-323 — y from x with the requested sign (parity) */
static int g1_setx(g1p* P, const fp* x, int s) {
    fp rhs, y; g1_rhs(&rhs, x);
    if (!fp_qr(&rhs)) { g1_inf(P); return 0; }
    fp_sqrt(&y, &rhs);
    if (fp_sign(&y) != s) fp_neg(&y, &y);
    P->x = *x; P->y = y; P->z = ONE; return 1;
}
static int all_zero(const uint8_t* p, size_t n) { uint8_t t = 0; for (size_t i = 0; i < n; i++) t |= p[i]; return t == 0; }
static int g1_load96(g1p* P, const uint8_t* b) {
    if (all_zero(b, 96)) { g1_inf(P); return 1; }
    fp x, y; fp_from_bytes(&x, b); fp_from_bytes(&y, b + 48); return g1_set(P, &x, &y);
}
/* ECP_toOctet :445-488 (non-ALT branch) */
static void g1_store(uint8_t* out, g1p* P, int fmt) {
    if (g1_is_inf(P)) { memset(out, 0, (size_t)fmt); return; }
    g1_affine(P);
    if (fmt == 49) { out[0] = (uint8_t)(0x02 | fp_sign(&P->y)); fp_to_bytes(out + 1, &P->x); }
    else { fp_to_bytes(out, &P->x); fp_to_bytes(out + 48, &P->y); }
}
/* scalar: 32 big-endian bytes -> k mod r   (PAIR_G1mul pair_BLS12381.cpp:879-881 reduces first) */
static void scalar_load(uint64_t k[4], const uint8_t* s) {
    for (int i = 0; i < 4; i++) { uint64_t w = 0; for (int j = 0; j < 8; j++) w = (w << 8) | s[(3 - i) * 8 + j]; k[i] = w; }
    for (;;) {
        int ge = 1;
        for (int i = 3; i >= 0; i--) { if (k[i] > RORD[i]) break; if (k[i] < RORD[i]) { ge = 0; break; } }
        if (!ge) break;
        u128 bw = 0;
        for (int i = 0; i < 4; i++) { u128 d = (u128)k[i] - RORD[i] - bw; k[i] = (uint64_t)d; bw = (d >> 64) & 1; }
    }
}
/* plain left-to-right double-and-add on the complete formulas (the exact group operation) */
static void g1_mul_plain(g1p* P, const uint64_t* k, int nw) {
    g1p acc; g1_inf(&acc);
    for (int i = bn_nbits(k, nw) - 1; i >= 0; i--) { g1_dbl(&acc); if (bn_bit(k, nw, i)) g1_add(&acc, P); }
    *P = acc;
}
/* q = a div d, rem = a mod d for a < 2^256 (4 words) and d < 2^128 (2 words): restoring division */
static void bn_divmod_128(uint64_t q[4], uint64_t rem[2], const uint64_t a[4], const uint64_t d[2]) {
    uint64_t r[3] = {0, 0, 0};
    memset(q, 0, 32);
    for (int i = 255; i >= 0; i--) {
        r[2] = (r[2] << 1) | (r[1] >> 63); r[1] = (r[1] << 1) | (r[0] >> 63); r[0] = (r[0] << 1) | (uint64_t)bn_bit(a, 4, i);
        int ge = r[2] || r[1] > d[1] || (r[1] == d[1] && r[0] >= d[0]);
        if (ge) { u128 t = (u128)r[0] - d[0]; r[0] = (uint64_t)t; uint64_t bw = (uint64_t)(t >> 64) & 1; t = (u128)r[1] - d[1] - bw; r[1] = (uint64_t)t; bw = (uint64_t)(t >> 64) & 1; r[2] -= bw; q[i / 64] |= 1ULL << (i % 64); }
    }
    rem[0] = r[0]; rem[1] = r[1];
}
static fp BETA;      /* CRu (rom_field_BLS12381.cpp:54): (beta x, y) = [-x^2](x, y) on G1; fixed by a self-test at init */
/* PAIR_G1mul pair_BLS12381.cpp:876-924 with glv() :793-805: u0 = k mod x^2, u1 = r - (k div x^2); the sign
 * minimisation (:896-914) then replaces (u1, phi(P)) by (k div x^2, -phi(P)), so what ECP_mul2 evaluates — with
 * complete formulas, i.e. as exact group operations on ANY curve point, in or out of the order-r subgroup — is
 *     R = [k mod x^2] P + [k div x^2] (-phi(P)),   phi(x, y) = (beta x, y),
 * and R = [k] P + [r] phi(P) when k div x^2 = 0 (see below).  (For P in G1 this is [k]P.)  The joint window schedule of ECP_clmul2 is an evaluation strategy. */
static void g1_mul(g1p* P, const uint64_t k[4]) {
    if (g1_is_inf(P)) return;
    u128 x2 = (u128)BNX * BNX;
    uint64_t d[2] = {(uint64_t)x2, (uint64_t)(x2 >> 64)}, q[4], u0[2];
    bn_divmod_128(q, u0, k, d);
    g1p A = *P, Q = *P;
    g1_affine(&Q);
    fp_mul(&Q.x, &Q.x, &BETA);
    g1_mul_plain(&A, u0, 2);
    if ((q[0] | q[1]) == 0) {
        /* k < x^2: u1 = r - 0 = r survives the sign minimisation (BIG_modneg(r) = r, no fewer bits), so ECP_mul2 adds
         * [r] phi(P) — infinity for P in G1, a cofactor-part point otherwise */
        g1_mul_plain(&Q, RORD, 4);
    } else {
        fp_neg(&Q.y, &Q.y);
        g1_mul_plain(&Q, q, 2);
    }
    g1_add(&A, &Q);
    *P = A;
}

/* ------------------------------------------------------------------ G2 (ecp2_BLS12381.cpp) */
static void g2_inf(g2p* P) { memset(P, 0, sizeof *P); P->y.a = ONE; }
static int g2_is_inf(const g2p* P) { return fp2_is_zero(&P->x) && fp2_is_zero(&P->z); }
static void fp2_mul_b3(fp2* r, const fp2* a) { fp2_imul(r, a, 12); fp2_mul_ip(r); }   /* 3b(1+i): imul then mul_ip, ecp2:381-385 */
/* ECP2_dbl :358-409 (M-type twist) */
static void g2_dbl(g2p* P) {
    fp2 t0, t1, t2, x3, y3, z3;
    fp2_sqr(&t0, &P->y); fp2_mul(&t1, &P->y, &P->z); fp2_sqr(&t2, &P->z);
    fp2_add(&z3, &t0, &t0); fp2_add(&z3, &z3, &z3); fp2_add(&z3, &z3, &z3);
    fp2_mul_b3(&t2, &t2);
    fp2_mul(&x3, &t2, &z3);
    fp2_add(&y3, &t0, &t2);
    fp2_mul(&z3, &z3, &t1);
    fp2_add(&t1, &t2, &t2); fp2_add(&t2, &t2, &t1);
    fp2_sub(&t0, &t0, &t2);
    fp2_mul(&y3, &y3, &t0); fp2_add(&y3, &y3, &x3);
    fp2_mul(&t1, &P->x, &P->y);
    fp2_mul(&x3, &t0, &t1); fp2_add(&x3, &x3, &x3);
    P->x = x3; P->y = y3; P->z = z3;
}
/* ECP2_add :413-502 */
static void g2_add(g2p* P, const g2p* Q) {
    fp2 t0, t1, t2, t3, t4, x3, y3, z3;
    fp2_mul(&t0, &P->x, &Q->x); fp2_mul(&t1, &P->y, &Q->y); fp2_mul(&t2, &P->z, &Q->z);
    fp2_add(&t3, &P->x, &P->y); fp2_add(&t4, &Q->x, &Q->y); fp2_mul(&t3, &t3, &t4);
    fp2_add(&t4, &t0, &t1); fp2_sub(&t3, &t3, &t4);
    fp2_add(&t4, &P->y, &P->z); fp2_add(&x3, &Q->y, &Q->z); fp2_mul(&t4, &t4, &x3);
    fp2_add(&x3, &t1, &t2); fp2_sub(&t4, &t4, &x3);
    fp2_add(&x3, &P->x, &P->z); fp2_add(&y3, &Q->x, &Q->z); fp2_mul(&x3, &x3, &y3);
    fp2_add(&y3, &t0, &t2); fp2_sub(&y3, &x3, &y3);
    fp2_add(&x3, &t0, &t0); fp2_add(&t0, &t0, &x3);
    fp2_mul_b3(&t2, &t2);
    fp2_add(&z3, &t1, &t2); fp2_sub(&t1, &t1, &t2);
    fp2_mul_b3(&y3, &y3);
    fp2_mul(&x3, &y3, &t4); fp2_mul(&t2, &t3, &t1); fp2_sub(&P->x, &t2, &x3);
    fp2_mul(&y3, &y3, &t0); fp2_mul(&t1, &t1, &z3); fp2_add(&P->y, &y3, &t1);
    fp2_mul(&t0, &t0, &t3); fp2_mul(&z3, &z3, &t4); fp2_add(&P->z, &z3, &t0);
}
static void g2_neg(g2p* P) { fp2_neg(&P->y, &P->y); }                                   /* ECP2_neg :348 */
static void fp2_one(fp2* r) { memset(r, 0, sizeof *r); r->a = ONE; }
/* ECP2_affine :109-133 */
static void g2_affine(g2p* P) {
    if (g2_is_inf(P)) return;
    fp2 zi; fp2_inv(&zi, &P->z);
    fp2_mul(&P->x, &P->x, &zi); fp2_mul(&P->y, &P->y, &zi); fp2_one(&P->z);
}
/* ECP2_rhs :270-296: x^3 + 4(1+i) on the M-type twist */
static void g2_rhs(fp2* r, const fp2* x) {
    fp2 t, b; fp2_sqr(&t, x); fp2_mul(&t, &t, x);
    memset(&b, 0, sizeof b); fp_set_int(&b.a, 4); fp2_mul_ip(&b);
    fp2_add(r, &t, &b);
}
static int g2_set(g2p* P, const fp2* x, const fp2* y) {                                  /* ECP2_set :299 */
    fp2 rhs, y2; g2_rhs(&rhs, x); fp2_sqr(&y2, y);
    if (!fp2_eq(&y2, &rhs)) { g2_inf(P); return 0; }
    P->x = *x; P->y = *y; fp2_one(&P->z); return 1;
}
static int g2_setx(g2p* P, const fp2* x, int s) {                                        /* ECP2_setx :322-344 */
    fp2 rhs, y; g2_rhs(&rhs, x);
    if (!fp2_qr(&rhs)) { g2_inf(P); return 0; }
    fp2_sqrt(&y, &rhs);
    if (fp2_sign(&y) != s) fp2_neg(&y, &y);
    P->x = *x; P->y = y; fp2_one(&P->z); return 1;
}
static int g2_load192(g2p* P, const uint8_t* b) {
    if (all_zero(b, 192)) { g2_inf(P); return 1; }
    fp2 x, y; fp2_from_bytes(&x, b); fp2_from_bytes(&y, b + 96); return g2_set(P, &x, &y);
}
static void g2_store(uint8_t* out, g2p* P, int fmt) {                                    /* ECP2_toOctet :184-220 */
    if (g2_is_inf(P)) { memset(out, 0, (size_t)fmt); return; }
    g2_affine(P);
    if (fmt == 97) { out[0] = (uint8_t)(0x02 | fp2_sign(&P->y)); fp2_to_bytes(out + 1, &P->x); }
    else { fp2_to_bytes(out, &P->x); fp2_to_bytes(out + 96, &P->y); }
}
static void g2_mul_plain(g2p* P, const uint64_t* k, int nw) {
    g2p acc; g2_inf(&acc);
    for (int i = bn_nbits(k, nw) - 1; i >= 0; i--) { g2_dbl(&acc); if (bn_bit(k, nw, i)) g2_add(&acc, P); }
    *P = acc;
}
static fp2 PSI_X, PSI_Y;   /* g^2, g^3 with g = 1/(Fra + i Frb): ECP2_frob ecp2_BLS12381.cpp:579-590 with X inverted (M-type, pair:944-947) */
/* ECP2_frob :579-590 */
static void g2_frob(g2p* P) {
    fp2_conj(&P->x, &P->x); fp2_conj(&P->y, &P->y); fp2_conj(&P->z, &P->z);
    fp2_mul(&P->x, &PSI_X, &P->x); fp2_mul(&P->y, &PSI_Y, &P->y);
}
/* PAIR_G2mul pair_BLS12381.cpp:927-983 with gs() :814-873 (BLS branch): k mod r is written in base |x|,
 * k = u0 + u1|x| + u2|x|^2 + u3|x|^3; x < 0 makes the odd digits negative (:868-871) and the sign minimisation
 * (:962-971) turns that into negated points, so ECP2_mul4 evaluates — exactly, on any point of the twist —
 *     R = u0 Q - u1 psi(Q) + u2 psi^2(Q) - u3 psi^3(Q),
 * with -u_i psi^i(Q) replaced by +[r] psi^i(Q) when an odd digit u_i is 0.  (For Q in G2, psi(Q) = [x]Q and this is [k]Q.) */
static void g2_mul(g2p* P, const uint64_t k[4]) {
    if (g2_is_inf(P)) return;
    uint64_t w[4], u[4];
    memcpy(w, k, 32);
    for (int i = 0; i < 3; i++) {
        u128 rem = 0; uint64_t q[4];
        for (int j = 3; j >= 0; j--) { u128 cur = (rem << 64) | w[j]; q[j] = (uint64_t)(cur / BNX); rem = cur % BNX; }
        u[i] = (uint64_t)rem; memcpy(w, q, 32);
    }
    u[3] = w[0];
    g2p acc, Q = *P; g2_inf(&acc);
    for (int i = 0; i < 4; i++) {
        g2p T = Q;
        if ((i & 1) && u[i] == 0) {
            /* BIG_modneg(0) = r (:868-871) and the sign minimisation keeps it: the term is [r] psi^i(Q), not negated —
             * infinity for Q in G2, a cofactor-part point otherwise */
            g2_mul_plain(&T, RORD, 4);
        } else {
            if (i & 1) g2_neg(&T);
            uint64_t e[1] = {u[i]};
            g2_mul_plain(&T, e, 1);
        }
        g2_add(&acc, &T);
        g2_frob(&Q);
    }
    *P = acc;
}

/* ------------------------------------------------------------------ pairing (pair_BLS12381.cpp) */
/* PAIR_double :40-78 */
static void pair_double(g2p* A, fp2* AA, fp2* BB, fp2* CC) {
    fp2 YY;
    *CC = A->x; YY = A->y; *BB = A->z;
    fp2_mul(AA, &YY, BB);
    fp2_sqr(CC, CC); fp2_sqr(&YY, &YY); fp2_sqr(BB, BB);
    fp2_add(AA, AA, AA); fp2_neg(AA, AA); fp2_mul_ip(AA);              /* -2YZ(1+i) */
    fp2_imul(BB, BB, 12); fp2_imul(CC, CC, 3);
    fp2_mul_ip(BB);                                                     /* M-type */
    fp2_sub(BB, BB, &YY);
    g2_dbl(A);
}
/* PAIR_add :81-116 (B affine) */
static void pair_add(g2p* A, const g2p* B, fp2* AA, fp2* BB, fp2* CC) {
    fp2 T1;
    *AA = A->x; *CC = A->y; T1 = A->z; *BB = T1;
    fp2_mul(&T1, &T1, &B->y); fp2_mul(BB, BB, &B->x);
    fp2_sub(AA, AA, BB); fp2_sub(CC, CC, &T1);
    T1 = *AA;
    fp2_mul_ip(AA);                                                     /* M-type */
    fp2_mul(&T1, &T1, &B->y);
    *BB = *CC; fp2_mul(BB, BB, &B->x); fp2_sub(BB, BB, &T1);
    fp2_neg(CC, CC);
    g2_add(A, B);
}
/* PAIR_line :119-144: sparse Fp12 a=[AA*Qy, BB], b=0, c=[0, CC*Qx] (M-type), here stored densely */
static void pair_line(fp12* v, g2p* A, const g2p* B, const fp* Qx, const fp* Qy) {
    fp2 AA, BB, CC;
    if (B == NULL) pair_double(A, &AA, &BB, &CC); else pair_add(A, B, &AA, &BB, &CC);
    fp2_pmul(&CC, &CC, Qx); fp2_pmul(&AA, &AA, Qy);
    memset(v, 0, sizeof *v);
    v->a.a = AA; v->a.b = BB; v->c.b = CC;
}
/* PAIR_ate :425-505 (BLS12 branch): n = |x|, n3 = 3n, signed digit n3_i - n_i */
static void pair_ate(fp12* r, const g2p* P1, const g1p* Q1) {
    fp12_one(r);
    if (g1_is_inf(Q1)) return;
    g2p P = *P1, A, NP; g1p Q = *Q1;
    g2_affine(&P); g1_affine(&Q);
    A = P; NP = P; g2_neg(&NP);
    uint64_t n[2] = {BNX, 0}, n3[2]; u128 t = (u128)BNX * 3; n3[0] = (uint64_t)t; n3[1] = (uint64_t)(t >> 64);
    int nb = bn_nbits(n3, 2);
    fp12 lv, lv2;
    for (int i = nb - 2; i >= 1; i--) {
        fp12_sqr(r, r);
        pair_line(&lv, &A, NULL, &Q.x, &Q.y);
        int bt = bn_bit(n3, 2, i) - bn_bit(n, 2, i);
        if (bt == 1) { pair_line(&lv2, &A, &P, &Q.x, &Q.y); fp12_mul(&lv, &lv2); }     /* FP12_smul :497 */
        if (bt == -1) { pair_line(&lv2, &A, &NP, &Q.x, &Q.y); fp12_mul(&lv, &lv2); }
        fp12_mul(r, &lv);                                                              /* FP12_ssmul :304 */
    }
    fp12_conj(r, r);                                                                   /* NEGATIVEX :485-487 */
}
/* PAIR_fexp :629-755, BLS12 branch :711-753 (eprint 2020/875) */
static void fp12_pow_x(fp12* r, const fp12* a) { uint64_t e[1] = {BNX}; fp12_pow(r, a, e, 1); fp12_conj(r, r); }
static void pair_fexp(fp12* r) {
    fp12 t0, y0, y1;
    fp12_inv(&t0, r); fp12_conj(r, r); fp12_mul(r, &t0); t0 = *r;
    fp12_frob(r, &FROB); fp12_frob(r, &FROB); fp12_mul(r, &t0);
    fp12_usqr(&y1, r); fp12_mul(&y1, r);                    /* r^3 */
    fp12_pow_x(&y0, r); fp12_conj(&t0, r); *r = y0; fp12_mul(r, &t0);     /* r^(x-1) */
    fp12_pow_x(&y0, r); fp12_conj(&t0, r); *r = y0; fp12_mul(r, &t0);     /* r^(x-1) */
    fp12_pow_x(&y0, r); t0 = *r; fp12_frob(&t0, &FROB); *r = y0; fp12_mul(r, &t0);   /* ^(x+p) */
    fp12_pow_x(&y0, r); fp12_pow_x(&y0, &y0);
    t0 = *r; fp12_frob(&t0, &FROB); fp12_frob(&t0, &FROB);
    fp12_mul(&y0, &t0); fp12_conj(&t0, r); *r = y0; fp12_mul(r, &t0);                /* ^(x^2+p^2-1) */
    fp12_mul(r, &y1);
}

/* ------------------------------------------------------------------ start-up constants */
static const char* G1X_HEX = "17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb";
static const char* G1Y_HEX = "08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1";
static const char* G2XA_HEX = "024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8";
static const char* G2XB_HEX = "13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e";
static const char* G2YA_HEX = "0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801";
static const char* G2YB_HEX = "0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be";
static void hex48(uint8_t* out, const char* h) {
    for (int i = 0; i < 48; i++) {
        int v = 0;
        for (int j = 0; j < 2; j++) { char c = h[2 * i + j]; v = v * 16 + (c <= '9' ? c - '0' : c - 'a' + 10); }
        out[i] = (uint8_t)v;
    }
}
static void bn_div_small(uint64_t* q, const uint64_t* a, int nw, uint64_t d) {
    u128 rem = 0;
    for (int i = nw - 1; i >= 0; i--) { u128 cur = (rem << 64) | a[i]; q[i] = (uint64_t)(cur / d); rem = cur % d; }
}
static pthread_once_t once = PTHREAD_ONCE_INIT;
static void init_consts(void) {
    uint64_t pm1[6], pp1[6];
    memcpy(E_PM2, P, 48); E_PM2[0] -= 2;
    memcpy(pm1, P, 48); pm1[0] -= 1;
    memcpy(pp1, P, 48); pp1[0] += 1;
    bn_div_small(E_PP1D4, pp1, 6, 4);
    bn_div_small(E_PM1D6, pm1, 6, 6);
    bn_div_small(E_PM1D2, pm1, 6, 2);
    fp_set_int(&B3_G1, 12);
    /* Frobenius constant (1+i)^((p-1)/6), by square-and-multiply in Fp2 */
    fp2 base, acc; memset(&base, 0, sizeof base); base.a = ONE; base.b = ONE; fp2_one(&acc);
    for (int i = 0; i < 384; i++) { if (bn_bit(E_PM1D6, 6, i)) fp2_mul(&acc, &acc, &base); fp2_sqr(&base, &base); }
    FROB = acc;
    /* psi constants: g = 1/f, g^2, g^3 */
    fp2 g; fp2_inv(&g, &FROB);
    fp2_sqr(&PSI_X, &g); fp2_mul(&PSI_Y, &PSI_X, &g);
    /* beta: the primitive cube root of unity with (beta Gx, Gy) = [-x^2 mod r] G on the generator */
    uint64_t pm1d3[6]; bn_div_small(pm1d3, pm1, 6, 3);
    fp cand, gsmall, b1, b2;
    for (int gi = 2;; gi++) { fp_set_int(&gsmall, gi); fp_pow(&cand, &gsmall, pm1d3, 6); if (!fp_eq(&cand, &ONE)) break; }
    b1 = cand; fp_sqr(&b2, &cand);
    uint8_t gb[96]; hex48(gb, G1X_HEX); hex48(gb + 48, G1Y_HEX);
    g1p G, T; fp gx, gy; fp_from_bytes(&gx, gb); fp_from_bytes(&gy, gb + 48); g1_set(&G, &gx, &gy);
    /* lam = r - x^2 */
    u128 x2 = (u128)BNX * BNX; uint64_t lam[4]; u128 bw = 0;
    uint64_t x2w[4] = {(uint64_t)x2, (uint64_t)(x2 >> 64), 0, 0};
    for (int i = 0; i < 4; i++) { u128 d = (u128)RORD[i] - x2w[i] - (uint64_t)bw; lam[i] = (uint64_t)d; bw = (d >> 64) & 1; }
    T = G; g1_mul_plain(&T, lam, 4); g1_affine(&T);
    fp t1; fp_mul(&t1, &gx, &b1);
    BETA = fp_eq(&t1, &T.x) ? b1 : b2;
}
#define INIT() pthread_once(&once, init_consts)

/* ------------------------------------------------------------------ threading helper */
typedef struct { void (*fn)(size_t, size_t, void*); void* ctx; size_t lo, hi; } job_t;
static void* job_run(void* p) { job_t* j = (job_t*)p; j->fn(j->lo, j->hi, j->ctx); return NULL; }
static void par_for(size_t n, int nthreads, void (*fn)(size_t, size_t, void*), void* ctx) {
    if (nthreads <= 1 || n < 2) { fn(0, n, ctx); return; }
    size_t T = (size_t)nthreads < n ? (size_t)nthreads : n;
    pthread_t* th = (pthread_t*)malloc(T * sizeof *th); job_t* jb = (job_t*)malloc(T * sizeof *jb);
    for (size_t t = 0; t < T; t++) { jb[t].fn = fn; jb[t].ctx = ctx; jb[t].lo = n * t / T; jb[t].hi = n * (t + 1) / T; pthread_create(&th[t], NULL, job_run, &jb[t]); }
    for (size_t t = 0; t < T; t++) pthread_join(th[t], NULL);
    free(th); free(jb);
}

/* ------------------------------------------------------------------ exported batch entry points */
int orc_g1_generator(uint8_t out[96]) { INIT(); hex48(out, G1X_HEX); hex48(out + 48, G1Y_HEX); g1p P; return g1_load96(&P, out) ? 0 : -1; }
int orc_g2_generator(uint8_t out[192]) {
    INIT(); hex48(out, G2XB_HEX); hex48(out + 48, G2XA_HEX); hex48(out + 96, G2YB_HEX); hex48(out + 144, G2YA_HEX);
    g2p P; return g2_load192(&P, out) ? 0 : -1;
}

int orc_fp_op_batch(int op, size_t n, const uint8_t* a, const uint8_t* b, uint8_t* out, uint8_t* ok) {
    INIT();
    for (size_t i = 0; i < n; i++) {
        fp x, y, r; int st = 1;
        fp_from_bytes(&x, a + 48 * i);
        if (b) fp_from_bytes(&y, b + 48 * i);
        switch (op) {
            case 0: fp_mul(&r, &x, &y); break;
            case 1: fp_add(&r, &x, &y); break;
            case 2: fp_sub(&r, &x, &y); break;
            case 3: fp_sqr(&r, &x); break;
            case 4: fp_neg(&r, &x); break;
            case 5: fp_inv(&r, &x); break;
            case 6: st = fp_qr(&x); if (st) fp_sqrt(&r, &x); else memset(&r, 0, sizeof r); break;
            default: return -1;
        }
        fp_to_bytes(out + 48 * i, &r);
        if (ok) ok[i] = (uint8_t)st;
    }
    return 0;
}

typedef struct { const uint8_t *p, *s; uint8_t* out; int fmt; int bad; } mul_ctx;
static void g1_mul_range(size_t lo, size_t hi, void* c) {
    mul_ctx* m = (mul_ctx*)c;
    for (size_t i = lo; i < hi; i++) {
        g1p P; uint64_t k[4];
        if (!g1_load96(&P, m->p + 96 * i)) { m->bad = 1; memset(m->out + (size_t)m->fmt * i, 0xff, (size_t)m->fmt); continue; }
        scalar_load(k, m->s + 32 * i);
        g1_mul(&P, k);
        g1_store(m->out + (size_t)m->fmt * i, &P, m->fmt);
    }
}
int orc_g1_mul_batch(size_t n, const uint8_t* pts96, const uint8_t* scalars32, uint8_t* out, int out_fmt, int nthreads) {
    INIT(); if (out_fmt != 49 && out_fmt != 96) return -1;
    mul_ctx m = {pts96, scalars32, out, out_fmt, 0};
    par_for(n, nthreads, g1_mul_range, &m);
    return m.bad ? -2 : 0;
}
int orc_g1_add_batch(size_t n, const uint8_t* a96, const uint8_t* b96, uint8_t* out, int out_fmt) {
    INIT();
    for (size_t i = 0; i < n; i++) {
        g1p A, B;
        if (!g1_load96(&A, a96 + 96 * i) || !g1_load96(&B, b96 + 96 * i)) return -2;
        g1_add(&A, &B);
        g1_store(out + (size_t)out_fmt * i, &A, out_fmt);
    }
    return 0;
}
/* ECP_fromOctet :495-545 for 49-byte input; a leading 0x00 is infinity (g1_point.hpp:89-93) */
int orc_g1_decompress_batch(size_t n, const uint8_t* in49, uint8_t* out96, uint8_t* status) {
    INIT();
    for (size_t i = 0; i < n; i++) {
        const uint8_t* s = in49 + 49 * i; g1p P; int ok = 0;
        if (s[0] == 0) { memset(out96 + 96 * i, 0, 96); status[i] = 1; continue; }
        if (s[0] == 0x02 || s[0] == 0x03) { fp x; fp_from_bytes(&x, s + 1); ok = g1_setx(&P, &x, s[0] & 1); }
        /* tag 0x04 needs 97 bytes: with a 49-byte view the reference reads y from beyond the buffer; not exercised */
        status[i] = (uint8_t)ok;
        if (ok) g1_store(out96 + 96 * i, &P, 96); else memset(out96 + 96 * i, 0, 96);
    }
    return 0;
}
int orc_g1_compress_batch(size_t n, const uint8_t* in96, uint8_t* out49) {
    INIT();
    for (size_t i = 0; i < n; i++) { g1p P; if (!g1_load96(&P, in96 + 96 * i)) return -2; g1_store(out49 + 49 * i, &P, 49); }
    return 0;
}
typedef struct { const uint8_t *p, *s; g1p* part; size_t n; int T; int bad; } msm_ctx;
static void g1_msm_range(size_t lo, size_t hi, void* c) {
    msm_ctx* m = (msm_ctx*)c;
    /* identify the shard by its lower bound */
    size_t t = 0; while (t + 1 < (size_t)m->T && m->n * (t + 1) / (size_t)m->T <= lo) t++;
    g1p acc; g1_inf(&acc);
    for (size_t i = lo; i < hi; i++) {
        g1p P; uint64_t k[4];
        if (!g1_load96(&P, m->p + 96 * i)) { m->bad = 1; continue; }
        scalar_load(k, m->s + 32 * i); g1_mul(&P, k); g1_add(&acc, &P);
    }
    m->part[t] = acc;
}
/* Π g_i^{x_i} (include/crypto12381/g1_point.hpp:371-404 of the reference): only the final point is canonical */
int orc_g1_msm(size_t n, const uint8_t* pts96, const uint8_t* scalars32, uint8_t* out, int out_fmt, int nthreads) {
    INIT();
    int T = nthreads < 1 ? 1 : nthreads; if ((size_t)T > n && n > 0) T = (int)n; if (n < 2) T = 1;
    g1p* part = (g1p*)malloc((size_t)T * sizeof *part);
    for (int t = 0; t < T; t++) g1_inf(&part[t]);
    msm_ctx m = {pts96, scalars32, part, n, T, 0};
    par_for(n, T, g1_msm_range, &m);
    for (int t = 1; t < T; t++) g1_add(&part[0], &part[t]);
    g1_store(out, &part[0], out_fmt);
    free(part);
    return m.bad ? -2 : 0;
}

/* sum_of_products(point1&, n, point1*, const big*) src/miracl_core_interface.cpp:134-137 -> ECP_muln ecp_BLS12381.cpp:1112-1148: a plain
 * Pippenger over the scalars as given — the true multiples sum [k_i]P_i for ANY curve points (no endomorphism), unlike multiply().
 * Scalars are reduced mod r first, as the header layer's Zp values are (ref_wrap.cpp does the same before ECP_muln). */
int orc_g1_sum_of_products(int n, const uint8_t* pts96, const uint8_t* scalars32, uint8_t* out, int out_fmt) {
    INIT();
    g1p acc; g1_inf(&acc);
    for (int i = 0; i < n; i++) {
        g1p P; uint64_t k[4];
        if (!g1_load96(&P, pts96 + 96 * (size_t)i)) return -2;
        scalar_load(k, scalars32 + 32 * (size_t)i);
        g1_mul_plain(&P, k, 4);
        g1_add(&acc, &P);
    }
    g1_store(out, &acc, out_fmt);
    return 0;
}

static void g2_mul_range(size_t lo, size_t hi, void* c) {
    mul_ctx* m = (mul_ctx*)c;
    for (size_t i = lo; i < hi; i++) {
        g2p P; uint64_t k[4];
        if (!g2_load192(&P, m->p + 192 * i)) { m->bad = 1; memset(m->out + (size_t)m->fmt * i, 0xff, (size_t)m->fmt); continue; }
        scalar_load(k, m->s + 32 * i);
        g2_mul(&P, k);
        g2_store(m->out + (size_t)m->fmt * i, &P, m->fmt);
    }
}
int orc_g2_mul_batch(size_t n, const uint8_t* pts192, const uint8_t* scalars32, uint8_t* out, int out_fmt, int nthreads) {
    INIT(); if (out_fmt != 97 && out_fmt != 192) return -1;
    mul_ctx m = {pts192, scalars32, out, out_fmt, 0};
    par_for(n, nthreads, g2_mul_range, &m);
    return m.bad ? -2 : 0;
}
int orc_g2_add_batch(size_t n, const uint8_t* a192, const uint8_t* b192, uint8_t* out, int out_fmt) {
    INIT();
    for (size_t i = 0; i < n; i++) {
        g2p A, B;
        if (!g2_load192(&A, a192 + 192 * i) || !g2_load192(&B, b192 + 192 * i)) return -2;
        g2_add(&A, &B);
        g2_store(out + (size_t)out_fmt * i, &A, out_fmt);
    }
    return 0;
}
/* ECP2_fromOctet :225-266: any tag other than 0x04 is "compressed, sign = tag & 1"; 0x00 is infinity (g2_point.hpp:73-77) */
int orc_g2_decompress_batch(size_t n, const uint8_t* in97, uint8_t* out192, uint8_t* status) {
    INIT();
    for (size_t i = 0; i < n; i++) {
        const uint8_t* s = in97 + 97 * i; g2p P; int ok = 0;
        if (s[0] == 0) { memset(out192 + 192 * i, 0, 192); status[i] = 1; continue; }
        if (s[0] != 0x04) { fp2 x; fp2_from_bytes(&x, s + 1); ok = g2_setx(&P, &x, s[0] & 1); }
        status[i] = (uint8_t)ok;
        if (ok) g2_store(out192 + 192 * i, &P, 192); else memset(out192 + 192 * i, 0, 192);
    }
    return 0;
}
int orc_g2_compress_batch(size_t n, const uint8_t* in192, uint8_t* out97) {
    INIT();
    for (size_t i = 0; i < n; i++) { g2p P; if (!g2_load192(&P, in192 + 192 * i)) return -2; g2_store(out97 + 97 * i, &P, 97); }
    return 0;
}

typedef struct { const uint8_t *a1, *a2, *b1, *b2; uint8_t* out; int bad; } pair_ctx;
static void pair_range(size_t lo, size_t hi, void* c) {
    pair_ctx* m = (pair_ctx*)c;
    for (size_t i = lo; i < hi; i++) {
        g1p P; g2p Q; fp12 f;
        if (!g1_load96(&P, m->a1 + 96 * i) || !g2_load192(&Q, m->a2 + 192 * i)) { m->bad = 1; continue; }
        pair_ate(&f, &Q, &P); pair_fexp(&f);
        fp12_to_bytes(m->out + 576 * i, &f);
    }
}
int orc_pair_batch(size_t n, const uint8_t* g1_96, const uint8_t* g2_192, uint8_t* gt576, int nthreads) {
    INIT();
    pair_ctx m = {g1_96, g2_192, NULL, NULL, gt576, 0};
    par_for(n, nthreads, pair_range, &m);
    return m.bad ? -2 : 0;
}
/* PAIR_ate alone and PAIR_fexp alone on FP12_toOctet bytes */
int orc_miller_batch(size_t n, const uint8_t* g1_96, const uint8_t* g2_192, uint8_t* out576) {
    INIT();
    for (size_t i = 0; i < n; i++) {
        g1p P; g2p Q; fp12 f;
        if (!g1_load96(&P, g1_96 + 96 * i) || !g2_load192(&Q, g2_192 + 192 * i)) return -2;
        pair_ate(&f, &Q, &P);
        fp12_to_bytes(out576 + 576 * i, &f);
    }
    return 0;
}
int orc_fexp_batch(size_t n, const uint8_t* in576, uint8_t* out576) {
    INIT();
    for (size_t i = 0; i < n; i++) { fp12 f; fp12_from_bytes(&f, in576 + 576 * i); pair_fexp(&f); fp12_to_bytes(out576 + 576 * i, &f); }
    return 0;
}
/* pair(a1,a2) == pair(b1,b2) as include/crypto12381/liner_pair.hpp:339-350 of the reference */
static void pair_eq_range(size_t lo, size_t hi, void* c) {
    pair_ctx* m = (pair_ctx*)c;
    for (size_t i = lo; i < hi; i++) {
        g1p P, R; g2p Q, S; fp12 f, g, gc;
        if (!g1_load96(&P, m->a1 + 96 * i) || !g2_load192(&Q, m->a2 + 192 * i) ||
            !g1_load96(&R, m->b1 + 96 * i) || !g2_load192(&S, m->b2 + 192 * i)) { m->bad = 1; continue; }
        pair_ate(&f, &Q, &P); pair_ate(&g, &S, &R);
        fp12_conj(&gc, &g); fp12_mul(&f, &gc); pair_fexp(&f);
        m->out[i] = (uint8_t)fp12_is_unity(&f);
    }
}
int orc_pair_eq_batch(size_t n, const uint8_t* a1, const uint8_t* a2, const uint8_t* b1, const uint8_t* b2, uint8_t* ok, int nthreads) {
    INIT();
    pair_ctx m = {a1, a2, b1, b2, ok, 0};
    par_for(n, nthreads, pair_eq_range, &m);
    return m.bad ? -2 : 0;
}
/* PAIR_double_ate :508-626 followed by PAIR_fexp: the product of two Miller values (shared squarings
 * are an evaluation strategy); infinity G1 arguments contribute 1 (:532-541) */
int orc_pair2_batch(size_t n, const uint8_t* a1, const uint8_t* a2, const uint8_t* b1, const uint8_t* b2, uint8_t* gt576) {
    INIT();
    for (size_t i = 0; i < n; i++) {
        g1p P, R; g2p Q, S; fp12 f, g;
        if (!g1_load96(&P, a1 + 96 * i) || !g2_load192(&Q, a2 + 192 * i) ||
            !g1_load96(&R, b1 + 96 * i) || !g2_load192(&S, b2 + 192 * i)) return -2;
        pair_ate(&f, &Q, &P); pair_ate(&g, &S, &R); fp12_mul(&f, &g); pair_fexp(&f);
        fp12_to_bytes(gt576 + 576 * i, &f);
    }
    return 0;
}
/* GT ops on canonical bytes: 0 multiply (FP12_mul), 1 conjugate (FP12_conj), 2 pow (FP12_pow; b = 32-byte exponent, used as given) */
int orc_gt_op_batch(int op, size_t n, const uint8_t* a576, const uint8_t* b, uint8_t* out576) {
    INIT();
    for (size_t i = 0; i < n; i++) {
        fp12 x, y, r;
        fp12_from_bytes(&x, a576 + 576 * i);
        if (op == 0) { fp12_from_bytes(&y, b + 576 * i); fp12_mul(&x, &y); r = x; }
        else if (op == 1) fp12_conj(&r, &x);
        else if (op == 2) {
            uint64_t e[4]; const uint8_t* s = b + 32 * i;
            for (int k = 0; k < 4; k++) { uint64_t w = 0; for (int j = 0; j < 8; j++) w = (w << 8) | s[(3 - k) * 8 + j]; e[k] = w; }
            fp12_pow(&r, &x, e, 4);
        } else return -1;
        fp12_to_bytes(out576 + 576 * i, &r);
    }
    return 0;
}

static int zr_geq_fwd(const uint64_t* a) { for (int i = 3; i >= 0; i--) { if (a[i] > RORD[i]) return 1; if (a[i] < RORD[i]) return 0; } return 1; }
/* threaded forms of the split pairing (CPU baselines) */
static void miller_range(size_t lo, size_t hi, void* c) {
    pair_ctx* m = (pair_ctx*)c;
    for (size_t i = lo; i < hi; i++) {
        g1p P; g2p Q; fp12 f;
        if (!g1_load96(&P, m->a1 + 96 * i) || !g2_load192(&Q, m->a2 + 192 * i)) { m->bad = 1; continue; }
        pair_ate(&f, &Q, &P);
        fp12_to_bytes(m->out + 576 * i, &f);
    }
}
int orc_miller_batch_t(size_t n, const uint8_t* g1_96, const uint8_t* g2_192, uint8_t* out576, int nthreads) {
    INIT();
    pair_ctx m = {g1_96, g2_192, NULL, NULL, out576, 0};
    par_for(n, nthreads, miller_range, &m);
    return m.bad ? -2 : 0;
}
static void fexp_range(size_t lo, size_t hi, void* c) {
    pair_ctx* m = (pair_ctx*)c;
    for (size_t i = lo; i < hi; i++) { fp12 f; fp12_from_bytes(&f, m->a1 + 576 * i); pair_fexp(&f); fp12_to_bytes(m->out + 576 * i, &f); }
}
int orc_fexp_batch_t(size_t n, const uint8_t* in576, uint8_t* out576, int nthreads) {
    INIT();
    pair_ctx m = {in576, NULL, NULL, NULL, out576, 0};
    par_for(n, nthreads, fexp_range, &m);
    return 0;
}

/* ------------------------------------------------------------------ BBS+ verification
 * examples/bbs-plus/src/bbs+.cpp:57-73:  pair(A, w * (g2^x)) == pair(g1 * (h0^r) * Π[n](h[i]^m[i]), g2), evaluated with the
 * boundary's operations: multiply (PAIR_G2mul / PAIR_G1mul), add, and pair == pair as liner_pair.hpp:339-350 (two PAIR_ate,
 * FP12_conj, FP12_mul, one PAIR_fexp, FP12_isunity).  Messages message-major: block i of signature j at m32[(i * n + j) * 32]. */
typedef struct {
    size_t n, nmsg; g1p G1p, H0; g1p* H; g2p G2p, W;
    const uint8_t *A, *x, *r, *m; uint8_t* ok;
    /* wire form */
    const uint8_t *sig, *msgs; size_t msg_len;
} bbs_ctx;
static int bbs_verify_one(const bbs_ctx* b, const g1p* A, const uint64_t x[4], const uint64_t r[4], const uint64_t (*m)[4]) {
    g2p Q = b->G2p; g2_mul(&Q, x);
    g2p Wc = b->W; g2_add(&Wc, &Q);
    g1p B = b->G1p, T = b->H0;
    g1_mul(&T, r); g1_add(&B, &T);
    for (size_t i = 0; i < b->nmsg; i++) { g1p Hi = b->H[i]; g1_mul(&Hi, m[i]); g1_add(&B, &Hi); }
    fp12 f, g, gc;
    pair_ate(&f, &Wc, A); pair_ate(&g, &b->G2p, &B);
    fp12_conj(&gc, &g); fp12_mul(&f, &gc); pair_fexp(&f);
    return fp12_is_unity(&f);
}
static void bbs_range(size_t lo, size_t hi, void* c) {
    bbs_ctx* b = (bbs_ctx*)c;
    uint64_t (*m)[4] = (uint64_t (*)[4])malloc((b->nmsg ? b->nmsg : 1) * sizeof *m);
    for (size_t j = lo; j < hi; j++) {
        g1p A; uint64_t x[4], r[4];
        if (!g1_load96(&A, b->A + 96 * j)) { b->ok[j] = 0xff; continue; }
        scalar_load(x, b->x + 32 * j); scalar_load(r, b->r + 32 * j);
        for (size_t i = 0; i < b->nmsg; i++) scalar_load(m[i], b->m + 32 * (i * b->n + j));
        b->ok[j] = (uint8_t)bbs_verify_one(b, &A, x, r, (const uint64_t (*)[4])m);
    }
    free(m);
}
int orc_bbs_plus_verify_batch(size_t n, size_t nmsg, const uint8_t* g1_96, const uint8_t* g2_192, const uint8_t* h0_96, const uint8_t* h_96,
                              const uint8_t* w_192, const uint8_t* A_96, const uint8_t* x32, const uint8_t* r32, const uint8_t* m32,
                              uint8_t* ok, int nthreads) {
    INIT();
    bbs_ctx b; memset(&b, 0, sizeof b);
    b.n = n; b.nmsg = nmsg; b.A = A_96; b.x = x32; b.r = r32; b.m = m32; b.ok = ok;
    if (!g1_load96(&b.G1p, g1_96) || !g1_load96(&b.H0, h0_96) || !g2_load192(&b.G2p, g2_192) || !g2_load192(&b.W, w_192)) return -2;
    b.H = (g1p*)malloc((nmsg ? nmsg : 1) * sizeof *b.H);
    for (size_t i = 0; i < nmsg; i++) if (!g1_load96(&b.H[i], h_96 + 96 * i)) { free(b.H); return -2; }
    par_for(n, nthreads, bbs_range, &b);
    free(b.H);
    return 0;
}
/* wire formats (see ref_wrap.cpp ref_bbs_plus_verify_wire_batch for the citations): parse<G1>/<G2> = leading 0x00 -> infinity,
 * else ECP_fromOctet / ECP2_fromOctet; parse<Zp> = 48 big-endian bytes below r; encode_to<Zp> = 31-byte units behind a 0x01 byte */
static int wire_g1(g1p* P, const uint8_t* s) {
    if (s[0] == 0) { g1_inf(P); return 1; }
    if (s[0] != 0x02 && s[0] != 0x03) return 0;             /* a 49-byte view cannot hold the 0x04 form */
    fp x; fp_from_bytes(&x, s + 1); return g1_setx(P, &x, s[0] & 1);
}
static int wire_g2(g2p* P, const uint8_t* s) {
    if (s[0] == 0) { g2_inf(P); return 1; }
    if (s[0] == 0x04) return 0;
    fp2 x; fp2_from_bytes(&x, s + 1); return g2_setx(P, &x, s[0] & 1);
}
static int wire_zp(uint64_t k[4], const uint8_t* b48) {
    for (int i = 0; i < 16; i++) if (b48[i]) return 0;       /* >= 2^256 > r */
    for (int i = 0; i < 4; i++) { uint64_t w = 0; for (int j = 0; j < 8; j++) w = (w << 8) | b48[16 + (3 - i) * 8 + j]; k[i] = w; }
    return !zr_geq_fwd(k);
}
static void encode_unit(uint64_t k[4], const uint8_t* msg, size_t msg_len, size_t i) {
    uint8_t buf[32] = {0};
    buf[0] = 1;
    const size_t len = (i + 1) * 31 <= msg_len ? 31 : msg_len - i * 31;
    memcpy(buf + 1, msg + 31 * i, len);
    for (int w = 0; w < 4; w++) { uint64_t v = 0; for (int j = 0; j < 8; j++) v = (v << 8) | buf[(3 - w) * 8 + j]; k[w] = v; }
}
static void bbs_wire_range(size_t lo, size_t hi, void* c) {
    bbs_ctx* b = (bbs_ctx*)c;
    uint64_t (*m)[4] = (uint64_t (*)[4])malloc((b->nmsg ? b->nmsg : 1) * sizeof *m);
    for (size_t j = lo; j < hi; j++) {
        const uint8_t* s = b->sig + 145 * j;
        g1p A; uint64_t x[4], r[4];
        if (!wire_g1(&A, s) || !wire_zp(x, s + 49) || !wire_zp(r, s + 97)) { b->ok[j] = 0xff; continue; }
        for (size_t i = 0; i < b->nmsg; i++) encode_unit(m[i], b->msgs + b->msg_len * j, b->msg_len, i);
        b->ok[j] = (uint8_t)bbs_verify_one(b, &A, x, r, (const uint64_t (*)[4])m);
    }
    free(m);
}
int orc_bbs_plus_verify_wire_batch(size_t n, size_t nh, size_t msg_len, const uint8_t* g1_g2_h0_195, const uint8_t* h49, const uint8_t* pk97,
                                   const uint8_t* sig145, const uint8_t* msgs, uint8_t* ok, int nthreads) {
    INIT();
    bbs_ctx b; memset(&b, 0, sizeof b);
    const size_t nblk = (msg_len + 30) / 31;
    if (nblk > nh) return -2;
    b.n = n; b.nmsg = nblk; b.ok = ok; b.sig = sig145; b.msgs = msgs; b.msg_len = msg_len;
    if (!wire_g1(&b.G1p, g1_g2_h0_195) || !wire_g2(&b.G2p, g1_g2_h0_195 + 49) || !wire_g1(&b.H0, g1_g2_h0_195 + 146) || !wire_g2(&b.W, pk97)) return -2;
    b.H = (g1p*)malloc((nblk ? nblk : 1) * sizeof *b.H);
    for (size_t i = 0; i < nblk; i++) if (!wire_g1(&b.H[i], h49 + 49 * i)) { free(b.H); return -2; }
    par_for(n, nthreads, bbs_wire_range, &b);
    free(b.H);
    return 0;
}
int orc_encode_to_zp(size_t msg_len, const uint8_t* msg, uint8_t* out32) {
    const size_t nblk = (msg_len + 30) / 31;
    for (size_t i = 0; i < nblk; i++) {
        uint64_t k[4]; encode_unit(k, msg, msg_len, i);
        for (int w = 0; w < 4; w++) for (int j = 0; j < 8; j++) out32[32 * i + (3 - w) * 8 + j] = (uint8_t)(k[w] >> (8 * (7 - j)));
    }
    return 0;
}

/* ------------------------------------------------------------------ hash-to-G1
 * G1Point::from_hash, include/crypto12381/g1_point.hpp:219-234, from the digest on:
 * ECP_map2point ecp_BLS12381.cpp:1495-1626 (simplified SWU on E' + 11-isogeny, hint-sharing FP_qr/FP_inv/FP_sqrt
 * fp_BLS12381.cpp:800-876 with PM1D2 = 1) and ECP_cfp :1252-1273 (multiply by CURVE_Cof = 1 - x). */
#include "h2c_consts.h"
static fp SSWU_A, SSWU_B, SSWU_HINTZ, ISO_XN[12], ISO_XD[10], ISO_YN[16], ISO_YD[15];
static uint64_t E_PM3D4[6];
static pthread_once_t once_h2c = PTHREAD_ONCE_INIT;
static void fp_from_hex(fp* r, const char* h) { uint8_t b[48]; hex48(b, h); fp_from_bytes(r, b); }
static void init_h2c(void) {
    INIT();
    uint64_t pm3[6]; memcpy(pm3, P, 48); pm3[0] -= 3;
    bn_div_small(E_PM3D4, pm3, 6, 4);
    fp_from_hex(&SSWU_A, SSWU_A_HEX); fp_from_hex(&SSWU_B, SSWU_B_HEX);
    for (int i = 0; i < 12; i++) fp_from_hex(&ISO_XN[i], ISO11_XNUM_HEX[i]);
    for (int i = 0; i < 10; i++) fp_from_hex(&ISO_XD[i], ISO11_XDEN_HEX[i]);
    for (int i = 0; i < 16; i++) fp_from_hex(&ISO_YN[i], ISO11_YNUM_HEX[i]);
    for (int i = 0; i < 15; i++) fp_from_hex(&ISO_YD[i], ISO11_YDEN_HEX[i]);
    fp z; fp_set_int(&z, 11);
    fp_pow(&SSWU_HINTZ, &z, E_PM3D4, 6);                 /* CURVE_HTPC = Z^((p-3)/4) */
}
/* FP_progen :782-797 with e = 1 */
static void fp_progen(fp* r, const fp* x) { fp_pow(r, x, E_PM3D4, 6); }
/* FP_qr :800-813 with hint */
static int fp_qr_hint(const fp* x, fp* h) { fp r; fp_progen(&r, x); *h = r; fp_sqr(&r, &r); fp_mul(&r, x, &r); return fp_eq(&r, &ONE); }
/* FP_inv :817-840 with hint: x * hint^4 */
static void fp_inv_hint(fp* r, const fp* x, const fp* h) { fp t; fp_sqr(&t, h); fp_sqr(&t, &t); fp_mul(r, &t, x); }
/* FP_sqrt :842-876 with hint, e = 1: hint * a, then the even ("positive") root */
static void fp_sqrt_hint(fp* r, const fp* a, const fp* h) { fp v; fp_mul(r, h, a); if (fp_sign(r)) { fp_neg(&v, r); *r = v; } }
static void horner(fp* r, const fp* cs, int n, const fp* x, int monic) {
    fp acc;
    if (monic) fp_add(&acc, x, &cs[n - 1]); else acc = cs[n - 1];
    for (int i = n - 2; i >= 0; i--) { fp_mul(&acc, &acc, x); fp_add(&acc, &acc, &cs[i]); }
    *r = acc;
}
static void g1_map2point(g1p* P, const fp* h) {
    fp X1, X2, X3, t, w, D, D2, hint, GX1, Y;
    int sgn = fp_sign(h);
    fp_sqr(&t, h); fp_imul(&t, &t, 11);
    fp_add(&w, &t, &ONE);
    fp_mul(&w, &w, &t);
    fp_mul(&D, &SSWU_A, &w);
    fp_add(&w, &w, &ONE); fp_mul(&w, &w, &SSWU_B); fp_neg(&w, &w);
    X2 = w; fp_mul(&X3, &t, &X2);
    fp_sqr(&GX1, &X2);
    fp_sqr(&D2, &D); fp_mul(&w, &SSWU_A, &D2); fp_add(&GX1, &GX1, &w); fp_mul(&GX1, &GX1, &X2);
    fp_mul(&D2, &D2, &D); fp_mul(&w, &SSWU_B, &D2); fp_add(&GX1, &GX1, &w);
    fp_mul(&w, &GX1, &D);
    int qr = fp_qr_hint(&w, &hint);
    fp_inv_hint(&D, &w, &hint);
    fp_mul(&D, &D, &GX1);
    fp_mul(&X2, &X2, &D); fp_mul(&X3, &X3, &D);
    fp_mul(&t, &t, h);
    fp_sqr(&D2, &D);
    fp_mul(&D, &D2, &t);
    fp_imul(&t, &w, 11);
    fp_mul(&X1, &SSWU_HINTZ, &hint);
    if (!qr) { X2 = X3; D2 = D; w = t; hint = X1; }
    fp_sqrt_hint(&Y, &w, &hint);
    fp_mul(&Y, &Y, &D2);
    if (fp_sign(&Y) ^ sgn) fp_neg(&Y, &Y);
    fp xnum, xden, ynum, yden;
    horner(&xnum, ISO_XN, 12, &X2, 0); horner(&xden, ISO_XD, 10, &X2, 1);
    horner(&ynum, ISO_YN, 16, &X2, 0); horner(&yden, ISO_YD, 15, &X2, 1);
    fp_mul(&ynum, &ynum, &Y);
    fp_mul(&P->x, &xnum, &yden); fp_mul(&P->y, &ynum, &xden); fp_mul(&P->z, &xden, &yden);
}
int orc_g1_from_hash_batch(size_t n, const uint8_t* digests64, uint8_t* out, int out_fmt) {
    pthread_once(&once_h2c, init_h2c);
    for (size_t i = 0; i < n; i++) {
        /* 512-bit big-endian integer mod p (fixed_time_mod), in Montgomery form (residue -> FP_nres) */
        fp u, b; memset(&u, 0, sizeof u);
        for (int j = 0; j < 64; j++) { fp_imul(&u, &u, 256); fp_set_int(&b, digests64[64 * i + j]); fp_add(&u, &u, &b); }
        g1p Q; g1_map2point(&Q, &u);
        uint64_t cof[1] = {BNX + 1};                       /* CURVE_Cof = 1 - x (x negative) */
        g1_mul_plain(&Q, cof, 1);
        g1_store(out + (size_t)out_fmt * i, &Q, out_fmt);
    }
    return 0;
}

/* ------------------------------------------------------------------ Zp helpers: arithmetic mod r
 * on canonical 32-byte values, zp_number.hpp:295-380 (operator*), :420-425 (inverse -> BIG_invmodp), :540-548 (from_hash) */
typedef struct { uint64_t w[4]; } zr;
static int zr_geq(const uint64_t* a) { for (int i = 3; i >= 0; i--) { if (a[i] > RORD[i]) return 1; if (a[i] < RORD[i]) return 0; } return 1; }
static void zr_subr(uint64_t* a) { u128 bw = 0; for (int i = 0; i < 4; i++) { u128 d = (u128)a[i] - RORD[i] - bw; a[i] = (uint64_t)d; bw = (d >> 64) & 1; } }
static void zr_add(zr* r, const zr* a, const zr* b) {
    u128 c = 0; uint64_t t[4];
    for (int i = 0; i < 4; i++) { c += (u128)a->w[i] + b->w[i]; t[i] = (uint64_t)c; c >>= 64; }
    if (c || zr_geq(t)) zr_subr(t);
    memcpy(r->w, t, 32);
}
static void zr_mul(zr* r, const zr* a, const zr* b) {           /* double-and-add: slow and obviously right */
    zr acc; memset(&acc, 0, sizeof acc);
    for (int i = 255; i >= 0; i--) { zr_add(&acc, &acc, &acc); if ((b->w[i / 64] >> (i % 64)) & 1) zr_add(&acc, &acc, a); }
    *r = acc;
}
static void zr_from_be(zr* r, const uint8_t* s, int len) {      /* big-endian integer of any length mod r */
    zr acc; memset(&acc, 0, sizeof acc);
    for (int j = 0; j < len; j++) {
        for (int k = 0; k < 8; k++) zr_add(&acc, &acc, &acc);
        zr b; memset(&b, 0, sizeof b); b.w[0] = s[j]; zr_add(&acc, &acc, &b);
    }
    *r = acc;
}
static void zr_to_be(uint8_t* out, const zr* a) { for (int i = 0; i < 32; i++) out[i] = (uint8_t)(a->w[(31 - i) / 8] >> (8 * ((31 - i) % 8))); }
int orc_zp_op_batch(int op, size_t n, const uint8_t* a32, const uint8_t* b32, uint8_t* out32) {
    for (size_t i = 0; i < n; i++) {
        zr a, b, r; zr_from_be(&a, a32 + 32 * i, 32);
        if (b32 && op <= 2) zr_from_be(&b, b32 + 32 * i, 32); else b = a;
        zr nb; memset(&nb, 0, sizeof nb);
        switch (op) {
            case 0: zr_mul(&r, &a, &b); break;
            case 1: zr_add(&r, &a, &b); break;
            case 2: case 3: {
                const zr* v = op == 2 ? &b : &a;
                if (v->w[0] | v->w[1] | v->w[2] | v->w[3]) { u128 bw = 0; for (int k = 0; k < 4; k++) { u128 d = (u128)RORD[k] - v->w[k] - bw; nb.w[k] = (uint64_t)d; bw = (d >> 64) & 1; } }
                if (op == 2) zr_add(&r, &a, &nb); else r = nb;
                break;
            }
            case 4: {                                            /* a^(r-2); 0 -> 0 */
                uint64_t e[4]; memcpy(e, RORD, 32); e[0] -= 2;
                zr acc; memset(&acc, 0, sizeof acc); acc.w[0] = 1;
                for (int k = 254; k >= 0; k--) { zr_mul(&acc, &acc, &acc); if ((e[k / 64] >> (k % 64)) & 1) zr_mul(&acc, &acc, &a); }
                r = acc; break;
            }
            default: return -1;
        }
        zr_to_be(out32 + 32 * i, &r);
    }
    return 0;
}
int orc_zp_from_hash_batch(size_t n, const uint8_t* digests64, uint8_t* out32) {
    for (size_t i = 0; i < n; i++) { zr r; zr_from_be(&r, digests64 + 64 * i, 64); zr_to_be(out32 + 32 * i, &r); }
    return 0;
}
