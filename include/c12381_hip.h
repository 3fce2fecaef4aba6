/* c12381_hip.h — C ABI of the MI355X-native batched BLS12-381 backend.
 *
 * This is the drop-in boundary for crypto12381's hot path.  The reference's seam is the set of
 * C++ free functions in namespace crypto12381::detail::miracl_core
 * (include/crypto12381/miracl_core_interface.hpp:16-204, defined in
 * src/miracl_core_interface.cpp:12-289 of the reference); it is scalar (one element per call)
 * and cannot express a 2^20 batch, so the entry points below are the batched forms of the
 * throughput functions of that seam.  Each entry cites the reference function it replaces.
 * INTEGRATION.md shows the reference-side binding (a replacement miracl_core_interface.cpp).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; every function returns 0 on success or a negative
 *     C12381_E_* code; no exceptions, no allocation visible to the caller.
 *   - All data is canonical big-endian bytes, bit-identical to the reference's encodings:
 *       Fp      48 B
 *       scalar  32 B   any value < 2^256; reduced mod r first (PAIR_G1mul pair_BLS12381.cpp:879-881)
 *       G1      96 B   x||y affine (ECP_toOctet uncompressed body, ecp_BLS12381.cpp:478-488);
 *                      96 zero bytes = point at infinity
 *               49 B   02|03 || x   compressed; 49 zero bytes = infinity (g1_point.hpp:113-117)
 *       G2     192 B   x.b||x.a||y.b||y.a (FP2_toBytes fp2_BLS12381.cpp:83-87: imaginary part first);
 *                      zeros = infinity;   97 B compressed 02|03 || x.b || x.a
 *       GT     576 B   c||b||a, each Fp4 b||a, each Fp2 b||a (FP12_toOctet fp12_BLS12381.cpp:923-929)
 *   - "host" entry points take host pointers (the library stages through its own device buffers);
 *     "_dev" entry points take DEVICE pointers (16-byte aligned) and run asynchronously on the
 *     context's stream; call c12381_sync() before reading results.
 *   - Stream ordering of "_dev" calls.  Every kernel of a context runs on the CONTEXT'S stream (its own, or the
 *     one given to c12381_set_stream) and only there: inputs must be COMPLETE ON THAT STREAM when the call
 *     is made, and outputs are ordered only with later work on that stream.  A caller that fills the inputs
 *     on another stream (a framework's default stream, a copy stream) records an event there and hands it to
 *     c12381_wait_event() before the call; a caller that consumes the outputs on another stream calls
 *     c12381_record_event() after it and waits for that event on its own stream.  Neither blocks the host.
 *     Without such an edge the library may read half-written inputs: by construction it cannot see a foreign
 *     stream (c12381_sync() or a device-wide synchronise are the host-blocking alternatives).
 *   - One context per host thread / per GPU; contexts are independent (thread-compatible).
 *   - There is no CPU fallback: without a usable HIP device c12381_create fails.
 */
#ifndef C12381_HIP_H
#define C12381_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct c12381_ctx c12381_ctx;

enum {
    C12381_OK = 0,
    C12381_E_ARG = -1,      /* bad argument (null pointer, unknown format or op) */
    C12381_E_HIP = -2,      /* HIP runtime error; see c12381_last_error */
    C12381_E_POINT = -3,    /* at least one input point is not on the curve (its output is all 0xff) */
    C12381_E_NOMEM = -4,
    C12381_E_INTERNAL = -5  /* library-internal failure (a work-queue hand-over between wavefronts timed out); the
                               affected outputs are all 0xff, never stale values; see c12381_last_error */
};

/* context -------------------------------------------------------------------------------- */
int c12381_create(int device, c12381_ctx** out);
void c12381_destroy(c12381_ctx* ctx);
const char* c12381_last_error(const c12381_ctx* ctx);
/* use an existing hipStream_t (passed as void*) for all work of this context; NULL = own stream */
int c12381_set_stream(c12381_ctx* ctx, void* hip_stream);
/* wait for the context's stream; returns C12381_E_INTERNAL / C12381_E_POINT if any kernel since the last
 * status read (every host entry point and every c12381_sync reads and clears the status) raised one:
 * callers of _dev entry points separate logical operations with c12381_sync() */
int c12381_sync(c12381_ctx* ctx);
/* Device-side ordering against other streams for the "_dev" entry points (see "Stream ordering" above); `hip_event` is a hipEvent_t
 * passed as void*, created and destroyed by the caller.
 *   c12381_wait_event:   all work the context launches AFTER this call waits for the event (hipStreamWaitEvent on the context's
 *                        stream): record the event on the stream that produced the inputs, then call this, then the _dev entry point.
 *   c12381_record_event: records the event on the context's stream behind everything launched so far (including the library's
 *                        side-stream work, which is joined first): make the consuming stream wait for it.
 * No counterpart in the reference (its seam is synchronous host code, include/crypto12381/miracl_core_interface.hpp:186-204). */
int c12381_wait_event(c12381_ctx* ctx, void* hip_event);
int c12381_record_event(c12381_ctx* ctx, void* hip_event);
/* Device workspaces grow to the largest call a context has served and are kept for the next one (a 2^20 G1 batch: 2.95 GB of window tables;
 * a GT power of 2^16 elements: 1.4 GB of power tables; 2^18 BBS+ verifications: a 172 MB state slab).  c12381_trim waits for the context's
 * streams and frees all of them; cached fixed-base / line tables are rebuilt by the next call that needs them.  (The reference's seam never
 * allocates: src/miracl_core_interface.cpp works on caller-owned PODs.) */
int c12381_trim(c12381_ctx* ctx);
/* Per-kernel timing with HIP events on the context's stream (used by bench.py for the roofline
 * figure).  enable != 0 starts a fresh recording; kind: 0 = G1 scalar-mul kernel, 1 = G1 finish
 * (inversion + encode) kernel, 2 = G2 scalar-mul kernel, 3 = pairing kernel, 4 = pairing-equality kernel, 5 = MSM bucket kernel,
 * 6 = Miller-loop kernel (c12381_miller_batch_dev), 7 = GT operation / final-exponentiation kernel (c12381_gt_op_batch_dev).
 * c12381_profile_read synchronises the stream. */
int c12381_profile(c12381_ctx* ctx, int enable);
int c12381_profile_read(c12381_ctx* ctx, int kind, double* total_ms, uint64_t* launches);
/* ABI version: major << 16 | minor */
int c12381_version(void);

/* Fp (test / roofline hook) --------------------------------------------------------------- */
/* op: 0 mul, 1 add, 2 sub, 3 sqr, 4 neg, 5 inv.  Replaces FP_nres + FP_mul/FP_add/FP_sub/FP_sqr/
 * FP_neg/FP_inv + FP_redc (fp_BLS12381.cpp:223,396,485,500,466,588,817,234).  b may be NULL for
 * unary ops. */
int c12381_fp_op_batch(c12381_ctx* ctx, int op, size_t n, const uint8_t* a48, const uint8_t* b48, uint8_t* out48);
int c12381_fp_op_batch_dev(c12381_ctx* ctx, int op, size_t n, const uint8_t* a48, const uint8_t* b48, uint8_t* out48);
/* register-resident chain of `iters` Montgomery multiplications per element (x <- x*y), used to
 * calibrate the integer-VALU roofline; out = final x */
int c12381_fp_mulchain_dev(c12381_ctx* ctx, size_t n, int iters, const uint8_t* a48, const uint8_t* b48, uint8_t* out48);

/* G1 ------------------------------------------------------------------------------------- */
/* out[i] = scalars[i] * pts[i].  Batched form of multiply(point1&, const big&)
 * (miracl_core_interface.hpp:122, src/miracl_core_interface.cpp:174-177 -> PAIR_G1mul) followed by
 * to_bytes(bytes_view&, point1&, compressed) (:113-116 -> ECP_toOctet).  out_fmt = 49 or 96. */
int c12381_g1_mul_batch(c12381_ctx* ctx, size_t n, const uint8_t* pts96, const uint8_t* scalars32, uint8_t* out, int out_fmt);
int c12381_g1_mul_batch_dev(c12381_ctx* ctx, size_t n, const uint8_t* pts96, const uint8_t* scalars32, uint8_t* out, int out_fmt);
/* The same with caller assertions.  C12381_F_IN_SUBGROUP: every input point lies in the order-r subgroup (G1 / G2).
 * Why it exists: the reference never checks membership and its multiply() (PAIR_G1mul pair_BLS12381.cpp:876-924 after
 * glv() :793-805; PAIR_G2mul :927-983 after gs() :814-873) adds [r]phi(P) to the result when k mod r < x^2 (any scalar of
 * at most 127 bits) and [r]psi^i(Q) for a zero odd base-|x| digit; those terms vanish on the subgroup and are cofactor
 * points elsewhere.  The default entry points reproduce them for EVERY curve point, which costs each such lane a
 * membership test as long as a scalar multiplication (2^20 G1 multiplications with 64-bit scalars: 40 ms instead of 28 ms).
 * With the flag the test is skipped: results are identical on subgroup points (tests/test_gpu_g1.py) and are plain
 * [k mod x^2]P - [k div x^2]phi(P) off it. */
#define C12381_F_IN_SUBGROUP 1u
/* C12381_F_COMPRESSED_IN: the points arrive in their serialized form — n x 49 bytes (G1) / n x 97 bytes (G2) as the header layer
 * parses them (g1_point.hpp:87-111, g2_point.hpp:73-77: a leading 0x00 is the point at infinity) in front of from_bytes(point1& /
 * point2&, bytes_view&) (src/miracl_core_interface.cpp:109-112, 187-190 -> ECP_fromOctet ecp_BLS12381.cpp:495-545 / ECP2_fromOctet
 * ecp2_BLS12381.cpp:225-266: tags 0x02 / 0x03 (G2: any tag but 0x04), x taken mod p, on-curve by construction, no subgroup check).
 * from_bytes -> multiply / pair / product is then ONE call: the square root runs in the prologue of the scalar-multiplication and
 * MSM-preparation kernels (the pairing entry decodes into a workspace first: its kernels read their inputs once per queue task).
 * A rejected encoding behaves like a point that is not on the curve: its output lane is 0xff (an MSM leaves the term out) and the
 * call returns C12381_E_POINT — the status semantics of c12381_g1_decompress_batch.  Accepted by c12381_g1_mul_batch_flags,
 * c12381_g2_mul_batch_flags, c12381_g1_msm_flags, c12381_pair_batch_flags and their _dev forms. */
#define C12381_F_COMPRESSED_IN 4u
int c12381_g1_mul_batch_flags(c12381_ctx* ctx, size_t n, const uint8_t* pts96, const uint8_t* scalars32, uint8_t* out, int out_fmt, unsigned flags);
int c12381_g1_mul_batch_flags_dev(c12381_ctx* ctx, size_t n, const uint8_t* pts96, const uint8_t* scalars32, uint8_t* out, int out_fmt, unsigned flags);
/* out[i] = a[i] + b[i].  Batched add(point1&, point1&) (:129-132 -> ECP_add). */
int c12381_g1_add_batch(c12381_ctx* ctx, size_t n, const uint8_t* a96, const uint8_t* b96, uint8_t* out, int out_fmt);
/* out = sum_i scalars[i] * pts[i]  (the reference's Π[n](g[i]^x[i]), g1_point.hpp:371-404, and
 * sum_of_products(point1&, int, point1*, const big*) :134-137 -> ECP_muln). */
int c12381_g1_msm(c12381_ctx* ctx, size_t n, const uint8_t* pts96, const uint8_t* scalars32, uint8_t* out, int out_fmt);
int c12381_g1_msm_dev(c12381_ctx* ctx, size_t n, const uint8_t* pts96, const uint8_t* scalars32, uint8_t* out, int out_fmt);
/* the same with C12381_F_COMPRESSED_IN (pts = n x 49 bytes): parse<G1> + Π in one call */
int c12381_g1_msm_flags(c12381_ctx* ctx, size_t n, const uint8_t* pts, const uint8_t* scalars32, uint8_t* out, int out_fmt, unsigned flags);
int c12381_g1_msm_flags_dev(c12381_ctx* ctx, size_t n, const uint8_t* pts, const uint8_t* scalars32, uint8_t* out, int out_fmt, unsigned flags);
/* out = sum_i pts[i] (no scalars): the header layer's product over G1Point values — a chain of add(point1&, point1&)
 * (miracl_core_interface.hpp:101, src/miracl_core_interface.cpp:129-132 -> ECP_add) — and the combine step of a product sharded
 * over GPUs (N partial points of 96 B).  n = 0 gives infinity; a point that is not on the curve is left out and
 * reported (C12381_E_POINT). */
int c12381_g1_sum(c12381_ctx* ctx, size_t n, const uint8_t* pts96, uint8_t* out, int out_fmt);
int c12381_g1_sum_dev(c12381_ctx* ctx, size_t n, const uint8_t* pts96, uint8_t* out, int out_fmt);
/* The boundary function sum_of_products(point1&, int, point1*, const big*) with the reference's value for EVERY input
 * (src/miracl_core_interface.cpp:134-137 -> ECP_muln ecp_BLS12381.cpp:1112-1148, a plain Pippenger): the sum of the true multiples
 * [k_i mod r]P_i.  For points of G1 this equals c12381_g1_msm, which is the fast path; off the subgroup the two differ because the
 * header-level Π — what c12381_g1_msm reproduces — goes through multiply()'s GLV form (tests/golden/g1.json
 * offsubgroup_msm49 vs offsubgroup_sum_of_products49). */
int c12381_g1_sum_of_products(c12381_ctx* ctx, size_t n, const uint8_t* pts96, const uint8_t* scalars32, uint8_t* out, int out_fmt);
int c12381_g1_sum_of_products_dev(c12381_ctx* ctx, size_t n, const uint8_t* pts96, const uint8_t* scalars32, uint8_t* out, int out_fmt);
/* the same product over several GPUs driven by ONE host process: terms split contiguously over ctxs[0..ngpu-1] (one
 * context per device, created by the caller), local MSMs run concurrently, the ngpu partial points are summed on
 * ctxs[0] (the combine is an elliptic-curve addition, 96 B per GPU). */
int c12381_g1_msm_multi(c12381_ctx** ctxs, int ngpu, size_t n, const uint8_t* points96, const uint8_t* scalars32, uint8_t* out, int out_fmt);

/* G2 ------------------------------------------------------------------------------------- */
/* out[i] = scalars[i] * pts[i].  Batched multiply(point2&, const big&) (miracl_core_interface.hpp:152,
 * src/miracl_core_interface.cpp:202-205 -> PAIR_G2mul) + to_bytes(bytes_view&, point2&, compressed)
 * (:192-195 -> ECP2_toOctet).  out_fmt = 97 or 192. */
int c12381_g2_mul_batch(c12381_ctx* ctx, size_t n, const uint8_t* pts192, const uint8_t* scalars32, uint8_t* out, int out_fmt);
int c12381_g2_mul_batch_dev(c12381_ctx* ctx, size_t n, const uint8_t* pts192, const uint8_t* scalars32, uint8_t* out, int out_fmt);
/* with caller assertions (C12381_F_IN_SUBGROUP, see c12381_g1_mul_batch_flags) */
int c12381_g2_mul_batch_flags(c12381_ctx* ctx, size_t n, const uint8_t* pts192, const uint8_t* scalars32, uint8_t* out, int out_fmt, unsigned flags);
int c12381_g2_mul_batch_flags_dev(c12381_ctx* ctx, size_t n, const uint8_t* pts192, const uint8_t* scalars32, uint8_t* out, int out_fmt, unsigned flags);
/* out = sum_i scalars[i] * pts[i] in G2; scalars == NULL: the plain sum of the points — product(type_identity<G2Point>, r)
 * (g2_point.hpp:225-236: get_infinity + a chain of add(point2&, point2&)); with scalars it is what the header layer evaluates
 * for Π[n](q[i]^x[i]) (eager multiply(point2&, big) :202-217, then that chain).  n = 0 gives infinity. */
int c12381_g2_msm(c12381_ctx* ctx, size_t n, const uint8_t* pts192, const uint8_t* scalars32, uint8_t* out, int out_fmt);
int c12381_g2_msm_dev(c12381_ctx* ctx, size_t n, const uint8_t* pts192, const uint8_t* scalars32, uint8_t* out, int out_fmt);
/* out[i] = a[i] + b[i].  Batched add(point2&, point2&) (:212-215 -> ECP2_add). */
int c12381_g2_add_batch(c12381_ctx* ctx, size_t n, const uint8_t* a192, const uint8_t* b192, uint8_t* out, int out_fmt);

/* pairing -------------------------------------------------------------------------------- */
/* gt[i] = e(g1[i], g2[i]): pair_ate(fp12&, point2&, point1&) + pair_final_exponentiation(fp12&) +
 * to_bytes(bytes_view&, fp12&) (miracl_core_interface.hpp:199-201,189; src/miracl_core_interface.cpp:
 * 276-284, 246-249 -> PAIR_ate, PAIR_fexp, FP12_toOctet). */
int c12381_pair_batch(c12381_ctx* ctx, size_t n, const uint8_t* g1_96, const uint8_t* g2_192, uint8_t* gt576);
int c12381_pair_batch_dev(c12381_ctx* ctx, size_t n, const uint8_t* g1_96, const uint8_t* g2_192, uint8_t* gt576);
/* the same with C12381_F_COMPRESSED_IN (g1 = n x 49, g2 = n x 97 bytes): from_bytes x 2 + pair in one call */
int c12381_pair_batch_flags(c12381_ctx* ctx, size_t n, const uint8_t* g1, const uint8_t* g2, uint8_t* gt576, unsigned flags);
int c12381_pair_batch_flags_dev(c12381_ctx* ctx, size_t n, const uint8_t* g1, const uint8_t* g2, uint8_t* gt576, unsigned flags);
/* gt[i] = prod_{j < k} e(g1s[j * n + i], g2s[j * n + i]) for k = 1, 2 or 3 (argument-major arrays of k * n points):
 * pair(a, b) * pair(c, d) as the header layer forms it (liner_pair.hpp:291-303) ->
 * pair_double_ate(fp12&, point2&, point1&, point2&, point1&) (miracl_core_interface.hpp:203-204,
 * src/miracl_core_interface.cpp:286-289 -> PAIR_double_ate pair_BLS12381.cpp:508-626) + pair_final_exponentiation: ONE
 * joint Miller loop with shared Fp12 squarings and ONE final exponentiation per element (k = 3: the triple products of
 * the PS / bbs04 / AC-* examples).  A G1 argument at infinity contributes 1 (:532-541).
 * flags: C12381_F_MILLER_ONLY stops before the final exponentiation — the output is the reference's Miller value
 * (FP12_toOctet bytes of the product of the k single-loop values), what pair_double_ate itself returns. */
#define C12381_F_MILLER_ONLY 2u
int c12381_pair_product_batch(c12381_ctx* ctx, size_t n, int k, const uint8_t* g1s_96, const uint8_t* g2s_192, uint8_t* gt576, unsigned flags);
int c12381_pair_product_batch_dev(c12381_ctx* ctx, size_t n, int k, const uint8_t* g1s_96, const uint8_t* g2s_192, uint8_t* gt576, unsigned flags);
/* ok[i] = (e(a1[i], a2[i]) == e(b1[i], b2[i])) as the reference's header evaluates it
 * (liner_pair.hpp:339-350): two pair_ate, conjugate, multiply, ONE pair_final_exponentiation, is_unity.
 * ok[i] is 1 / 0, or 0xff when an input point of lane i is not on its curve. */
int c12381_pair_eq_batch(c12381_ctx* ctx, size_t n, const uint8_t* a1_96, const uint8_t* a2_192, const uint8_t* b1_96,
                         const uint8_t* b2_192, uint8_t* ok);
int c12381_pair_eq_batch_dev(c12381_ctx* ctx, size_t n, const uint8_t* a1_96, const uint8_t* a2_192, const uint8_t* b1_96,
                             const uint8_t* b2_192, uint8_t* ok);

/* wire formats either side of the path -------------------------------------- */
/* Batched from_bytes(point1&, bytes_view&) for 49-byte compressed input (miracl_core_interface.hpp:92,
 * src/miracl_core_interface.cpp:109-112 -> ECP_fromOctet -> ECP_setx: one square root per point, on-curve by
 * construction, NO subgroup check, x taken mod p).  A leading 0x00 byte means infinity (g1_point.hpp:89-93).
 * status[i] = 1 ok / 0 reject; out96[i] = x||y, zeros for infinity or reject. */
int c12381_g1_decompress_batch(c12381_ctx* ctx, size_t n, const uint8_t* in49, uint8_t* out96, uint8_t* status);
/* Batched from_bytes(point2&, bytes_view&) for 97-byte input (:146, :187-190 -> ECP2_fromOctet -> ECP2_setx);
 * any tag other than 0x04 and 0x00 is treated as compressed with sign = tag & 1, as the reference does. */
int c12381_g2_decompress_batch(c12381_ctx* ctx, size_t n, const uint8_t* in97, uint8_t* out192, uint8_t* status);
/* device-pointer forms (wire bytes already on the device; the 49- / 97-byte records are read bytewise, no alignment
 * requirement on them; out and status as above) */
int c12381_g1_decompress_batch_dev(c12381_ctx* ctx, size_t n, const uint8_t* in49, uint8_t* out96, uint8_t* status);
int c12381_g2_decompress_batch_dev(c12381_ctx* ctx, size_t n, const uint8_t* in97, uint8_t* out192, uint8_t* status);

/* split pairing and GT arithmetic (what the reference's GTMiller / GTPoint types call) ----------- */
/* pair_ate(fp12&, point2&, point1&) alone (:199 -> 276-279 -> PAIR_ate): the Miller value as FP12_toOctet bytes. */
int c12381_miller_batch(c12381_ctx* ctx, size_t n, const uint8_t* g1_96, const uint8_t* g2_192, uint8_t* out576);
int c12381_miller_batch_dev(c12381_ctx* ctx, size_t n, const uint8_t* g1_96, const uint8_t* g2_192, uint8_t* out576);
/* pair_final_exponentiation(fp12&) alone (:201 -> 281-284 -> PAIR_fexp). */
int c12381_fexp_batch(c12381_ctx* ctx, size_t n, const uint8_t* in576, uint8_t* out576);
int c12381_fexp_batch_dev(c12381_ctx* ctx, size_t n, const uint8_t* in576, uint8_t* out576);
/* op 0: multiply(fp12& r, fp12& v) r*=v (:193 -> 256-259 -> FP12_mul); 1: conjugate (:191 -> 251-254);
 * 2: pow(fp12&, fp12& base, const big&) (:195 -> 261-264 -> FP12_pow; b = 32-byte exponents, used as given,
 *    unitary squarings exactly like the reference: bases outside the cyclotomic subgroup go through the reference's
 *    own digit sequence, members — every pairing value — through a 4-bit windowed ladder that returns the same
 *    bytes; the device holds 224 KB of table per wavefront of 21 elements, at most 1.4 GB);
 * 3: final exponentiation. */
int c12381_gt_op_batch(c12381_ctx* ctx, int op, size_t n, const uint8_t* a576, const uint8_t* b, uint8_t* out576);
int c12381_gt_op_batch_dev(c12381_ctx* ctx, int op, size_t n, const uint8_t* a576, const uint8_t* b, uint8_t* out576);
/* is_unity(fp12&) (:197 -> 271-274 -> FP12_isunity): out[i] = 1 / 0. */
int c12381_gt_is_unity_batch(c12381_ctx* ctx, size_t n, const uint8_t* a576, uint8_t* out);
int c12381_gt_is_unity_batch_dev(c12381_ctx* ctx, size_t n, const uint8_t* a576, uint8_t* out);

/* gt[i] = e(g1[i], Q) with ONE G2 argument for the batch (pair(P_i, g2) against a generator or public key — the
 * shape of every verification equation in the reference's examples): pair_ate + pair_final_exponentiation + to_bytes as
 * c12381_pair_batch, but the line coefficients of Q are computed once and kept until Q changes.  Identical bytes to
 * c12381_pair_batch on n copies of Q for every Q (infinity included); Q not on the twist poisons all outputs. */
int c12381_pair_fixed_g2_batch(c12381_ctx* ctx, size_t n, const uint8_t* g1_96, const uint8_t* g2_192, uint8_t* gt576);
int c12381_pair_fixed_g2_batch_dev(c12381_ctx* ctx, size_t n, const uint8_t* g1_96, const uint8_t* g2_192, uint8_t* gt576);

/* one base for the whole batch ------------------------------------------------------------------ */
/* out[i] = scalars[i] * base: g^x with one g — the reference's most common call shape (the cached default generators,
 * g1_point.hpp:257 / g2_point.hpp:246; setup / key_gen of examples/bbs-plus/src/bbs+.cpp:7-36), still `multiply`
 * (:122 -> 174-177, :152 -> 202-205) per element there.  Here a subgroup base is served from a device-built table of
 * its multiples (32 additions, no doubling; the table is kept in the context until the base changes); any other base
 * goes through the generic kernels, so the result equals c12381_g1_mul_batch / c12381_g2_mul_batch on n copies of the
 * base for every input. */
int c12381_g1_mul_fixed_batch(c12381_ctx* ctx, size_t n, const uint8_t* base96, const uint8_t* scalars32, uint8_t* out, int out_fmt);
int c12381_g1_mul_fixed_batch_dev(c12381_ctx* ctx, size_t n, const uint8_t* base96, const uint8_t* scalars32, uint8_t* out, int out_fmt);
int c12381_g2_mul_fixed_batch(c12381_ctx* ctx, size_t n, const uint8_t* base192, const uint8_t* scalars32, uint8_t* out, int out_fmt);
int c12381_g2_mul_fixed_batch_dev(c12381_ctx* ctx, size_t n, const uint8_t* base192, const uint8_t* scalars32, uint8_t* out, int out_fmt);

/* caller pattern of BASELINE config 5 ------------------------------------------ */
/* ok[j] = [ e(A_j, w + x_j*g2) == e(g1 + r_j*h0 + sum_i m[i*n + j]*h_i, g2) ]: the BBS+ verification equation of the
 * reference's examples/bbs-plus/src/bbs+.cpp:57-73 for n signatures of nmsg message blocks each, evaluated like
 * liner_pair.hpp:339-350 (two Miller loops, ONE final exponentiation, is_unity).  Public parameters are single
 * points (g1 96 B, g2 192 B, h0 96 B, h nmsg x 96 B, w 192 B); per-signature inputs are arrays (A n x 96 B, x and r
 * n x 32 B, m nmsg x n x 32 B message-major).  ok[j] = 1 / 0, 0xff if an input point of lane j is invalid.
 * Message encoding (encode_to<Zp>, zp_number.hpp:1011-1037) and parsing stay on the caller's side. */
int c12381_bbs_plus_verify_batch(c12381_ctx* ctx, size_t n, size_t nmsg, const uint8_t* g1_96, const uint8_t* g2_192,
                                 const uint8_t* h0_96, const uint8_t* h_96, const uint8_t* w_192, const uint8_t* A_96,
                                 const uint8_t* x_32, const uint8_t* r_32, const uint8_t* m_32, uint8_t* ok);
int c12381_bbs_plus_verify_batch_dev(c12381_ctx* ctx, size_t n, size_t nmsg, const uint8_t* g1_96, const uint8_t* g2_192,
                                     const uint8_t* h0_96, const uint8_t* h_96, const uint8_t* w_192, const uint8_t* A_96,
                                     const uint8_t* x_32, const uint8_t* r_32, const uint8_t* m_32, uint8_t* ok);
/* Optional aggregate mode ("random-linear-combination batching ... behind a flag"; the reference has no
 * counterpart, it verifies one signature per call, bbs+.cpp:57-73).  ONE verdict for the batch:
 *   *all_ok = 1  iff g2 and w are elements of G2 and
 *                prod_j [ e(A_j, w) e(x_j A_j - B_j, g2) ]^rho[j] == 1,   B_j = g1 + r_j h0 + sum_i m[i*n + j] h_i,
 * evaluated as inner products mod r, two bucket products over the A_j and ONE product of two pairings.  rho: n x 32 B
 * scalars drawn by the caller, unpredictable to whoever produced the signatures (128 random bits each suffice).
 * If every signature passes c12381_bbs_plus_verify_batch, *all_ok = 1; if one does not, *all_ok = 1 with probability
 * at most 2^-k over rho drawn uniformly from k-bit values (k <= 254).  *all_ok = 0 settles nothing (an invalid signature, or public keys outside G2): run the
 * per-signature entry then.  n = 0 gives 1.  At most 2^26 - nmsg - 2 signatures per call.  The _dev form writes one
 * byte. */
int c12381_bbs_plus_verify_aggregate(c12381_ctx* ctx, size_t n, size_t nmsg, const uint8_t* g1_96, const uint8_t* g2_192,
                                     const uint8_t* h0_96, const uint8_t* h_96, const uint8_t* w_192, const uint8_t* A_96,
                                     const uint8_t* x_32, const uint8_t* r_32, const uint8_t* m_32, const uint8_t* rho_32, int* all_ok);
int c12381_bbs_plus_verify_aggregate_dev(c12381_ctx* ctx, size_t n, size_t nmsg, const uint8_t* g1_96, const uint8_t* g2_192,
                                         const uint8_t* h0_96, const uint8_t* h_96, const uint8_t* w_192, const uint8_t* A_96,
                                         const uint8_t* x_32, const uint8_t* r_32, const uint8_t* m_32, const uint8_t* rho_32,
                                         uint8_t* all_ok);

/* hash-to-G1 ------------------------------------------------------------------- */
/* G1Point::from_hash (include/crypto12381/g1_point.hpp:219-234) from the 64-byte SHA3-512 digest on: the digest as a
 * big-endian integer mod p (fixed_time_mod :55 -> 94-97), residue (:110 -> 149-152 -> FP_nres), map_to_point
 * (:113 -> 154-157 -> ECP_map2point: simplified SWU + 11-isogeny), multiply_cofactor (:116 -> 159-162 -> ECP_cfp),
 * encoded like to_bytes.  Hashing the caller's serialisation (hash_state, set.hpp:317-392) stays on the host.
 * out_fmt 49 or 96.  Not RFC 9380 hash_to_curve: one field element per digest, as the reference defines it. */
int c12381_g1_from_hash_batch(c12381_ctx* ctx, size_t n, const uint8_t* digests64, uint8_t* out, int out_fmt);
int c12381_g1_from_hash_batch_dev(c12381_ctx* ctx, size_t n, const uint8_t* digests64, uint8_t* out, int out_fmt);
/* map_to_point(point1&, const fp&) alone (:113 -> 154-157 -> ECP_map2point): 48-byte field elements (taken mod p, as
 * residue/FP_nres does) -> 96-byte affine points of E, NOT yet multiplied by the cofactor. */
int c12381_g1_map_to_point_batch(c12381_ctx* ctx, size_t n, const uint8_t* u48, uint8_t* out96);
/* multiply_cofactor(point1&) alone (:116 -> 159-162 -> ECP_cfp): P -> [1 - x]P = [0xd201000000010001]P as a plain
 * multiple.  (Not c12381_g1_mul_batch: `multiply` is PAIR_G1mul, whose GLV evaluation differs from the plain
 * multiple for points outside the subgroup — which is what map_to_point produces.) */
int c12381_g1_clear_cofactor_batch(c12381_ctx* ctx, size_t n, const uint8_t* in96, uint8_t* out96);

/* scalar-field (Zp) batch helpers ----------------------------------------------- */
/* Element-wise arithmetic mod r on canonical 32-byte big-endian values (inputs < 2^256 are reduced first); what
 * zp_number.hpp evaluates per element through multiply(big2&,..) :47 + mod :63 (operator*, :295-380), mod_negate :57,
 * mod_inverse :59 -> 84-87 -> BIG_invmodp (inverse(), zp_number.hpp:420-425; the inverse of 0 is 0).
 * op 0: a*b, 1: a+b, 2: a-b, 3: -a, 4: 1/a (b ignored for 3 and 4, may be NULL). */
int c12381_zp_op_batch(c12381_ctx* ctx, int op, size_t n, const uint8_t* a32, const uint8_t* b32, uint8_t* out32);
int c12381_zp_op_batch_dev(c12381_ctx* ctx, int op, size_t n, const uint8_t* a32, const uint8_t* b32, uint8_t* out32);
/* Zp from_hash (zp_number.hpp:540-548): 64-byte digest as a big-endian integer mod r (fixed_time_mod). */
int c12381_zp_from_hash_batch(c12381_ctx* ctx, size_t n, const uint8_t* digests64, uint8_t* out32);
/* sum over i of a[i]*b[i] mod r (b == NULL: sum of a[i]) — sum()/inner products of zp_number.hpp:549-615; n = 0 -> 0. */
int c12381_zp_inner_product(c12381_ctx* ctx, size_t n, const uint8_t* a32, const uint8_t* b32, uint8_t out32[32]);
int c12381_zp_inner_product_dev(c12381_ctx* ctx, size_t n, const uint8_t* a32, const uint8_t* b32, uint8_t* out32);

/* BBS+ verification from the WIRE formats of the reference's example (bbs+.cpp:57-73 as a whole): pp.g1_g2_h0 = serialize(g1, g2, h0)
 * (49 + 97 + 49 bytes, set.hpp:235-293), pp.h (nh entries of 49 bytes), pk = serialize(w) (97 bytes), signatures =
 * serialize(A, x, r) (49 + 48 + 48 bytes) and raw message bytes — msg_len bytes per message, encoded by encode_to<Zp>
 * (zp_number.hpp:1011-1037) into ceil(msg_len / 31) scalars.  Decoding (ECP_fromOctet / ECP2_fromOctet with the header layer's
 * "leading 0x00 = infinity", g1_point.hpp:87-111; parse<Zp> range check zp_number.hpp:226-236), encoding and verification run
 * on the context's stream.  ok[j] = 1 / 0 = what verify() returns, 0xff = the reference would throw for signature j (malformed
 * A, x or r); public material that does not decode poisons every lane and returns C12381_E_POINT; more message units than
 * h entries ("message is too long") is C12381_E_ARG. */
int c12381_bbs_plus_verify_wire_batch(c12381_ctx* ctx, size_t n, size_t nh, size_t msg_len, const uint8_t* g1_g2_h0_195, const uint8_t* h_49,
                                      const uint8_t* pk_97, const uint8_t* sig_145, const uint8_t* msgs, uint8_t* ok);
int c12381_bbs_plus_verify_wire_batch_dev(c12381_ctx* ctx, size_t n, size_t nh, size_t msg_len, const uint8_t* g1_g2_h0_195, const uint8_t* h_49,
                                          const uint8_t* pk_97, const uint8_t* sig_145, const uint8_t* msgs, uint8_t* ok);
/* BBS+ signing for a batch (examples/bbs-plus/src/bbs+.cpp:38-55): A[j] = (g1 * h0^r[j] * prod_i h_i^m[i*n + j])^(1/(gamma + x[j]))
 * — `^` is multiply (:122), `inverse` is mod_inverse (:59; inverse(0) = 0, so A is then the point at infinity), Π the
 * sum of the columns.  x and r are the caller's random scalars (the reference draws them inside sign()); gamma is the
 * secret key (32 B).  Output 96-byte affine points, message-major m as in the verification entry. */
int c12381_bbs_plus_sign_batch(c12381_ctx* ctx, size_t n, size_t nmsg, const uint8_t* g1_96, const uint8_t* h0_96, const uint8_t* h_96,
                               const uint8_t* gamma_32, const uint8_t* x_32, const uint8_t* r_32, const uint8_t* m_32, uint8_t* A_out96);
int c12381_bbs_plus_sign_batch_dev(c12381_ctx* ctx, size_t n, size_t nmsg, const uint8_t* g1_96, const uint8_t* h0_96, const uint8_t* h_96,
                                   const uint8_t* gamma_32, const uint8_t* x_32, const uint8_t* r_32, const uint8_t* m_32, uint8_t* A_out96);

#ifdef __cplusplus
}
#endif
#endif /* C12381_HIP_H */
