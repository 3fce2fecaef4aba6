/* TEST INFRASTRUCTURE ONLY — see c12381_oracle.c.  Never included by the product. */
#ifndef C12381_ORACLE_H
#define C12381_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* All buffers are canonical big-endian bytes:
 *   Fp 48 B | scalar 32 B (any value < 2^256, reduced mod r) | G1 96 B x‖y (zeros = infinity) / 49 B
 *   G2 192 B x.b‖x.a‖y.b‖y.a (zeros = infinity) / 97 B | GT 576 B (c‖b‖a, each Fp4 b‖a, each Fp2 b‖a) */
int orc_g1_generator(uint8_t out[96]);
int orc_g2_generator(uint8_t out[192]);
int orc_fp_op_batch(int op, size_t n, const uint8_t* a, const uint8_t* b, uint8_t* out, uint8_t* ok);
int orc_g1_mul_batch(size_t n, const uint8_t* pts96, const uint8_t* scalars32, uint8_t* out, int out_fmt, int nthreads);
int orc_g1_add_batch(size_t n, const uint8_t* a96, const uint8_t* b96, uint8_t* out, int out_fmt);
int orc_g1_decompress_batch(size_t n, const uint8_t* in49, uint8_t* out96, uint8_t* status);
int orc_g1_compress_batch(size_t n, const uint8_t* in96, uint8_t* out49);
int orc_g1_msm(size_t n, const uint8_t* pts96, const uint8_t* scalars32, uint8_t* out, int out_fmt, int nthreads);
int orc_g2_mul_batch(size_t n, const uint8_t* pts192, const uint8_t* scalars32, uint8_t* out, int out_fmt, int nthreads);
int orc_g2_add_batch(size_t n, const uint8_t* a192, const uint8_t* b192, uint8_t* out, int out_fmt);
int orc_g2_decompress_batch(size_t n, const uint8_t* in97, uint8_t* out192, uint8_t* status);
int orc_g2_compress_batch(size_t n, const uint8_t* in192, uint8_t* out97);
int orc_pair_batch(size_t n, const uint8_t* g1_96, const uint8_t* g2_192, uint8_t* gt576, int nthreads);
int orc_miller_batch(size_t n, const uint8_t* g1_96, const uint8_t* g2_192, uint8_t* out576);
int orc_fexp_batch(size_t n, const uint8_t* in576, uint8_t* out576);
int orc_pair_eq_batch(size_t n, const uint8_t* a1, const uint8_t* a2, const uint8_t* b1, const uint8_t* b2, uint8_t* ok, int nthreads);
int orc_pair2_batch(size_t n, const uint8_t* a1, const uint8_t* a2, const uint8_t* b1, const uint8_t* b2, uint8_t* gt576);
int orc_gt_op_batch(int op, size_t n, const uint8_t* a576, const uint8_t* b, uint8_t* out576);
int orc_g1_from_hash_batch(size_t n, const uint8_t* digests64, uint8_t* out, int out_fmt);
int orc_zp_op_batch(int op, size_t n, const uint8_t* a32, const uint8_t* b32, uint8_t* out32);
int orc_zp_from_hash_batch(size_t n, const uint8_t* digests64, uint8_t* out32);

#ifdef __cplusplus
}
#endif
#endif
