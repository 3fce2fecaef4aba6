// Kernel entry points of the backend, one translation unit per family so the families compile in parallel:
//   k_g1.hip     Fp hooks, G1 scalar multiplication / addition / finish / tree sum, MSM stages, G1 decompression
//   k_g2gt.hip   G2 scalar multiplication (one lane per point) / addition / finish / decompression, one-lane pairing kernels, GT arithmetic
//   k_g2h.hip    G2 scalar multiplication with two lanes per point (half an Fp2 element per lane)
//   k_pair3.hip  three-lanes-per-pairing Miller loop + final exponentiation
//   k_hash_zp.hip  hash-to-G1 and the scalar-field (Zp) helpers
//   k_fixed.hip  fixed-base tables and their evaluation (public-parameter columns of BBS+)
// c12381_hip.hip (context, workspaces, C ABI) launches them.  Every kernel is built for 2 waves per SIMD
// (__launch_bounds__(BLOCK, 2)): the field routines are not inlined and get the full 256-VGPR budget.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>

namespace c12381 {

constexpr int BLOCK = 256;
constexpr int FINISH_M = 16;                     // most elements per lane in the simultaneous inversion (c12381_hip.hip finish_lanes)
constexpr int TRI_PER_WAVE = 21;                 // pairings per 64-lane wavefront in the three-lane kernels (lane 63 idles along)
// Header words in front of a device-built table (fixed-base multiples, line coefficients) and of a gate buffer
constexpr int HDR_VALID = 48;                    // 1 = table usable / this path runs; kernels of the other path return at once
constexpr int HDR_REBUILD = 49;                  // set by fixed_cache_check_kernel when the cached point differs
constexpr int HDR_MAGIC = 50;                    // the header has been written before
constexpr int HDR_RULE = 51;                     // validity rule the flag was computed under (line tables)
constexpr int HDR_DWORDS = 64;                   // table data starts here
int set_queue_groups_override(int v);             // k_pair3.hip: tuning switch (C12381_QUEUE_GROUPS)
constexpr int GATE_OTHER = 49;                   // (gate + GATE_OTHER)[HDR_VALID] = the complement of gate[HDR_VALID]

// Waves per SIMD of the three scalar-multiplication kernels (register budget 512 / occupancy: 256, 168 or 128 VGPRs).  Three per SIMD were
// measured twice and lose: the spills cost more than the third wavefront fills (profiles/r03_ab_occupancy.txt, r04_ab_occupancy3.txt).
constexpr int G1_OCC = 2, MSM_OCC = 2, G2H_OCC = 2;
__global__ void __launch_bounds__(BLOCK, 2) fp_op_kernel(int op, size_t n, const uint8_t* a, const uint8_t* b, uint8_t* out);
__global__ void __launch_bounds__(BLOCK, 2) fp_mulchain_kernel(size_t n, int iters, const uint8_t* a, const uint8_t* b, uint8_t* out);
__global__ void __launch_bounds__(BLOCK, G1_OCC) g1_mul_kernel(size_t n, const uint8_t* pts, size_t pt_stride, const uint8_t* scalars, int32_t* tab, int32_t* proj, size_t proj_stride, size_t proj_off, int* bad_flag, const int32_t* skip_if, int small_term);
__global__ void __launch_bounds__(BLOCK, 2) g1_rsub_kernel(size_t n, int32_t* acc, size_t acc_stride, const int32_t* other, size_t other_stride, size_t other_off, const int32_t* run_if);
__global__ void __launch_bounds__(BLOCK, 2) g1_add_kernel(size_t n, const uint8_t* a, const uint8_t* b, int32_t* proj, size_t proj_stride, int* bad_flag);
__global__ void __launch_bounds__(BLOCK, 2) g1_finish_kernel(size_t n, const int32_t* proj, size_t stride, int32_t* pref, uint8_t* out, int fmt, size_t T);
__global__ void __launch_bounds__(BLOCK, 2) g1_lift_kernel(size_t n, const uint8_t* pts, int32_t* proj, size_t stride, int* bad_flag);
__global__ void __launch_bounds__(BLOCK, 2) g1_reduce_kernel(size_t n, const int32_t* in, size_t in_stride, size_t m, int32_t* outp, size_t out_stride);
__global__ void __launch_bounds__(BLOCK, 2) g1_mul_plain_kernel(size_t n, const uint8_t* pts, const uint8_t* scalars, int32_t* proj, size_t proj_stride, int* bad_flag);
__global__ void __launch_bounds__(BLOCK, 2) g1_add_const_kernel(size_t n, int32_t* proj, size_t stride, const uint8_t* pt96, int* bad_flag);
__global__ void __launch_bounds__(BLOCK, 2) msm_prep_kernel(size_t n, const uint8_t* pts, int in_fmt, const uint8_t* scalars, int c, int W, int32_t* pts2, uint32_t* keys, uint32_t* vals, int* bad_flag);
__global__ void __launch_bounds__(BLOCK, 2) msm_ranges_kernel(size_t E, const uint32_t* keys, int c, int W, uint32_t* lo, uint32_t* hi);
__global__ void __launch_bounds__(BLOCK, 2) msm_prep16_kernel(size_t n, const uint8_t* pts, int in_fmt, const uint8_t* scalars, int c, int W, int32_t* pts2, uint16_t* keys, int* bad_flag);
__global__ void __launch_bounds__(BLOCK, 2) msm_ranges16_kernel(size_t n, const uint16_t* keys, int c, int W, uint32_t* lo, uint32_t* hi);
constexpr int MSM_RANGES_PER_THREAD = 8;           // entries per thread of msm_ranges16_kernel (the host sizes its grid with it)
__global__ void __launch_bounds__(BLOCK, MSM_OCC) msm_bucket_kernel(size_t nbk, const uint32_t* lo, const uint32_t* hi, const uint32_t* vals, const int32_t* pts2, int32_t* bk, const uint32_t* order, uint32_t cap, uint32_t early_max);
__global__ void __launch_bounds__(BLOCK, 2) msm_sizes_kernel(size_t nbk, const uint32_t* lo, const uint32_t* hi, uint32_t* key, uint32_t* ident, uint32_t cap, uint32_t* cnt, uint2* seg, uint4* big, uint32_t early_max);
__global__ void __launch_bounds__(BLOCK, 2) msm_overflow_kernel(const uint32_t* cnt, const uint2* seg, const uint32_t* lo, const uint32_t* hi, const uint32_t* vals, const int32_t* pts2, int32_t* part, uint32_t cap);
__global__ void __launch_bounds__(BLOCK, 2) msm_overflow_combine_kernel(const uint32_t* cnt, const uint4* big, const int32_t* part, int32_t* bk);
__global__ void __launch_bounds__(BLOCK, 2) msm_wreduce_kernel(int W, uint32_t nb, uint32_t chunks, const int32_t* bk, int32_t* out, size_t out_stride);
__global__ void __launch_bounds__(BLOCK, 2) g1_decompress_kernel(size_t n, const uint8_t* in, uint8_t* out, uint8_t* status, int mark_invalid);
__global__ void __launch_bounds__(64, 1) msm_horner_kernel(const int32_t* rw, size_t stride, int W, int c, int32_t* out, size_t out_stride, const int32_t* term_in);
__global__ void __launch_bounds__(BLOCK, 2) g1_wave_reduce_kernel(size_t groups, int W, const int32_t* in, size_t in_stride, int32_t* outp, size_t out_stride);
__global__ void __launch_bounds__(64, 1) msm_small_term_kernel(const int32_t* sbucket, int32_t* term_out, const uint32_t* done);
__global__ void __launch_bounds__(64, 1) msm_small_early_kernel(const uint32_t* lo, const uint32_t* hi, uint32_t small_bucket, uint32_t early_max, const uint32_t* vals, const int32_t* pts2, int32_t* term_out, uint32_t* done);
constexpr uint32_t MSM_SMALL_EARLY_MAX = 4096;     // longest small-scalar bucket msm_small_early_kernel takes (64 additions per lane)
__global__ void __launch_bounds__(BLOCK, 2) g2_mul_kernel(size_t n, const uint8_t* pts, size_t pt_stride, const uint8_t* scalars, int32_t* tab, size_t tab_stride, uint8_t* out, int fmt, int* bad_flag, const int32_t* skip_if, int32_t* proj, size_t proj_stride, size_t proj_off, int in_g2);
__global__ void __launch_bounds__(BLOCK, G2H_OCC) g2_mul2_kernel(size_t n, const uint8_t* pts, size_t pt_stride, const uint8_t* scalars, int32_t* tab, int* bad_flag, const int32_t* skip_if, int32_t* proj, size_t proj_stride, size_t proj_off, int in_g2);
__global__ void __launch_bounds__(BLOCK, 2) g2_finish_kernel(size_t n, const int32_t* proj, size_t stride, int32_t* pref, uint8_t* out, int fmt, size_t T);
__global__ void __launch_bounds__(BLOCK, 2) g2_lift_kernel(size_t n, const uint8_t* pts, int32_t* proj, size_t stride, int* bad_flag);
__global__ void __launch_bounds__(BLOCK, 2) g2_reduce_kernel(size_t n, const int32_t* in, size_t in_stride, size_t m, int32_t* outp, size_t out_stride);
__global__ void __launch_bounds__(BLOCK, 2) g2_add_kernel(size_t n, const uint8_t* a, size_t a_stride, const uint8_t* b, uint8_t* out, int fmt, int* bad_flag, const int32_t* skip_if);
#ifdef C12381_EXPERIMENTS      // the superseded one-lane pairing kernels (k_g2gt.hip): experiments builds only
__global__ void __launch_bounds__(BLOCK, 2) clock_probe_kernel(unsigned long long* out, int n, int gap);
__global__ void __launch_bounds__(BLOCK, 2) pair_kernel(size_t n, const uint8_t* g1, const uint8_t* g2, uint8_t* gt, int* bad_flag);
__global__ void __launch_bounds__(BLOCK, 2) pair_eq_kernel(size_t n, const uint8_t* a1, const uint8_t* a2, const uint8_t* b1, const uint8_t* b2, size_t b2_stride, uint8_t* out, int* bad_flag);
#endif
__global__ void __launch_bounds__(BLOCK, 2) g2_decompress_kernel(size_t n, const uint8_t* in, uint8_t* out, uint8_t* status, int mark_invalid);
#ifdef C12381_EXPERIMENTS
__global__ void __launch_bounds__(BLOCK, 2) miller_kernel(size_t n, const uint8_t* g1, const uint8_t* g2, uint8_t* out, int* bad_flag);
__global__ void __launch_bounds__(BLOCK, 2) gt_op_kernel(int op, size_t n, const uint8_t* a, const uint8_t* b, uint8_t* out);
__global__ void __launch_bounds__(BLOCK, 2) gt_is_unity_kernel(size_t n, const uint8_t* a, uint8_t* out);
#endif
__global__ void __launch_bounds__(BLOCK, 2) pair3_kernel(size_t n, const uint8_t* g1, const uint8_t* g2, uint8_t* gt, int* bad_flag);
__global__ void __launch_bounds__(BLOCK, 2) pair3_prod_kernel(size_t n, int k, const uint8_t* g1, const uint8_t* g2, uint8_t* gt, int* bad_flag, int miller_only);
__global__ void __launch_bounds__(BLOCK, 2) pair3_eq_kernel(size_t n, const uint8_t* a1, const uint8_t* a2, const uint8_t* b1, const uint8_t* b2, size_t b2_stride, uint8_t* out, int* bad_flag, const int32_t* skip_if);
__global__ void __launch_bounds__(BLOCK, 2) miller3_queue_kernel(size_t n, const uint8_t* g1, const uint8_t* g2, uint8_t* out, int* bad_flag, uint4* state, unsigned int* flags, unsigned int* counter, int spin_limit, unsigned int epoch, unsigned long long* wstats);
__global__ void __launch_bounds__(BLOCK, 2) fexp3_queue_kernel(size_t n, const uint8_t* in576, uint8_t* out, int* bad_flag, uint4* state, unsigned int* flags, unsigned int* counter, int spin_limit, unsigned int epoch, unsigned long long* wstats);
__global__ void __launch_bounds__(BLOCK, 2) pair3_queue_kernel(size_t n, const uint8_t* g1, const uint8_t* g2, uint8_t* gt, int* bad_flag, uint4* state, unsigned int* flags, unsigned int* counter, int spin_limit, unsigned int epoch, unsigned long long* stamps, unsigned long long* wstats);
__global__ void __launch_bounds__(BLOCK, 2) pair3_eq_queue_kernel(size_t n, const uint8_t* a1, const uint8_t* a2, const uint8_t* b1, const uint8_t* b2, size_t b2_stride, uint8_t* out, int* bad_flag, uint4* state, unsigned int* flags, unsigned int* counter, const int32_t* skip_if, int spin_limit, unsigned int epoch);
__global__ void __launch_bounds__(BLOCK, 2) g2_lines_table_kernel(const uint8_t* q192, int32_t* buf, int need_g2);
__global__ void __launch_bounds__(BLOCK, 2) gate_and_kernel(int32_t* gate, const int32_t* a, const int32_t* b);
__global__ void __launch_bounds__(BLOCK, 2) pair3_prod_fixed_queue_kernel(size_t n, const uint8_t* a96, const uint8_t* c96, const int32_t* tabw, const int32_t* tabg, uint8_t* out, int* bad_flag, uint4* state, unsigned int* flags, unsigned int* counter, const int32_t* run_if, int spin_limit, unsigned int epoch);
__global__ void __launch_bounds__(BLOCK, 2) pair3_fixed_queue_kernel(size_t n, const uint8_t* g1_96, const int32_t* buf, uint8_t* gt, int* bad_flag, uint4* state, unsigned int* flags, unsigned int* counter, int spin_limit, unsigned int epoch);
__global__ void __launch_bounds__(BLOCK, 2) miller3_kernel(size_t n, const uint8_t* g1, const uint8_t* g2, uint8_t* out, int* bad_flag);
__global__ void __launch_bounds__(BLOCK, 2) gt3_op_kernel(int op, size_t n, const uint8_t* a, const uint8_t* b, uint8_t* out, uint4* pow_tab);
__global__ void __launch_bounds__(BLOCK, 2) gt3_pow_queue_kernel(size_t n, const uint8_t* a, const uint8_t* b, uint8_t* out, int* bad_flag, uint4* pow_tab, uint4* state, unsigned int* flags, unsigned int* counter, int spin_limit);
constexpr size_t GT_POW_TAB_BYTES_PER_WAVE = (size_t)16 * 14 * 64 * 16;   // 16 entries x 14 rows (one Fp4 per lane) x 64 lanes x 16 bytes
__global__ void __launch_bounds__(BLOCK, 2) gt3_is_unity_kernel(size_t n, const uint8_t* a, uint8_t* out);
constexpr int PAIR_QUEUE_STATE_ROWS = 42;        // 16-byte rows x 64 lanes per group of 21 pairings (F, tc1, tc2, y1): the fenced form (GT power queue)
constexpr size_t PAIR_QUEUE_STATE_BYTES = (size_t)168 * 64 * 8;   // a queued group's block: 168 32-bit words per lane, each in an 8-byte tagged word (k_pair3.hip stw_store)
__global__ void __launch_bounds__(BLOCK, 2) g1_from_hash_kernel(size_t n, const uint8_t* in, int mode, int32_t* proj, size_t proj_stride, int* bad_flag);
__global__ void __launch_bounds__(BLOCK, 2) zp_op_kernel(int op, size_t n, const uint8_t* a, const uint8_t* b, uint8_t* out);
__global__ void __launch_bounds__(BLOCK, 2) zp_fold_cols_kernel(size_t n, const uint8_t* a, size_t a_col_stride, const uint8_t* r32, const uint8_t* m32, int first, size_t T, uint8_t* out);
__global__ void __launch_bounds__(BLOCK, 2) bbs_wire_pub_kernel(size_t nblk, const uint8_t* g1_g2_h0, const uint8_t* h49, const uint8_t* pk97, uint8_t* g1s49, uint8_t* g2s97);
__global__ void __launch_bounds__(BLOCK, 2) bbs_wire_prep_kernel(size_t n, size_t msg_len, size_t nblk, const uint8_t* sig145, const uint8_t* msgs, uint8_t* a49, uint8_t* x32, uint8_t* r32, uint8_t* m32, uint8_t* status);
__global__ void __launch_bounds__(BLOCK, 2) bbs_wire_finish_kernel(size_t n, size_t npub1, const uint8_t* st_sig, const uint8_t* st_a, const uint8_t* st_pub1, const uint8_t* st_pub2, uint8_t* ok, int* bad_flag);
constexpr int ZP_INV_RUN = 16;
__global__ void __launch_bounds__(BLOCK, 2) zp_batch_inv_kernel(size_t n, size_t T, const uint8_t* x, const uint8_t* gamma, uint8_t* out, uint32_t* pref);
__global__ void __launch_bounds__(BLOCK, 2) zp_from_hash_kernel(size_t n, const uint8_t* digests, uint8_t* out);
__global__ void __launch_bounds__(BLOCK, 2) zp_fold_kernel(size_t n, const uint8_t* a, const uint8_t* b, size_t T, uint8_t* out);
__global__ void __launch_bounds__(64, 1) fixed_cache_check_kernel(const uint8_t* base, int nbytes, int32_t* header);
__global__ void __launch_bounds__(BLOCK, 2) g1_fixed_table_kernel(const uint8_t* base96, int32_t* buf);
__global__ void __launch_bounds__(BLOCK, 2) g1_fixed_eval_kernel(size_t n, const int32_t* buf, const uint8_t* scalars, int32_t* proj, size_t proj_stride, size_t proj_off);
__global__ void __launch_bounds__(BLOCK, 2) g2_fixed_table_kernel(const uint8_t* base192, int32_t* buf);
__global__ void __launch_bounds__(BLOCK, 2) g2_fixed_eval_kernel(size_t n, const int32_t* buf, const uint8_t* scalars, const uint8_t* addend192, uint8_t* out, int fmt, int* bad_flag, int32_t* proj, size_t proj_stride);

}  // namespace c12381
