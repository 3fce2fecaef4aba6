// Three-lanes-per-pairing kernels (pairing3.hpp): each triple of lanes holds one Fp12 (one Fp4 coefficient per lane)
// and one G2 point (one projective coordinate per lane); 21 pairings per wavefront.
//   pair3_kernel       Miller loop + final exponentiation -> 576-byte GT
//   pair3_eq_kernel    e(a1,a2) == e(b1,b2)
#include "kernels_common.hpp"
#include "pairing3.hpp"

using namespace c12381;

namespace {

__device__ __forceinline__ void tri_setup(tri& t, size_t& idx, bool& active, size_t n) {
    const unsigned lane = threadIdx.x & 63u;
    const size_t wave = ((size_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
    const unsigned trip = lane / 3u;
    t.role = lane == 63u ? 0 : (int)(lane - 3u * trip);
    t.base = lane == 63u ? 63 : (int)(3u * trip);
    const size_t i = wave * TRI_PER_WAVE + (lane == 63u ? TRI_PER_WAVE - 1 : trip);
    active = lane < 63u && i < n;
    idx = i < n ? i : n - 1;                    // inactive lanes shadow the last element: same instruction stream
}
__device__ __forceinline__ void gt_store_coeff(uint8_t* o576, const fp4& x, int role) {
    uint8_t* o = o576 + (role == 0 ? 384 : (role == 1 ? 192 : 0));      // FP12_toOctet: c | b | a
    uint32_t raw[12];
    fp_to_raw48(raw, x.b.b); store_raw48(o, raw);
    fp_to_raw48(raw, x.b.a); store_raw48(o + 48, raw);
    fp_to_raw48(raw, x.a.b); store_raw48(o + 96, raw);
    fp_to_raw48(raw, x.a.a); store_raw48(o + 144, raw);
}

// One Fp4 per lane in LDS (240-byte stride: 16-byte aligned, lanes spread over the banks): the running Miller value and
// the accumulator of the exponentiations by x live here, so the out-of-line tower routines read and write their hot
// operand at LDS latency.  60 KB per workgroup, two workgroups per CU.
struct alignas(16) fp4_slot { fp4 v; int32_t pad[4]; };

}  // namespace

namespace c12381 {

__global__ void __launch_bounds__(BLOCK, 2) pair3_kernel(size_t n, const uint8_t* g1, const uint8_t* g2, uint8_t* gt, int* bad_flag) {
    if ((((size_t)blockIdx.x * BLOCK + threadIdx.x) >> 6) * TRI_PER_WAVE >= n) return;      // whole wavefront idle
    tri t; size_t i; bool active;
    tri_setup(t, i, active, n);
    fp px, py; fp2 qx, qy; bool pinf, qinf, ok;
    pair_inputs(px, py, pinf, qx, qy, qinf, ok, g1 + 96 * i, g2 + 192 * i);
    if (!ok) { if (active) *bad_flag = 1; pinf = true; qinf = true; }
    __shared__ fp4_slot slots[BLOCK];
    fp4& H = slots[threadIdx.x].v;
    miller3_loop(H, px, py, pinf, qx, qy, qinf, t);
    fp4 F = H;
    f12t_final_exp_ws(F, H, t);
    if (active) {
        if (!ok) { uint4* q = reinterpret_cast<uint4*>(gt + 576 * i + (t.role == 0 ? 384 : (t.role == 1 ? 192 : 0))); for (int j = 0; j < 12; ++j) q[j] = make_uint4(~0u, ~0u, ~0u, ~0u); }
        else gt_store_coeff(gt + 576 * i, F, t.role);
    }
}

__global__ void __launch_bounds__(BLOCK, 2) pair3_eq_kernel(size_t n, const uint8_t* a1, const uint8_t* a2, const uint8_t* b1, const uint8_t* b2,
                                                         size_t b2_stride, uint8_t* out, int* bad_flag) {
    if ((((size_t)blockIdx.x * BLOCK + threadIdx.x) >> 6) * TRI_PER_WAVE >= n) return;
    tri t; size_t i; bool active;
    tri_setup(t, i, active, n);
    // e(a1, a2) == e(b1, b2)  <=>  e(a1, a2) * e(-b1, b2) == 1: one joint Miller loop (shared squarings), one final
    // exponentiation.  liner_pair.hpp:339-350 forms ate(a) * conj(ate(b)) from two separate loops; after the final
    // exponentiation both are e(a) / e(b) (conj and negating the G1 argument both invert the pairing value; the
    // Miller values differ by factors in Fp6, which the easy part kills), so the boolean is the same for all
    // curve points, infinity arguments included.
    fp px, py, px2, py2; fp2 qx, qy, qx2, qy2; bool pinf, qinf, pinf2, qinf2, ok, okb;
    __shared__ fp4_slot slots[BLOCK];
    fp4& H = slots[threadIdx.x].v;
    pair_inputs(px, py, pinf, qx, qy, qinf, ok, a1 + 96 * i, a2 + 192 * i);
    if (!ok) { pinf = true; qinf = true; }
    pair_inputs(px2, py2, pinf2, qx2, qy2, qinf2, okb, b1 + 96 * i, b2 + b2_stride * i);
    if (!okb) { pinf2 = true; qinf2 = true; }
    {
        fp ny;
        fp_neg(ny, py2);
        fp_norm1(py2, ny);
    }
    miller3_loop2(H, px, py, pinf, qx, qy, qinf, px2, py2, pinf2, qx2, qy2, qinf2, t);
    fp4 F = H;
    f12t_final_exp_ws(F, H, t);
    const bool one = f12t_is_one(F, t);
    const bool valid = ok && okb;
    if (active && t.role == 0) {
        if (!valid) *bad_flag = 1;
        out[i] = valid ? (one ? 1 : 0) : 0xff;
    }
}

}  // namespace c12381
