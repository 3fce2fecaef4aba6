// G2 scalar multiplication with TWO lanes per point (fp2h.hpp / g2h.hpp): lane 2j holds the real parts and lane 2j+1 the
// imaginary parts of the coordinates of element j, so the running point, the table entry being added and the
// temporaries of the complete addition fit the 256-register budget that the one-lane kernel (k_g2gt.hip, 4.7 KB of
// private memory per lane) overflows.  Same arithmetic, same table of multiples, same treatment of points outside G2
// (g2_scalar_mul of g2.hpp is instantiated for both element types); results go to g2_finish_kernel as projective SoA.
#include "kernels_common.hpp"
#include "g2h.hpp"

using namespace c12381;

namespace c12381 {

__global__ void __launch_bounds__(BLOCK, G2H_OCC) g2_mul2_kernel(size_t n, const uint8_t* pts, size_t pt_stride, const uint8_t* scalars, int32_t* tab,
                                                        int* bad_flag, const int32_t* skip_if, int32_t* proj, size_t proj_stride, size_t proj_off, int in_g2) {
    if (skip_if && skip_if[HDR_VALID] != 0) return;          // served by a valid fixed-base table (k_fixed.hip)
    const size_t lane = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    const size_t i = lane >> 1;                              // both lanes of a pair take every branch together
    if (i >= n) return;
    fp2 qx, qy;
    bool inf, ok;
    g2_parse_any(qx, qy, inf, ok, pts, pt_stride, i);
    fp2h hx, hy;
    fp2h_from(hx, qx); fp2h_from(hy, qy);
    uint32_t raw[8], k[8];
    load_raw32(raw, scalars + 32 * i);
    scalar_from_raw32(k, raw);
    g2hp acc;
    g2_scalar_mul(acc, hx, hy, inf || !ok, k, tab + lane * (size_t)G2H_TAB_DWORDS, in_g2 != 0);
    if (!ok) *bad_flag = 1;
    g2h_store_proj(proj, proj_stride, proj_off + i, acc, !ok);
}

}  // namespace c12381
