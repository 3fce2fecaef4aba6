// Fixed-base columns (fixed_base.hpp): table construction with a device-side "same base as last time?" check, and
// the table-driven evaluation kernels.  Table buffer = FB_HEADER_DWORDS header (cached base bytes, state) + entries.
//   header[0..47]  the base's canonical bytes as dwords (96 B for G1, 192 B for G2)
//   header[HDR_VALID]    ok: 1 = table valid (base on the curve, not infinity, in the order-r subgroup)
//   header[HDR_REBUILD]  build: 1 = the table kernel must (re)build, 0 = cached table matches the base
#include "kernels_common.hpp"
#include "fixed_base.hpp"

using namespace c12381;

namespace c12381 {

// one wavefront: compare the base with the cached copy; on a miss store the new copy and request a rebuild
__global__ void __launch_bounds__(64, 1) fixed_cache_check_kernel(const uint8_t* base, int nbytes, int32_t* header) {
    const int lane = threadIdx.x;
    const int nd = nbytes / 4;
    const uint32_t* b = reinterpret_cast<const uint32_t*>(base);
    uint32_t mine = 0, cached = 0;
    if (lane < nd) { mine = b[lane]; cached = (uint32_t)header[lane]; }
    const bool same = __all(mine == cached) && header[HDR_MAGIC] == 0x46423031;      // magic: the header has been written before
    if (lane < nd) header[lane] = (int32_t)mine;
    if (lane == 0) { header[HDR_REBUILD] = same ? 0 : 1; header[HDR_MAGIC] = 0x46423031; if (!same) header[HDR_VALID] = 0; }
}

__global__ void __launch_bounds__(BLOCK, 2) g1_fixed_table_kernel(const uint8_t* base96, int32_t* buf) {
    int32_t* header = buf;
    if (header[HDR_REBUILD] == 0) return;                                   // cached table is current
    const size_t L = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (L >= (size_t)FB_G1_WINDOWS * FB_ENTRIES) return;
    g1p base;
    bool inf, ok;
    g1_parse96(base.x, base.y, inf, ok, base96);
    fp_one(base.z);
    if (L == 0) header[HDR_VALID] = (ok && !inf && g1_in_subgroup(base)) ? 1 : 0;
    if (!ok || inf) return;
    const int j = (int)(L / FB_ENTRIES);
    const uint32_t d = (uint32_t)(L % FB_ENTRIES) + 1u;
    g1p acc;
    g1_fixed_entry(acc, base, d, 8 * j);
    fp zn, zi, ax, ay;
    fp_norm1(zn, acc.z);
    fp_inv(zi, zn);
    g1p an;
    g1_norm1(an, acc);
    g1_to_affine(ax, ay, an, zi);
    msm_store_pt(buf + FB_HEADER_DWORDS + L * FB_G1_DWORDS, ax, ay);
}

// proj[off + i] = [k_i]B from the table; does nothing when the table is not valid (the generic kernel runs then)
__global__ void __launch_bounds__(BLOCK, 2) g1_fixed_eval_kernel(size_t n, const int32_t* buf, const uint8_t* scalars, int32_t* proj, size_t proj_stride,
                                                              size_t proj_off) {
    if (buf[HDR_VALID] == 0) return;
    const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    uint32_t raw[8], k[8];
    load_raw32(raw, scalars + 32 * i);
    scalar_from_raw32(k, raw);
    g1p acc, o;
    g1_fixed_eval(acc, buf + FB_HEADER_DWORDS, k);
    g1_norm1(o, acc);
    soa_store_g1(proj, proj_stride, proj_off + i, o);
}

__global__ void __launch_bounds__(BLOCK, 2) g2_fixed_table_kernel(const uint8_t* base192, int32_t* buf) {
    int32_t* header = buf;
    if (header[HDR_REBUILD] == 0) return;
    const size_t L = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (L >= (size_t)FB_G2_WINDOWS * FB_ENTRIES) return;
    g2p base;
    bool inf, ok;
    g2_parse192(base.x, base.y, inf, ok, base192);
    fp2_one(base.z);
    if (L == 0) header[HDR_VALID] = (ok && !inf && g2_in_subgroup(base)) ? 1 : 0;
    if (!ok || inf) return;
    const int j = (int)(L / FB_ENTRIES);
    const uint32_t d = (uint32_t)(L % FB_ENTRIES) + 1u;
    g2p acc;
    g2_fixed_entry(acc, base, d, 8 * j);
    fp2 zn, zi, ax, ay;
    fp2_norm1(zn, acc.z);
    fp2_inv(zi, zn);
    fp2_mul(ax, acc.x, zi); fp2_mul(ay, acc.y, zi);
    fp2_norm1(ax, ax); fp2_norm1(ay, ay);
    fb_store_g2(buf + FB_HEADER_DWORDS + L * FB_G2_DWORDS, ax, ay);
}

// out[i] = addend + [k_i]Q (affine, canonical 192 B or 97 B — or projective SoA for g2_finish_kernel when proj is given);
// addend = one broadcast 192-byte point or nullptr.  Does nothing when the table is not valid.
__global__ void __launch_bounds__(BLOCK, 2) g2_fixed_eval_kernel(size_t n, const int32_t* buf, const uint8_t* scalars, const uint8_t* addend192,
                                                              uint8_t* out, int fmt, int* bad_flag, int32_t* proj, size_t proj_stride) {
    if (buf[HDR_VALID] == 0) return;
    const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    uint32_t raw[8], k[8];
    load_raw32(raw, scalars + 32 * i);
    scalar_from_raw32(k, raw);
    g2p acc;
    g2_fixed_eval(acc, buf + FB_HEADER_DWORDS, k);
    bool wok = true;
    if (addend192) {                                      // kernel-uniform
        g2p w, inf_pt;
        bool winf;
        g2_parse192(w.x, w.y, winf, wok, addend192);
        fp2_one(w.z);
        g2_set_inf(inf_pt);
        fp2_select(w.x, winf, inf_pt.x, w.x); fp2_select(w.y, winf, inf_pt.y, w.y); fp2_select(w.z, winf, inf_pt.z, w.z);
        if (!wok) *bad_flag = 1;
        g2_norm1(acc, acc);
        g2_add(acc, w);
    }
    if (proj) g2_store_proj(proj, proj_stride, i, acc, !wok);                // kernel-uniform: affine conversion by g2_finish_kernel
    else g2_store_affine(out + (size_t)fmt * i, acc, fmt, !wok);
}

}  // namespace c12381
