// TEST HARNESS (CPU): compiles the DEVICE headers of crypto12381_amd/csrc for the host with
// C12381_CHECK_BOUNDS, so that (1) the limb/value bound discipline of fp.hpp is asserted on
// every operation and (2) the exact device algorithms can be compared with the oracle in the
// build container, which has no GPU.  This is not a product path and not a fallback: the
// C-ABI library never contains this code.
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../crypto12381_amd/csrc/fp.hpp"
#include "../../crypto12381_amd/csrc/codec.hpp"
#include "../../crypto12381_amd/csrc/g1.hpp"

using namespace c12381;

static void load_raw(uint32_t* w, const uint8_t* p, int nwords) { std::memcpy(w, p, 4 * (size_t)nwords); }

extern "C" {

// op: 0 mul 1 add 2 sub 3 sqr 4 neg 5 inv 6 sqrt-candidate 7 mul_small(12) 8 norm1 round trip
int sim_fp_op_batch(int op, size_t n, const uint8_t* a, const uint8_t* b, uint8_t* out) {
    for (size_t i = 0; i < n; ++i) {
        uint32_t ra[12], rb[12], ro[12];
        fp x, y, r;
        load_raw(ra, a + 48 * i, 12); fp_from_raw48(x, ra);
        if (b) { load_raw(rb, b + 48 * i, 12); fp_from_raw48(y, rb); }
        switch (op) {
            case 0: fp_mul(r, x, y); break;
            case 1: fp_add(r, x, y); break;
            case 2: fp_sub(r, x, y); break;
            case 3: fp_sqr(r, x); break;
            case 4: fp_neg(r, x); break;
            case 5: fp_inv(r, x); break;
            case 6: fp_sqrt_candidate(r, x); break;
            case 7: fp_mul_small(r, x, 12); break;
            case 8: { fp t; fp_add(t, x, y); fp_sub(t, t, y); fp_dbl(t, t); fp_norm1(r, t); fp_sub(r, r, x); } break;
            default: return -1;
        }
        fp_to_raw48(ro, r);
        std::memcpy(out + 48 * i, ro, 48);
    }
    return 0;
}

// the per-lane scalar multiplication of g1.hpp, one "lane" at a time, table slab of stride 1
int sim_g1_mul_batch(size_t n, const uint8_t* pts96, const uint8_t* scalars32, uint8_t* out, int fmt) {
    std::vector<int32_t> tab(G1_TAB_DWORDS);
    for (size_t i = 0; i < n; ++i) {
        uint32_t rp[24], rs[8], k[8];
        load_raw(rp, pts96 + 96 * i, 24); load_raw(rs, scalars32 + 32 * i, 8);
        const bool inf = raw_all_zero(rp, 24);
        fp px, py;
        fp_from_raw48(px, rp); fp_from_raw48(py, rp + 12);
        scalar_from_raw32(k, rs);
        g1p acc;
        g1_scalar_mul(acc, px, py, inf, k, tab.data(), 1, 0);
        uint8_t* o = out + (size_t)fmt * i;
        if (g1_is_inf(acc)) { std::memset(o, 0, fmt); continue; }
        fp zn, zi, ax, ay;
        fp_norm1(zn, acc.z);
        fp_inv(zi, zn);
        g1_to_affine(ax, ay, acc, zi);
        uint32_t rx[12], ry[12];
        fp_to_raw48(rx, ax); fp_to_raw48(ry, ay);
        if (fmt == 96) { std::memcpy(o, rx, 48); std::memcpy(o + 48, ry, 48); }
        else { o[0] = (uint8_t)(0x02 | fp_sign(ay)); std::memcpy(o + 1, rx, 48); }
    }
    return 0;
}

// scalar decomposition check: returns k0, k1 (16 bytes each, little-endian words)
int sim_glv_split(const uint8_t* scalar32, uint32_t* k0, uint32_t* k1) {
    uint32_t rs[8], k[8];
    load_raw(rs, scalar32, 8);
    scalar_from_raw32(k, rs);
    scalar_mod_r(k);
    uint32_t a[4], b[4];
    scalar_glv_split(a, b, k);
    std::memcpy(k0, a, 16); std::memcpy(k1, b, 16);
    return 0;
}

}  // extern "C"
