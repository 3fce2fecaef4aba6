// G2 points with half an Fp2 element per lane (fp2h.hpp): table records and result stores for the templated point
// arithmetic of g2.hpp.  Two adjacent lanes hold one point; a lane's record is three Fp values — the format of a G1
// table entry (176 bytes, eleven 16-byte accesses).  The host build (tests/host_sim) holds both halves in one object: its
// "lane table" is two consecutive lane tables, real half first.
#pragma once
#include "g2.hpp"
#include "fp2h.hpp"

namespace c12381 {

using g2hp = g2pt<fp2h>;
template <> struct g2_inline_loop<fp2h> { static constexpr bool value = true; };   // (out of line, the running point in private memory: profiles/r03_ab_acc_fence_g2_inline.txt)
constexpr int G2H_TAB_DWORDS = G2_TAB * G1_ENT_DWORDS;          // per lane: 16 entries x 44 dwords (4-bit windows: 8)

C12381_HD constexpr int g2_ent_dwords(const g2hp&) { return G1_ENT_DWORDS; }
#if defined(__HIPCC__)
C12381_D void tab_store_g2(int32_t* ent, const g2hp& p) {
    g1p t;
    t.x = p.x.v; t.y = p.y.v; t.z = p.z.v;
    tab_store_g1(ent, t);
}
C12381_D void tab_load_g2(g2hp& p, const int32_t* ent) {
    g1p t;
    tab_load_g1(t, ent);
    p.x.v = t.x; p.y.v = t.y; p.z.v = t.z;
}
// this lane's halves into the projective SoA of g2_finish_kernel (soa_store_g2 layout: x.a, x.b, y.a, y.b, z.a, z.b);
// an invalid input is marked X = 1, Y = Z = 0
C12381_D void g2h_store_proj(int32_t* proj, size_t stride, size_t i, const g2hp& acc, bool invalid) {
    g2hp o;
    g2_norm1(o, acc);
    if (invalid) { fp2_one(o.x); fp2_zero(o.y); fp2_zero(o.z); }
    const size_t h = (size_t)fp2h_role();
    soa_store_fp(proj + (0 + h) * NL * stride, stride, i, o.x.v);
    soa_store_fp(proj + (2 + h) * NL * stride, stride, i, o.y.v);
    soa_store_fp(proj + (4 + h) * NL * stride, stride, i, o.z.v);
}
#else
inline void tab_store_g2(int32_t* ent, const g2hp& p) {
    for (int h = 0; h < 2; ++h) {
        g1p t;
        t.x = p.x.h[h]; t.y = p.y.h[h]; t.z = p.z.h[h];
        tab_store_g1(ent + h * G2H_TAB_DWORDS, t);
    }
}
inline void tab_load_g2(g2hp& p, const int32_t* ent) {
    for (int h = 0; h < 2; ++h) {
        g1p t;
        tab_load_g1(t, ent + h * G2H_TAB_DWORDS);
        p.x.h[h] = t.x; p.y.h[h] = t.y; p.z.h[h] = t.z;
    }
}
#endif

}  // namespace c12381
