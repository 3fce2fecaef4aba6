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

// operation counters (fp.hpp): reset / read.  cols = scanned product columns (27 per 14 x 14 limb product), reds = reductions
void sim_ops_reset(void) { g_ops_cols = 0; g_ops_reds = 0; }
void sim_ops_read(unsigned long long* cols, unsigned long long* reds) { *cols = g_ops_cols.load(); *reds = g_ops_reds.load(); }

// op: 0 mul 1 add 2 sub 3 sqr 4 neg 5 inv (divsteps) 6 sqrt-candidate 7 mul_small(12) 8 norm1 round trip 9 inv (Fermat)
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
            case 9: fp_inv_fermat(r, x); break;
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
    std::vector<int32_t> tabv(G1_TAB_DWORDS + 4);
    int32_t* tab = reinterpret_cast<int32_t*>((reinterpret_cast<uintptr_t>(tabv.data()) + 15) & ~(uintptr_t)15);
    for (size_t i = 0; i < n; ++i) {
        uint32_t rp[24], rs[8], k[8];
        load_raw(rp, pts96 + 96 * i, 24); load_raw(rs, scalars32 + 32 * i, 8);
        const bool inf = raw_all_zero(rp, 24);
        fp px, py;
        fp_from_raw48(px, rp); fp_from_raw48(py, rp + 12);
        scalar_from_raw32(k, rs);
        g1p acc;
        g1_scalar_mul(acc, px, py, inf, k, tab);
        if (scalar_below_x2(k) && !inf) {                 // g1_small_scalar_kernel
            g1p base, n;
            base.x = px; base.y = py; fp_one(base.z);
            g1_norm1(n, acc);
            g1_glv_small_scalar_term(n, base);
            acc = n;
        }
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

// ---------------------------------------------------------------- G2 / pairing
#include "../../crypto12381_amd/csrc/fp2.hpp"
#include "../../crypto12381_amd/csrc/fp12.hpp"
#include "../../crypto12381_amd/csrc/g2.hpp"
#include "../../crypto12381_amd/csrc/g2h.hpp"
#include "../../crypto12381_amd/csrc/pairing.hpp"

static void fp2_from_bytes96(fp2& r, const uint8_t* p) {      // b || a
    uint32_t raw[24];
    load_raw(raw, p, 24);
    fp_from_raw48(r.b, raw); fp_from_raw48(r.a, raw + 12);
}
static void fp2_to_bytes96(uint8_t* p, const fp2& x) {
    uint32_t raw[24];
    fp_to_raw48(raw, x.b); fp_to_raw48(raw + 12, x.a);
    std::memcpy(p, raw, 96);
}

extern "C" {

int sim_g2_mul_batch(size_t n, const uint8_t* pts192, const uint8_t* scalars32, uint8_t* out, int fmt) {
    std::vector<int32_t> tabv(G2_TAB_DWORDS + 4);
    int32_t* tab = reinterpret_cast<int32_t*>((reinterpret_cast<uintptr_t>(tabv.data()) + 15) & ~(uintptr_t)15);
    for (size_t i = 0; i < n; ++i) {
        uint32_t rp[48], rs[8], k[8];
        load_raw(rp, pts192 + 192 * i, 48); load_raw(rs, scalars32 + 32 * i, 8);
        const bool inf = raw_all_zero(rp, 48);
        fp2 qx, qy;
        fp2_from_bytes96(qx, pts192 + 192 * i); fp2_from_bytes96(qy, pts192 + 192 * i + 96);
        scalar_from_raw32(k, rs);
        g2p acc;
        g2_scalar_mul(acc, qx, qy, inf, k, tab);
        uint8_t* o = out + (size_t)fmt * i;
        if (fp2_is_zero(acc.z)) { std::memset(o, 0, fmt); continue; }
        fp2 zn, zi, ax, ay;
        fp2_norm1(zn, acc.z);
        fp2_inv(zi, zn);
        fp2_mul(ax, acc.x, zi); fp2_mul(ay, acc.y, zi);
        if (fmt == 192) { fp2_to_bytes96(o, ax); fp2_to_bytes96(o + 96, ay); }
        else { o[0] = (uint8_t)(0x02 | fp2_sign(ay)); fp2_to_bytes96(o + 1, ax); }
    }
    return 0;
}

// the TWO-lanes-per-point form of the same multiplication (k_g2h.hip: g2_scalar_mul<fp2h>): both halves in one object, the
// per-lane routines of fp2h.hpp run once per role, so their limb / value bounds are asserted here
int sim_g2h_mul_batch(size_t n, const uint8_t* pts192, const uint8_t* scalars32, uint8_t* out, int fmt) {
    std::vector<int32_t> tabv(2 * G2H_TAB_DWORDS + 4);
    int32_t* tab = reinterpret_cast<int32_t*>((reinterpret_cast<uintptr_t>(tabv.data()) + 15) & ~(uintptr_t)15);
    for (size_t i = 0; i < n; ++i) {
        uint32_t rp[48], rs[8], k[8];
        load_raw(rp, pts192 + 192 * i, 48); load_raw(rs, scalars32 + 32 * i, 8);
        const bool inf = raw_all_zero(rp, 48);
        fp2 qx, qy;
        fp2_from_bytes96(qx, pts192 + 192 * i); fp2_from_bytes96(qy, pts192 + 192 * i + 96);
        scalar_from_raw32(k, rs);
        fp2h hx, hy;
        fp2h_from(hx, qx); fp2h_from(hy, qy);
        g2hp hacc;
        g2_scalar_mul(hacc, hx, hy, inf, k, tab);
        g2p acc;
        fp2h_to(acc.x, hacc.x); fp2h_to(acc.y, hacc.y); fp2h_to(acc.z, hacc.z);
        uint8_t* o = out + (size_t)fmt * i;
        if (fp2_is_zero(acc.z)) { std::memset(o, 0, fmt); continue; }
        fp2 zn, zi, ax, ay;
        fp2_norm1(zn, acc.z);
        fp2_inv(zi, zn);
        fp2_mul(ax, acc.x, zi); fp2_mul(ay, acc.y, zi);
        if (fmt == 192) { fp2_to_bytes96(o, ax); fp2_to_bytes96(o + 96, ay); }
        else { o[0] = (uint8_t)(0x02 | fp2_sign(ay)); fp2_to_bytes96(o + 1, ax); }
    }
    return 0;
}

static void pair_load(fp& px, fp& py, bool& pinf, fp2& qx, fp2& qy, bool& qinf, const uint8_t* g1, const uint8_t* g2) {
    uint32_t rp[24], rq[48];
    load_raw(rp, g1, 24); load_raw(rq, g2, 48);
    pinf = raw_all_zero(rp, 24); qinf = raw_all_zero(rq, 48);
    fp_from_raw48(px, rp); fp_from_raw48(py, rp + 12);
    fp2_from_bytes96(qx, g2); fp2_from_bytes96(qy, g2 + 96);
}
static void gt_store(uint8_t* o, const fp12& f) {
    for (int j = 0; j < 12; ++j) {
        uint32_t raw[12];
        fp_to_raw48(raw, fp12_coord(f, j));
        std::memcpy(o + 48 * j, raw, 48);
    }
}

int sim_pair_batch(size_t n, const uint8_t* g1_96, const uint8_t* g2_192, uint8_t* gt576) {
    for (size_t i = 0; i < n; ++i) {
        fp px, py; fp2 qx, qy; bool pinf, qinf;
        pair_load(px, py, pinf, qx, qy, qinf, g1_96 + 96 * i, g2_192 + 192 * i);
        fp12 f;
        miller_loop(f, px, py, pinf, qx, qy, qinf);
        final_exp(f);
        gt_store(gt576 + 576 * i, f);
    }
    return 0;
}

int sim_pair_eq_batch(size_t n, const uint8_t* a1, const uint8_t* a2, const uint8_t* b1, const uint8_t* b2, uint8_t* ok) {
    for (size_t i = 0; i < n; ++i) {
        fp px, py; fp2 qx, qy; bool pinf, qinf;
        fp12 f, g, gc, t;
        pair_load(px, py, pinf, qx, qy, qinf, a1 + 96 * i, a2 + 192 * i);
        miller_loop(f, px, py, pinf, qx, qy, qinf);
        pair_load(px, py, pinf, qx, qy, qinf, b1 + 96 * i, b2 + 192 * i);
        miller_loop(g, px, py, pinf, qx, qy, qinf);
        fp12_conj(gc, g);
        fp12_mul(t, f, gc);
        final_exp(t);
        ok[i] = fp12_is_one(t) ? 1 : 0;
    }
    return 0;
}

}  // extern "C"

// ---------------------------------------------------------------- decompression, Miller/fexp split, GT ops
extern "C" {

int sim_g1_decompress_batch(size_t n, const uint8_t* in49, uint8_t* out96, uint8_t* status) {
    for (size_t i = 0; i < n; ++i) {
        const uint8_t* sp = in49 + 49 * i;
        uint8_t* o = out96 + 96 * i;
        std::memset(o, 0, 96);
        if (sp[0] == 0) { status[i] = 1; continue; }
        if (sp[0] != 2 && sp[0] != 3) { status[i] = 0; continue; }
        uint32_t raw[12];
        std::memcpy(raw, sp + 1, 48);
        fp x, y;
        fp_from_raw48(x, raw);
        const bool ok = g1_set_x(y, x, sp[0] & 1);
        status[i] = ok ? 1 : 0;
        if (!ok) continue;
        uint32_t rx[12], ry[12];
        fp_to_raw48(rx, x); fp_to_raw48(ry, y);
        std::memcpy(o, rx, 48); std::memcpy(o + 48, ry, 48);
    }
    return 0;
}
int sim_g2_decompress_batch(size_t n, const uint8_t* in97, uint8_t* out192, uint8_t* status) {
    for (size_t i = 0; i < n; ++i) {
        const uint8_t* sp = in97 + 97 * i;
        uint8_t* o = out192 + 192 * i;
        std::memset(o, 0, 192);
        if (sp[0] == 0) { status[i] = 1; continue; }
        if (sp[0] == 4) { status[i] = 0; continue; }
        uint8_t tmp[96];
        std::memcpy(tmp, sp + 1, 96);
        fp2 x, y;
        fp2_from_bytes96(x, tmp);
        const bool ok = g2_set_x(y, x, sp[0] & 1);
        status[i] = ok ? 1 : 0;
        if (!ok) continue;
        fp2_to_bytes96(o, x); fp2_to_bytes96(o + 96, y);
    }
    return 0;
}
static void gt_load(fp12& f, const uint8_t* p) {
    for (int j = 0; j < 12; ++j) {
        uint32_t raw[12];
        std::memcpy(raw, p + 48 * j, 48);
        fp_from_raw48(fp12_coord_mut(f, j), raw);
    }
}
// op 0 mul, 1 conj, 2 pow (b = 32-byte exponents), 3 final exponentiation, 4 is-unity (out[0] per element)
int sim_gt_op_batch(int op, size_t n, const uint8_t* a576, const uint8_t* b, uint8_t* out) {
    for (size_t i = 0; i < n; ++i) {
        fp12 x, y, r;
        gt_load(x, a576 + 576 * i);
        if (op == 0) { gt_load(y, b + 576 * i); fp12_mul(r, x, y); }
        else if (op == 1) fp12_conj(r, x);
        else if (op == 2) { uint32_t rs[8], e[8]; load_raw(rs, b + 32 * i, 8); scalar_from_raw32(e, rs); fp12_pow_generic(r, x, e); }
        else if (op == 3) { r = x; final_exp(r); }
        else if (op == 4) { out[i] = fp12_is_one(x) ? 1 : 0; continue; }
        else return -1;
        gt_store(out + 576 * i, r);
    }
    return 0;
}
int sim_miller_batch(size_t n, const uint8_t* g1_96, const uint8_t* g2_192, uint8_t* out576) {
    for (size_t i = 0; i < n; ++i) {
        fp px, py; fp2 qx, qy; bool pinf, qinf;
        pair_load(px, py, pinf, qx, qy, qinf, g1_96 + 96 * i, g2_192 + 192 * i);
        fp12 f;
        miller_loop(f, px, py, pinf, qx, qy, qinf);
        gt_store(out576 + 576 * i, f);
    }
    return 0;
}

}  // extern "C"

// ---------------------------------------------------------------- three-lanes-per-pairing variant (pairing3.hpp)
#include <pthread.h>
#include "../../crypto12381_amd/csrc/pairing3.hpp"

namespace {
struct TriBox { pthread_barrier_t bar; unsigned char slot[3][sizeof(fp4)]; };
thread_local TriBox* tl_box = nullptr;
}
namespace c12381 {
void c12381_tri_exchange(void* out, const void* in, size_t bytes, int src_role, const tri& t) {
    std::memcpy(tl_box->slot[t.role], in, bytes);
    pthread_barrier_wait(&tl_box->bar);
    std::memcpy(out, tl_box->slot[src_role], bytes);
    pthread_barrier_wait(&tl_box->bar);
}
}
namespace {
int g_pow_route = 0, g_pow_windowed = 0;
struct Tri3Job { TriBox* box; int role; size_t n; const uint8_t *a1, *a2, *b1, *b2; uint8_t* out; int mode; const int32_t *tab1, *tab2; };
void gt_store_coeff(uint8_t* o576, const fp4& x, int role) {
    // FP12_toOctet order c | b | a, each Fp4 as b.b, b.a, a.b, a.a
    uint8_t* o = o576 + (role == 0 ? 384 : (role == 1 ? 192 : 0));
    const fp* order[4] = {&x.b.b, &x.b.a, &x.a.b, &x.a.a};
    for (int j = 0; j < 4; ++j) { uint32_t raw[12]; fp_to_raw48(raw, *order[j]); std::memcpy(o + 48 * j, raw, 48); }
}
void gt_load_coeff(fp4& x, const uint8_t* p576, int role) {
    const uint8_t* p = p576 + (role == 0 ? 384 : (role == 1 ? 192 : 0));
    fp* order[4] = {&x.b.b, &x.b.a, &x.a.b, &x.a.a};
    for (int j = 0; j < 4; ++j) { uint32_t raw[12]; load_raw(raw, p + 48 * j, 12); fp_from_raw48(*order[j], raw); }
}
void* tri3_worker(void* arg) {
    Tri3Job* jb = (Tri3Job*)arg;
    tl_box = jb->box;
    tri t{jb->role, 0};
    if (jb->mode >= 10) {                 // GT operators on triples: a1 = 576-byte elements, b1 = second operand
        const int op = jb->mode - 10;
        for (size_t i = 0; i < jb->n; ++i) {
            fp4 x, r, h;
            gt_load_coeff(x, jb->a1 + 576 * i, jb->role);
            if (op == 0) { fp4 y; gt_load_coeff(y, jb->b1 + 576 * i, jb->role); f12t_mul(r, x, y, t); }
            else if (op == 1) f12t_conj(r, x, t);
            else if (op == 2) {
                // as gt3_op_kernel: the windowed ladder for members of the cyclotomic subgroup, the reference's digit sequence otherwise
                // (g_pow_route: 0 = as the kernel decides, 1 = always the generic ladder, 2 = in the five pieces of the work-queue kernel; g_pow_windowed counts the elements that took the windows)
                uint32_t raw[8], e[8]; load_raw(raw, jb->b1 + 32 * i, 8); scalar_from_raw32(e, raw);
                const bool cyc = f12t_is_cyclotomic(h, x, t);
                if (g_pow_route == 2) {
                    // the five tasks of gt3_pow_queue_kernel: table / start value, then four quarters of the ladder on the accumulator
                    fp4 tab[16];
                    if (cyc) f12t_pow_window_table(r, x, t, [&](int k, const fp4& v) { tab[k] = v; });
                    f12t_pow_acc_init(r, t);
                    for (int p = 1; p <= 4; ++p) {
                        if (cyc) { const int whi = 63 - 16 * (p - 1); f12t_pow_window_range(r, e, whi, whi - 15, t, [&](fp4& m, int k) { m = tab[k]; }); }
                        else { const int hi = 257 - 64 * (p - 1), lo = p == 4 ? 1 : hi - 63; f12t_pow_generic_range(r, x, e, hi, lo, t); }
                    }
                    f12t_unscale3_h(r);
                    if (cyc && jb->role == 0) ++g_pow_windowed;
                } else if (cyc && g_pow_route == 0) {
                    fp4 tab[16];
                    f12t_pow_window(r, x, e, t, [&](int k, const fp4& v) { tab[k] = v; }, [&](fp4& m, int k) { m = tab[k]; });
                    if (jb->role == 0) ++g_pow_windowed;
                } else { r = x; f12t_pow_generic(r, e, t); }
            }
            else if (op == 3) { r = x; f12t_final_exp_ws(r, h, t); }
            else { const bool one = f12t_is_one(x, t); if (jb->role == 0) jb->out[i] = one ? 1 : 0; continue; }
            gt_store_coeff(jb->out + 576 * i, r, jb->role);
        }
        return nullptr;
    }
    for (size_t i = 0; i < jb->n; ++i) {
        fp px, py; fp2 qx, qy; bool pinf, qinf;
        pair_slot slot;                   // the Miller loop keeps the lane's G1 coordinate next to the Fp4 (pairing3.hpp)
        fp4& F = slot.v;
        pair_load(px, py, pinf, qx, qy, qinf, jb->a1 + 96 * i, jb->a2 + 192 * i);
        if (jb->mode == 3) {          // Miller value alone
            miller3_loop(F, px, py, pinf, qx, qy, qinf, t);
            gt_store_coeff(jb->out + 576 * i, F, jb->role);
        } else if (jb->mode == 2) {          // product of two pairings whose G2 arguments are fixed: lines from the tables
            fp px2, py2; fp2 qx2, qy2; bool pinf2, qinf2;
            pair_load(px2, py2, pinf2, qx2, qy2, qinf2, jb->b1 + 96 * i, jb->b2);
            f12t_one(F, t);
            miller3_range2_fixed(F, px, py, pinf, jb->tab1, px2, py2, pinf2, jb->tab2, 64, 33, t);
            miller3_range2_fixed(F, px, py, pinf, jb->tab1, px2, py2, pinf2, jb->tab2, 32, 1, t);
            f12t_conj(F, F, t);
            f12t_final_exp(F, t);
            gt_store_coeff(jb->out + 576 * i, F, jb->role);
        } else if (jb->mode == 1) {          // equality as pair3_eq_kernel evaluates it: joint loop on (a1, a2), (-b1, b2)
            fp px2, py2, ny; fp2 qx2, qy2; bool pinf2, qinf2;
            pair_load(px2, py2, pinf2, qx2, qy2, qinf2, jb->b1 + 96 * i, jb->b2 + 192 * i);
            fp_neg(ny, py2); fp_norm1(py2, ny);
            miller3_loop2(F, px, py, pinf, qx, qy, qinf, px2, py2, pinf2, qx2, qy2, qinf2, t);
            f12t_final_exp(F, t);
            const bool ok = f12t_is_one(F, t);
            if (jb->role == 0) jb->out[i] = ok ? 1 : 0;
        } else {
            miller3_loop(F, px, py, pinf, qx, qy, qinf, t);
            f12t_final_exp(F, t);
            gt_store_coeff(jb->out + 576 * i, F, jb->role);
        }
    }
    return nullptr;
}
int run_tri3(size_t n, const uint8_t* a1, const uint8_t* a2, const uint8_t* b1, const uint8_t* b2, uint8_t* out, int mode,
             const int32_t* tab1 = nullptr, const int32_t* tab2 = nullptr) {
    TriBox box;
    pthread_barrier_init(&box.bar, nullptr, 3);
    pthread_t th[3]; Tri3Job jb[3];
    for (int r = 0; r < 3; ++r) { jb[r] = Tri3Job{&box, r, n, a1, a2, b1, b2, out, mode, tab1, tab2}; pthread_create(&th[r], nullptr, tri3_worker, &jb[r]); }
    for (int r = 0; r < 3; ++r) pthread_join(th[r], nullptr);
    pthread_barrier_destroy(&box.bar);
    return 0;
}
}
extern "C" {
int sim_pair3_batch(size_t n, const uint8_t* g1_96, const uint8_t* g2_192, uint8_t* gt576) { return run_tri3(n, g1_96, g2_192, nullptr, nullptr, gt576, 0); }
int sim_pair3_eq_batch(size_t n, const uint8_t* a1, const uint8_t* a2, const uint8_t* b1, const uint8_t* b2, uint8_t* ok) { return run_tri3(n, a1, a2, b1, b2, ok, 1); }
}

// ---------------------------------------------------------------- bucket-method MSM (msm.hpp), run sequentially
#include <algorithm>
#include "../../crypto12381_amd/csrc/msm.hpp"
// G2 scalar decomposition check: u[0..3] as (lo, hi) word pairs
extern "C" int sim_gs_split(const uint8_t* scalar32, uint32_t* u8) {
    uint32_t rs[8], k[8];
    load_raw(rs, scalar32, 8);
    scalar_from_raw32(k, rs);
    scalar_mod_r(k);
    uint32_t u[4][2];
    scalar_gs_split(u, k);
    for (int i = 0; i < 4; ++i) { u8[2 * i] = u[i][0]; u8[2 * i + 1] = u[i][1]; }
    return 0;
}
extern "C" int sim_g1_msm_pippenger(size_t n, const uint8_t* pts96, const uint8_t* scalars32, uint8_t* out, int fmt, int force_c) {
    const int c = force_c > 0 ? force_c : msm_window_bits(n);
    const int W = msm_windows(c);
    const size_t E = msm_entries(n, W);
    std::vector<int32_t> pts2v((size_t)2 * n * MSM_PT_STRIDE + 4);
    int32_t* pts2 = reinterpret_cast<int32_t*>((reinterpret_cast<uintptr_t>(pts2v.data()) + 15) & ~(uintptr_t)15);
    std::vector<uint32_t> keys(E), vals(E);
    for (size_t i = 0; i < n; ++i) {
        uint32_t rp[24], rs[8];
        load_raw(rp, pts96 + 96 * i, 24); load_raw(rs, scalars32 + 32 * i, 8);
        if (!msm_prep_one<uint32_t>(i, n, rp, rs, c, W, pts2, keys.data(), vals.data())) return -3;
    }
    {   // the large-product form (16-bit digit keys, positional values) describes the same entries
        std::vector<uint16_t> k16(E);
        std::vector<int32_t> p2((size_t)2 * n * MSM_PT_STRIDE + 4);
        int32_t* p2a = reinterpret_cast<int32_t*>((reinterpret_cast<uintptr_t>(p2.data()) + 15) & ~(uintptr_t)15);
        if (c <= 16) {
            for (size_t i = 0; i < n; ++i) {
                uint32_t rp[24], rs[8];
                load_raw(rp, pts96 + 96 * i, 24); load_raw(rs, scalars32 + 32 * i, 8);
                if (!msm_prep_one<uint16_t>(i, n, rp, rs, c, W, p2a, k16.data(), nullptr)) return -3;
            }
            for (int w = 0; w <= W; ++w) {
                const size_t seg = (size_t)2 * w * n, len = w < W ? 2 * n : n;
                for (size_t x = 0; x < len; ++x) {
                    if (k16[seg + x] != (keys[seg + x] & ((1u << c) - 1u))) return -4;
                    if (msm_entry_value((uint32_t)x, (uint32_t)n) != vals[seg + x]) return -5;
                }
            }
        }
    }
    std::vector<size_t> order(E);
    for (size_t j = 0; j < E; ++j) order[j] = j;
    // as the device does it: every window segment of 2n entries (and the small-scalar segment) is sorted on its own by the digit bits
    for (int w = 0; w <= W; ++w) {
        const size_t a = (size_t)2 * w * n, b = w < W ? a + 2 * n : E;
        std::stable_sort(order.begin() + a, order.begin() + b, [&](size_t x, size_t y) { return (keys[x] & ((1u << c) - 1u)) < (keys[y] & ((1u << c) - 1u)); });
    }
    std::vector<uint32_t> ks(E), vs(E);
    for (size_t j = 0; j < E; ++j) { ks[j] = keys[order[j]]; vs[j] = vals[order[j]]; }
    const size_t nb = (size_t)1 << c, nbk = nb * W;
    std::vector<uint32_t> lo(nbk + 2, 0), hi(nbk + 2, 0);        // + the small-scalar bucket (nbk)
    for (size_t j = 0; j < E; ++j) msm_ranges_one(j, E, ks.data(), c, W, lo.data(), hi.data());
    std::vector<int32_t> bkv((nbk + 1) * G1_ENT_DWORDS + 4);
    int32_t* bk = reinterpret_cast<int32_t*>((reinterpret_cast<uintptr_t>(bkv.data()) + 15) & ~(uintptr_t)15);
    for (size_t b = 0; b <= nbk; ++b) {
        g1p acc, nn;
        msm_bucket_one(acc, lo[b], hi[b], vs.data(), pts2);
        g1_norm1(nn, acc);
        tab_store_g1(bk + b * G1_ENT_DWORDS, nn);
    }
    const size_t chunks = (nb + MSM_CHUNK - 1) / MSM_CHUNK;
    std::vector<int32_t> rw((size_t)3 * NL * W);
    for (int w = 0; w < W; ++w) {
        g1p sum; g1_set_inf(sum);
        for (size_t ch = 0; ch < chunks; ++ch) {
            g1p part;
            msm_wreduce_one(part, bk + (size_t)w * nb * G1_ENT_DWORDS, (uint32_t)(ch * MSM_CHUNK), (uint32_t)nb);
            g1_add(sum, part);
            g1p nn; g1_norm1(nn, sum); sum = nn;
        }
        soa_store_g1(rw.data(), (size_t)W, (size_t)w, sum);
    }
    g1p acc;
    msm_horner(acc, rw.data(), (size_t)W, W, c);
    {   // the [r]phi(S) owed by the scalars below x^2
        g1p S, term, nn;
        tab_load_g1(S, bk + nbk * G1_ENT_DWORDS);
        msm_small_term(term, S);
        g1_norm1(nn, term);
        g1_add(acc, nn);
        g1_norm1(nn, acc); acc = nn;
    }
    if (g1_is_inf(acc)) { std::memset(out, 0, fmt); return 0; }
    fp zn, zi, ax, ay;
    fp_norm1(zn, acc.z);
    fp_inv(zi, zn);
    g1_to_affine(ax, ay, acc, zi);
    uint32_t rx[12], ry[12];
    fp_to_raw48(rx, ax); fp_to_raw48(ry, ay);
    if (fmt == 96) { std::memcpy(out, rx, 48); std::memcpy(out + 48, ry, 48); }
    else { out[0] = (uint8_t)(0x02 | fp_sign(ay)); std::memcpy(out + 1, rx, 48); }
    return 0;
}

// ---------------------------------------------------------------- hash-to-G1, Zp helpers
#include "../../crypto12381_amd/csrc/fr.hpp"
#include "../../crypto12381_amd/csrc/h2c.hpp"

static void fr_from_bytes32(fr& r, const uint8_t* p) {
    uint32_t raw[8], k[8];
    load_raw(raw, p, 8);
    scalar_from_raw32(k, raw);
    fr_from_words(r, k);
}
static void fr_to_bytes32(uint8_t* p, const fr& a) {
    uint32_t k[8], raw[8];
    fr_to_words(k, a);
    for (int i = 0; i < 8; ++i) raw[i] = bswap32(k[7 - i]);
    std::memcpy(p, raw, 32);
}

extern "C" {

int sim_g1_from_hash_batch(size_t n, const uint8_t* digests64, uint8_t* out, int fmt) {
    for (size_t i = 0; i < n; ++i) {
        uint32_t raw[16];
        load_raw(raw, digests64 + 64 * i, 16);
        g1p acc;
        g1_from_digest(acc, raw);
        uint8_t* o = out + (size_t)fmt * i;
        if (g1_is_inf(acc)) { std::memset(o, 0, fmt); continue; }
        fp zn, zi, ax, ay;
        fp_norm1(zn, acc.z);
        fp_inv(zi, zn);
        g1p an;
        g1_norm1(an, acc);
        g1_to_affine(ax, ay, an, zi);
        uint32_t rx[12], ry[12];
        fp_to_raw48(rx, ax); fp_to_raw48(ry, ay);
        if (fmt == 96) { std::memcpy(o, rx, 48); std::memcpy(o + 48, ry, 48); }
        else { o[0] = (uint8_t)(0x02 | fp_sign(ay)); std::memcpy(o + 1, rx, 48); }
    }
    return 0;
}

int sim_zp_op_batch(int op, size_t n, const uint8_t* a, const uint8_t* b, uint8_t* out) {
    for (size_t i = 0; i < n; ++i) {
        fr x, y, r;
        fr_from_bytes32(x, a + 32 * i);
        if (b) fr_from_bytes32(y, b + 32 * i); else y = x;
        switch (op) {
            case 0: fr_mul(r, x, y); break;
            case 1: fr_add(r, x, y); break;
            case 2: fr_sub(r, x, y); break;
            case 3: fr_neg(r, x); break;
            case 4: fr_inv(r, x); break;
            default: return -1;
        }
        fr_to_bytes32(out + 32 * i, r);
    }
    return 0;
}

int sim_zp_from_hash_batch(size_t n, const uint8_t* digests64, uint8_t* out) {
    for (size_t i = 0; i < n; ++i) {
        uint32_t raw[16], w[16];
        load_raw(raw, digests64 + 64 * i, 16);
        for (int j = 0; j < 16; ++j) w[j] = bswap32(raw[j]);
        fr r;
        fr_from_digest_words(r, w);
        fr_to_bytes32(out + 32 * i, r);
    }
    return 0;
}

}  // extern "C"

// ---------------------------------------------------------------- fixed-base columns (fixed_base.hpp)
#include "../../crypto12381_amd/csrc/fixed_base.hpp"

extern "C" {

// [k_i]B through the table path (entries built lazily with g1_fixed_entry, exactly as the table kernel computes them);
// returns -2 if the base is not a subgroup point (the library then runs the generic path)
int sim_g1_fixed_mul_batch(size_t n, const uint8_t* base96, const uint8_t* scalars32, uint8_t* out96) {
    uint32_t rp[24];
    load_raw(rp, base96, 24);
    if (raw_all_zero(rp, 24)) return -2;
    g1p base;
    fp_from_raw48(base.x, rp); fp_from_raw48(base.y, rp + 12); fp_one(base.z);
    if (!g1_in_subgroup(base)) return -2;
    const size_t entries = (size_t)FB_G1_WINDOWS * FB_ENTRIES;
    std::vector<int32_t> tabv(entries * FB_G1_DWORDS + 4, 0);
    int32_t* tab = reinterpret_cast<int32_t*>((reinterpret_cast<uintptr_t>(tabv.data()) + 15) & ~(uintptr_t)15);
    std::vector<char> done(entries, 0);
    for (size_t i = 0; i < n; ++i) {
        uint32_t rs[8], k[8], kk[8], k0[4], k1[4];
        load_raw(rs, scalars32 + 32 * i, 8);
        scalar_from_raw32(k, rs);
        for (int w = 0; w < 8; ++w) kk[w] = k[w];
        scalar_mod_r(kk);
        scalar_glv_split(k0, k1, kk);
        for (int j = 0; j < FB_G1_WINDOWS; ++j)
            for (int h = 0; h < 2; ++h) {
                const uint32_t d = ((h ? k1 : k0)[j >> 2] >> (8 * (j & 3))) & 255u;
                if (!d) continue;
                const size_t L = (size_t)j * FB_ENTRIES + (d - 1);
                if (done[L]) continue;
                g1p acc, an;
                g1_fixed_entry(acc, base, d, 8 * j);
                fp zn, zi, ax, ay;
                fp_norm1(zn, acc.z); fp_inv(zi, zn);
                g1_norm1(an, acc);
                g1_to_affine(ax, ay, an, zi);
                fp axn, ayn;
                fp_norm1(axn, ax); fp_norm1(ayn, ay);
                msm_store_pt(tab + L * FB_G1_DWORDS, axn, ayn);
                done[L] = 1;
            }
        g1p acc;
        g1_fixed_eval(acc, tab, k);
        uint8_t* o = out96 + 96 * i;
        if (g1_is_inf(acc)) { std::memset(o, 0, 96); continue; }
        fp zn, zi, ax, ay;
        fp_norm1(zn, acc.z); fp_inv(zi, zn);
        g1p an; g1_norm1(an, acc);
        g1_to_affine(ax, ay, an, zi);
        uint32_t rx[12], ry[12];
        fp_to_raw48(rx, ax); fp_to_raw48(ry, ay);
        std::memcpy(o, rx, 48); std::memcpy(o + 48, ry, 48);
    }
    return 0;
}

int sim_g2_fixed_mul_batch(size_t n, const uint8_t* base192, const uint8_t* scalars32, uint8_t* out192) {
    uint32_t rp[48];
    load_raw(rp, base192, 48);
    if (raw_all_zero(rp, 48)) return -2;
    g2p base;
    fp2_from_bytes96(base.x, base192); fp2_from_bytes96(base.y, base192 + 96); fp2_one(base.z);
    if (!g2_in_subgroup(base)) return -2;
    const size_t entries = (size_t)FB_G2_WINDOWS * FB_ENTRIES;
    std::vector<int32_t> tabv(entries * FB_G2_DWORDS + 4, 0);
    int32_t* tab = reinterpret_cast<int32_t*>((reinterpret_cast<uintptr_t>(tabv.data()) + 15) & ~(uintptr_t)15);
    std::vector<char> done(entries, 0);
    for (size_t i = 0; i < n; ++i) {
        uint32_t rs[8], k[8], kk[8], u[4][2];
        load_raw(rs, scalars32 + 32 * i, 8);
        scalar_from_raw32(k, rs);
        for (int w = 0; w < 8; ++w) kk[w] = k[w];
        scalar_mod_r(kk);
        scalar_gs_split(u, kk);
        for (int a = 0; a < 4; ++a)
            for (int j = 0; j < FB_G2_WINDOWS; ++j) {
                const uint32_t d = (u[a][j >> 2] >> (8 * (j & 3))) & 255u;
                if (!d) continue;
                const size_t L = (size_t)j * FB_ENTRIES + (d - 1);
                if (done[L]) continue;
                g2p acc;
                g2_fixed_entry(acc, base, d, 8 * j);
                fp2 zn, zi, ax, ay;
                fp2_norm1(zn, acc.z); fp2_inv(zi, zn);
                fp2_mul(ax, acc.x, zi); fp2_mul(ay, acc.y, zi);
                fp2_norm1(ax, ax); fp2_norm1(ay, ay);
                fb_store_g2(tab + L * FB_G2_DWORDS, ax, ay);
                done[L] = 1;
            }
        g2p acc;
        g2_fixed_eval(acc, tab, k);
        uint8_t* o = out192 + 192 * i;
        if (g2_is_inf(acc)) { std::memset(o, 0, 192); continue; }
        fp2 zn, zi, ax, ay;
        fp2_norm1(zn, acc.z); fp2_inv(zi, zn);
        fp2_mul(ax, acc.x, zi); fp2_mul(ay, acc.y, zi);
        fp2_to_bytes96(o, ax); fp2_to_bytes96(o + 96, ay);
    }
    return 0;
}

}  // extern "C"

// e(a_i, W) * e(c_i, G) with the two G2 arguments fixed for the batch: coefficient tables + table-driven joint loop
extern "C" int sim_pair2_fixed_batch(size_t n, const uint8_t* a96, const uint8_t* w192, const uint8_t* c96, const uint8_t* g192, uint8_t* gt576) {
    std::vector<int32_t> t1v(FQ_TABLE_DWORDS + 4), t2v(FQ_TABLE_DWORDS + 4);
    int32_t* t1 = reinterpret_cast<int32_t*>((reinterpret_cast<uintptr_t>(t1v.data()) + 15) & ~(uintptr_t)15);
    int32_t* t2 = reinterpret_cast<int32_t*>((reinterpret_cast<uintptr_t>(t2v.data()) + 15) & ~(uintptr_t)15);
    fp2 qx, qy;
    fp2_from_bytes96(qx, w192); fp2_from_bytes96(qy, w192 + 96);
    miller_lines_precompute(t1, qx, qy);
    fp2_from_bytes96(qx, g192); fp2_from_bytes96(qy, g192 + 96);
    miller_lines_precompute(t2, qx, qy);
    // a2 is only parsed for its infinity flag by pair_load: give every lane the fixed point
    std::vector<uint8_t> wrep(192 * n);
    for (size_t i = 0; i < n; ++i) std::memcpy(&wrep[192 * i], w192, 192);
    return run_tri3(n, a96, wrep.data(), c96, g192, gt576, 2, t1, t2);
}

extern "C" int sim_miller3_batch(size_t n, const uint8_t* g1_96, const uint8_t* g2_192, uint8_t* out576) { return run_tri3(n, g1_96, g2_192, nullptr, nullptr, out576, 3); }
// op: 0 mul, 1 conj, 2 pow, 3 final exponentiation, 4 is-unity (out = n bytes)
extern "C" int sim_gt3_op_batch(int op, size_t n, const uint8_t* a576, const uint8_t* b, uint8_t* out) { return run_tri3(n, a576, nullptr, b, nullptr, out, 10 + op); }
// route of the power on triples: 0 = the kernel's choice, 1 = generic ladder only, 2 = the kernel's choice in the queue kernel's five pieces; returns the number of elements that took the windowed ladder so far
extern "C" int sim_gt3_pow_route(int route) { g_pow_route = route; const int v = g_pow_windowed; g_pow_windowed = 0; return v; }
