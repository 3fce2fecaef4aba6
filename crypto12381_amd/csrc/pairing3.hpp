// Optimal-ate pairing with THREE lanes per pairing.
//
// Why: with one pairing per lane the live state (an Fp12 accumulator, the G2 point, Karatsuba temporaries)
// is ~7 KB per lane and lives in private memory; measured on MI355X that kernel moves ~2.5 MB of HBM traffic
// per pairing and is bandwidth-bound (profiles/r01_pmc_summary_*).  Fp12 = Fp4[w]/(w^3 - s) has exactly three
// Fp4 coefficients, so a TRIPLE of adjacent lanes holds one pairing: lane role 0/1/2 owns coefficient a/b/c
// of every Fp12 value (56 dwords instead of 168) and coordinate X/Y/Z of the running G2 point.  Karatsuba
// over Fp4 needs six Fp4 products — two per lane — and the operands/results that cross coefficients travel
// through wavefront shuffles (ds_bpermute), not memory.  A 64-lane wavefront carries 21 pairings (lane 63
// idles along).  All role-dependent choices are selects, so the three roles run one instruction stream.
//
// Same mathematics as pairing.hpp (the reference's PAIR_ate / PAIR_fexp, pair_BLS12381.cpp:425-505, 629-755):
// the GT element — and its canonical bytes — are identical.
#pragma once
#include "fp12.hpp"
#include "g2.hpp"
#include "pairing.hpp"
#include "msm.hpp"

namespace c12381 {

struct tri { int role; int base; };      // role 0,1,2 = coefficient a,b,c; base = lane of role 0 in the wavefront

#if defined(__HIP_DEVICE_COMPILE__)
C12381_HD void tri_fetch_fp(fp& out, const fp& v, int src_role, const tri& t) {
#if defined(C12381_MICROBENCH_NO_SHUFFLE)          // csrc/microbench/pair_routines.hip only: what do the cross-lane moves cost?
    out = v; (void)src_role; (void)t;
#else
    int src = t.base + src_role;
    src = src > 63 ? 63 : src;
#pragma unroll
    for (int i = 0; i < NL; ++i) out.l[i] = __shfl(v.l[i], src, 64);
#endif
}
C12381_HD int tri_fetch_int(int v, int src_role, const tri& t) {
    int src = t.base + src_role;
    src = src > 63 ? 63 : src;
    return __shfl(v, src, 64);
}
#else
// host simulation: three threads per pairing exchange through a barrier-protected mailbox (tests/host_sim/sim.cpp)
void c12381_tri_exchange(void* out, const void* in, size_t bytes, int src_role, const tri& t);
inline void tri_fetch_fp(fp& out, const fp& v, int src_role, const tri& t) { fp tmp; c12381_tri_exchange(&tmp, &v, sizeof(fp), src_role, t); out = tmp; }
inline int tri_fetch_int(int v, int src_role, const tri& t) { int tmp; c12381_tri_exchange(&tmp, &v, sizeof(int), src_role, t); return tmp; }
#endif
C12381_HD void tri_fetch_fp2(fp2& out, const fp2& v, int s, const tri& t) { tri_fetch_fp(out.a, v.a, s, t); tri_fetch_fp(out.b, v.b, s, t); }
C12381_HD void tri_fetch_fp4(fp4& out, const fp4& v, int s, const tri& t) { tri_fetch_fp2(out.a, v.a, s, t); tri_fetch_fp2(out.b, v.b, s, t); }
C12381_HD int tri_next(const tri& t) { return t.role == 2 ? 0 : t.role + 1; }
C12381_HD int tri_prev(const tri& t) { return t.role == 0 ? 2 : t.role - 1; }

// A pointer or integer that is the same on every lane of the wavefront, as the compiler cannot know it of a function argument:
// the copy lives in SGPRs, which survive the out-of-line calls of the hot loops — an argument in a VGPR is clobbered by every
// call (the big routines save nothing) and is reloaded from private memory before the next one, a load whose latency is exposed.
#if defined(__HIP_DEVICE_COMPILE__)
template <class T> C12381_HD T* wave_uniform(T* p) {
    const uint64_t v = (uint64_t)(uintptr_t)p;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return (T*)(uintptr_t)(((uint64_t)hi << 32) | lo);
}
C12381_HD int wave_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
#else
template <class T> C12381_HD T* wave_uniform(T* p) { return p; }
C12381_HD int wave_uniform(int v) { return v; }
#endif

// ------------------------------------------------------------------ the per-lane LDS slot
// The running Fp12 coefficient of the Miller loop and of the exponentiations lives in one 224-byte LDS slot per lane (k_pair3.hip).
// The out-of-line routines below that end in _h take THAT slot: they move it with explicit LDS instructions (ds_read_b128 /
// ds_write_b128 through an address-space-3 pointer).  Through a plain reference the same accesses compile to flat_load / flat_store,
// which the hardware completes out of order — every wait on one of them is a full vmcnt(0) + lgkmcnt(0) drain that also waits
// for the wavefront's pending private-memory stores (MI355X_MICROARCH.md, s_waitcnt).
// slot_park / slot_unpark are the volatile forms: a value parked in the slot is really written and really read back instead of
// being kept alive in 56 registers across an Fp4 product (f12t_mul_h).
// C12381_PHASE(): a scheduling fence.  The build schedules for instruction-level parallelism (build.py), and two independent Fp4
// products in one routine are exactly what such a scheduler interleaves — doubling the live registers and spilling hundreds of
// dwords.  The fence keeps the phases of a routine apart (no instruction crosses it), as a call boundary used to.
// C12381_FAIR_SHARE(i, slot), once per iteration of the long loops, in the kernels that ask for it (slot_fair_set(slot, k): k of every 16 iterations,
// spread evenly): the wavefront in an ODD hardware slot of its SIMD raises its issue priority in those iterations.  A SIMD serves the older of its
// two wavefronts first whenever both can issue: the one in slot 0 ran a whole pairing in 15.0 M cycles, its partner in 37.7 M (profiles/r04_ab_fair_share.txt).
//   k = 8 in the PLAIN grids (every other iteration): in a launch that just fills the machine the younger wavefront otherwise finishes alone, at the issue
//         efficiency of a single wavefront — 2048 wavefronts of pairings 11.87 -> 11.35 ms.
//   k = 4 in the work-queue kernels of the Miller loop and of the final exponentiation alone (round 5): there the younger wavefront's one whole group
//         spans the launch and the older ones run out of queue tasks 13 % before it ends (profiles/r05_queue_wave_stats.txt); a quarter of the
//         iterations shifts enough issue slots to the younger one that both end together: Miller loop -1.6 ... -2.6 %, final exponentiation -1.7 %
//         (profiles/r05_ab_queue_fair_share.txt).  The cliff is close: at 5 of 16 the pairing kernel loses 2 %, at 6 of 16 everything 5-13 % — a task
//         holder that is held back stalls the wavefronts waiting for its hand-over (k = 8 there: 2^16 pairings 17.5 -> 20.3 ms, round 4) — so the
//         pairing kernels (gain 0.4 %) stay at k = 0.
#if defined(__HIP_DEVICE_COMPILE__)
// (the conditions as scalars: on per-lane values the two branches become two exec-masked regions and BOTH s_setprio execute)
#define C12381_FAIR_SHARE(i, slot) do { if ((__builtin_amdgcn_s_getreg((3 << 11) | 4) & 1u) && __builtin_amdgcn_readfirstlane(slot_fair(slot)) != 0) { \
        if ((__builtin_bitreverse32((unsigned)__builtin_amdgcn_readfirstlane((int)(i))) >> 28) < (unsigned)__builtin_amdgcn_readfirstlane(slot_fair(slot))) \
            __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); } } while (0)
#else
#define C12381_FAIR_SHARE(i, slot) do { } while (0)
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define C12381_PHASE() __builtin_amdgcn_sched_barrier(0)
#else
#define C12381_PHASE() do { } while (0)
#endif
#if defined(__HIP_DEVICE_COMPILE__)
typedef int32_t c12381_v4i __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) c12381_v4i c12381_lds_v4i;
static_assert(sizeof(fp4) == 14 * 16, "fp4 is 14 rows of 16 bytes on the device");
template <class P> C12381_HD void slot_rd(fp4& r, P p) {
    int32_t* w = reinterpret_cast<int32_t*>(&r);
#pragma unroll
    for (int i = 0; i < 14; ++i) { const c12381_v4i v = p[i]; w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w; }
}
template <class P> C12381_HD void slot_wr(P p, const fp4& r) {
    const int32_t* w = reinterpret_cast<const int32_t*>(&r);
#pragma unroll
    for (int i = 0; i < 14; ++i) { c12381_v4i v; v.x = w[4 * i]; v.y = w[4 * i + 1]; v.z = w[4 * i + 2]; v.w = w[4 * i + 3]; p[i] = v; }
}
C12381_HD void slot_load(fp4& r, const fp4& slot) { slot_rd(r, (const c12381_lds_v4i*)(&slot)); }
C12381_HD void slot_store(fp4& slot, const fp4& r) { slot_wr((c12381_lds_v4i*)(&slot), r); }
C12381_HD void slot_unpark(fp4& r, const fp4& slot) { slot_rd(r, (const volatile c12381_lds_v4i*)(&slot)); }
C12381_HD void slot_park(fp4& slot, const fp4& r) { slot_wr((volatile c12381_lds_v4i*)(&slot), r); }
// the last row of the lane's pair_slot (pad[2]): in how many of every 16 iterations the younger wavefront raises its priority (C12381_FAIR_SHARE); every kernel that declares slots sets it
typedef __attribute__((address_space(3))) int32_t c12381_lds_i32;
C12381_HD void slot_fair_set(fp4& slot, int on) { ((volatile c12381_lds_i32*)(&slot))[72] = on; }
C12381_HD int slot_fair(const fp4& slot) { return ((const volatile c12381_lds_i32*)(&slot))[72]; }
// one Fp2 half of the slot's Fp4 (half 0 = .a: rows 0..6, half 1 = .b: rows 7..13); `half` may differ from lane to lane (an LDS
// address is per lane anyway): a role-dependent placement of two results costs an address, not 56 selects
static_assert(sizeof(fp2) == 7 * 16, "an fp2 is 7 rows of 16 bytes");
C12381_HD void slot_load_half(fp2& r, const fp4& slot, int half) {
    const c12381_lds_v4i* p = (const c12381_lds_v4i*)(&slot) + 7 * half;
    int32_t* w = reinterpret_cast<int32_t*>(&r);
#pragma unroll
    for (int i = 0; i < 7; ++i) { const c12381_v4i v = p[i]; w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w; }
}
C12381_HD void slot_store_half(fp4& slot, int half, const fp2& r) {
    c12381_lds_v4i* p = (c12381_lds_v4i*)(&slot) + 7 * half;
    const int32_t* w = reinterpret_cast<const int32_t*>(&r);
#pragma unroll
    for (int i = 0; i < 7; ++i) { c12381_v4i v; v.x = w[4 * i]; v.y = w[4 * i + 1]; v.z = w[4 * i + 2]; v.w = w[4 * i + 3]; p[i] = v; }
}
#else
C12381_HD void slot_load(fp4& r, const fp4& slot) { r = slot; }
C12381_HD void slot_store(fp4& slot, const fp4& r) { slot = r; }
C12381_HD void slot_unpark(fp4& r, const fp4& slot) { r = slot; }
C12381_HD void slot_park(fp4& slot, const fp4& r) { slot = r; }
C12381_HD void slot_fair_set(fp4&, int) {}
C12381_HD int slot_fair(const fp4&) { return 0; }
C12381_HD void slot_load_half(fp2& r, const fp4& slot, int half) { r = half ? slot.b : slot.a; }
C12381_HD void slot_store_half(fp4& slot, int half, const fp2& r) { (half ? slot.b : slot.a) = r; }
#endif

// The slot as the kernels allocate it: the Fp4, then one Fp of the same lane — the affine G1 coordinate this lane's role needs
// in the doubling step (px for role 0, py for roles 1 and 2), kept here for the whole Miller loop so that it is not read from
// private memory every iteration.  304-byte stride = 76 dwords: 16 lanes x 16 bytes hit 64 distinct banks (76 = 12 mod 64).
// 256 lanes x 304 B = 76 KB per workgroup, two workgroups per CU (of 160 KB).
struct alignas(16) pair_slot { fp4 v; fp psel; int32_t pad[6]; };
#if defined(__HIP_DEVICE_COMPILE__)
static_assert(sizeof(pair_slot) == 304, "pair_slot stride");
// The Fp4 in the slot of the lane that holds role `src_role` of this lane's triple, read straight from LDS: the lanes of a triple are
// adjacent lanes of one wavefront and their slots adjacent records (19 rows of 16 bytes apart), a wavefront's LDS accesses are served in
// program order, and the value was stored by the same instruction stream earlier — 14 ds_read_b128 where a register shuffle of the
// same Fp4 is 56 ds_bpermute_b32 (24 issue cycles each, profiles/r03_issue_mix.txt).  The slot must be an element of a pair_slot array
// indexed by the lane (every kernel's is).
C12381_HD void slot_load_role(fp4& r, const fp4& slot, int src_role, const tri& t) {
    const int delta = t.base == 63 ? 0 : src_role - t.role;        // lane 63 (no triple of its own, results discarded) stays inside its own record
    slot_rd(r, (const c12381_lds_v4i*)(&slot) + 19 * delta);
}
#else
// host simulation: the three threads of a triple exchange their slot contents through the mailbox
inline void slot_load_role(fp4& r, const fp4& slot, int src_role, const tri& t) { fp4 tmp; c12381_tri_exchange(&tmp, &slot, sizeof(fp4), src_role, t); r = tmp; }
#endif
#if defined(__HIP_DEVICE_COMPILE__)
C12381_HD void slot_psel_store(fp4& slot, const fp& v) {       // 4 rows: 14 dwords + 2 of the pad
    c12381_lds_v4i* p = (c12381_lds_v4i*)(&slot) + 14;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        c12381_v4i w;
        w.x = v.l[4 * i]; w.y = v.l[4 * i + 1]; w.z = i < 3 ? v.l[(4 * i + 2) % NL] : 0; w.w = i < 3 ? v.l[(4 * i + 3) % NL] : 0;
        p[i] = w;
    }
}
C12381_HD void slot_psel_load(fp& v, const fp4& slot) {
    const volatile c12381_lds_v4i* p = (const volatile c12381_lds_v4i*)(&slot) + 14;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const c12381_v4i w = p[i];
        v.l[4 * i] = w.x; v.l[4 * i + 1] = w.y;
        if (i < 3) { v.l[4 * i + 2] = w.z; v.l[4 * i + 3] = w.w; }
    }
}
// The register file of miller3_iter's interface: the running point coordinate (28 dwords), the iteration's flags and the LDS
// address of the lane's slot as ONE 32-dword vector — it is argument and return value, travels in v0..v31 both ways and stays
// there from one iteration to the next (a 112-byte struct would be passed through private memory by the calling convention,
// and a 28-element vector is legalised through a stack temporary).
typedef int32_t miller3_regs __attribute__((ext_vector_type(32)));
typedef __attribute__((address_space(3))) fp4 c12381_lds_fp4;
C12381_HD miller3_regs m3r_make(const fp2& tc, int info, int slot) {       // one build: no chain of partial vectors
    const miller3_regs r = {tc.a.l[0], tc.a.l[1], tc.a.l[2], tc.a.l[3], tc.a.l[4], tc.a.l[5], tc.a.l[6], tc.a.l[7], tc.a.l[8], tc.a.l[9],
                            tc.a.l[10], tc.a.l[11], tc.a.l[12], tc.a.l[13], tc.b.l[0], tc.b.l[1], tc.b.l[2], tc.b.l[3], tc.b.l[4], tc.b.l[5],
                            tc.b.l[6], tc.b.l[7], tc.b.l[8], tc.b.l[9], tc.b.l[10], tc.b.l[11], tc.b.l[12], tc.b.l[13], info, slot, 0, 0};
    return r;
}
C12381_HD miller3_regs m3r_pack(const fp2& tc, fp4& F, int info) { return m3r_make(tc, info, (int32_t)(uint32_t)(uintptr_t)(c12381_lds_fp4*)(&F)); }
C12381_HD miller3_regs m3r_with_tc(const miller3_regs& r, const fp2& tc) { return m3r_make(tc, r[28], r[29]); }
C12381_HD void m3r_tc(fp2& tc, const miller3_regs& r) {
#pragma unroll
    for (int i = 0; i < NL; ++i) { tc.a.l[i] = r[i]; tc.b.l[i] = r[NL + i]; }
}
C12381_HD int m3r_info(const miller3_regs& r) { return r[28]; }
C12381_HD fp4& m3r_slot(const miller3_regs& r) { return *(fp4*)(c12381_lds_fp4*)(uintptr_t)(uint32_t)r[29]; }
#else
C12381_HD void slot_psel_store(fp4& slot, const fp& v) { reinterpret_cast<pair_slot*>(&slot)->psel = v; }
C12381_HD void slot_psel_load(fp& v, const fp4& slot) { v = reinterpret_cast<const pair_slot*>(&slot)->psel; }
struct miller3_regs { fp2 tc; fp4* F; int info; };
C12381_HD miller3_regs m3r_pack(const fp2& tc, fp4& F, int info) { miller3_regs r; r.tc = tc; r.F = &F; r.info = info; return r; }
C12381_HD miller3_regs m3r_with_tc(const miller3_regs& r, const fp2& tc) { miller3_regs q = r; q.tc = tc; return q; }
C12381_HD void m3r_tc(fp2& tc, const miller3_regs& r) { tc = r.tc; }
C12381_HD int m3r_info(const miller3_regs& r) { return r.info; }
C12381_HD fp4& m3r_slot(const miller3_regs& r) { return *r.F; }
#endif

// ------------------------------------------------------------------ inlined Fp4 cores (operands stay in registers)
// Reductions that absorb their linear terms (fp_reduce_cols_inj, round 4) are used where they measured faster, one session, every digest equal
// (profiles/r04_ab_injection_switches.txt): the line product against a normalised table entry (f12t_mul_line1_core, BBS+ -0.7 %), the cyclotomic
// squaring (f12t_usqr3_h: fexp -3.7 %, pairing -3.4 %) and the G1 formulas (g1.hpp: G1 -0.6 %, MSM -0.8 %).  The injected forms of the Fp4
// product / squaring and of the generic line product issued 1-3 % fewer instructions but 2-4 % MORE multiply-adds and spilled more in the fused
// Miller iteration (Miller loop +1.2 %, pairing +1.4 % with all three): measured, not kept in the tree.
C12381_HD void fp4_mul_core(fp4& w, const fp4& x, const fp4& y) {
    fp2 t1, t2, t3, t4;
    fp2_mul(t1, x.a, y.a);
    fp2_mul(t2, x.b, y.b);
    fp2_add(t3, y.b, y.a);
    fp2_add(t4, x.b, x.a);
    fp2_mul(t4, t4, t3);
    fp2_sub(t4, t4, t1);
    fp2_sub(t4, t4, t2);
    fp2_mul_ip(t3, t2);
    fp2_add(t3, t3, t1);
    fp2_norm1(w.b, t4);
    fp2_norm1(w.a, t3);
}
// the same without the final carry round (limbs up to 3 * 2^28): for a product that only enters a difference which is
// carried afterwards (the Karatsuba middle term)
C12381_HD void fp4_mul_core_raw(fp4& w, const fp4& x, const fp4& y) {
    fp2 t1, t2, t3, t4;
    fp2_mul(t1, x.a, y.a);
    fp2_mul(t2, x.b, y.b);
    fp2_add(t3, y.b, y.a);
    fp2_add(t4, x.b, x.a);
    fp2_mul(t4, t4, t3);
    fp2_sub(t4, t4, t1);
    fp2_sub(w.b, t4, t2);
    fp2_mul_ip(t3, t2);
    fp2_add(w.a, t3, t1);
}

// ------------------------------------------------------------------ Fp12 arithmetic on a triple
// combine step shared by product and square: given this lane's z = x_r y_r and e = x_r y_{r+1} + x_{r+1} y_r,
//   w_a = z_a + s e_b,   w_b = e_a + s z_c,   w_c = e_c + z_b
C12381_HD void f12t_combine(fp4& w, const fp4& z, const fp4& zn, const fp4& e, const tri& t) {
    // role a: z + s e_b;  role b: e_a + s z_c (= s zn);  role c: e + z_b   ==   P + (role c ? Q : s Q)
    fp4 v1, v2, P, Q, sQ, r;
    const int s1 = t.role == 0 ? tri_next(t) : (t.role == 1 ? tri_prev(t) : t.role);
    tri_fetch_fp4(v1, e, s1, t);                         // a <- e_b, b <- e_a
    const int s2 = t.role == 2 ? tri_prev(t) : t.role;
    tri_fetch_fp4(v2, z, s2, t);                         // c <- z_b
    fp4_select(P, t.role == 0, z, v1); fp4_select(P, t.role == 2, e, P);
    fp4_select(Q, t.role == 0, v1, zn); fp4_select(Q, t.role == 2, v2, Q);
    fp4_times_i(sQ, Q);
    fp4_select(Q, t.role == 2, Q, sQ);
    fp4_add(r, P, Q);
    fp4_norm1(w, r);
}
// One Fp4 product as a real call: an Fp4 product alone fills the 256-VGPR budget (two operands, four Fp2 temporaries,
// the column accumulators), so keeping a second pair of operands alive across it costs ~600 single-dword spills;
// two calls exchange 2 x 56 dwords through memory instead.
C12381_HDN void fp4_mul_call(fp4& w, const fp4& x, const fp4& y) { fp4_mul_core(w, x, y); }
// w = x * y  (FP12_mul fp12_BLS12381.cpp:246-299).  Two Fp4 products per lane.  w may alias x or y.
C12381_HDN void f12t_mul(fp4& w, const fp4& x, const fp4& y, const tri& t) {
    fp4 z, zc, zn, e;
    {
        fp4 xn, yn, sx, sy;
        tri_fetch_fp4(xn, x, tri_next(t), t);
        tri_fetch_fp4(yn, y, tri_next(t), t);
        fp4_addn(sx, x, xn); fp4_addn(sy, y, yn);
        fp4_mul_call(zc, sx, sy);
    }
    fp4_mul_call(z, x, y);
    tri_fetch_fp4(zn, z, tri_next(t), t);
    fp4_sub(e, zc, z); fp4_sub(e, e, zn); fp4_norm1(e, e);
    f12t_combine(w, z, zn, e, t);
}
// H <- H * y for the value in this lane's LDS slot (y: any memory, read only).  Same Karatsuba as f12t_mul, scheduled so that
// NOTHING is stored to private memory: (x + x')(y + y') is formed first and parked in the slot while x y takes the whole register
// budget (x has been read back from the slot by then), and read back for the combination.  f12t_mul exchanges
// its operands and both products with the out-of-line fp4_mul_call through private memory — four dependent store -> load round
// trips to HBM per product (measured: 64.7 K cycles per call against 37.7 K of issue, profiles/r02_pair_routines_2waves.txt).
// y is ALWAYS an object in the caller's private memory (a local of the exponentiation routines / the kernels): said explicitly — through the
// generic reference the two reads of y are flat loads, whose every wait drains vmcnt AND lgkmcnt (the routine's LDS traffic with it)
#if defined(__HIP_DEVICE_COMPILE__)
C12381_HD void fp4_load_private(fp4& r, const fp4* p) {
    typedef __attribute__((address_space(5))) const c12381_v4i priv_v4i;
    priv_v4i* q = (priv_v4i*)(const void*)p;
    int32_t* w = reinterpret_cast<int32_t*>(&r);
#pragma unroll
    for (int i = 0; i < 14; ++i) { const c12381_v4i v = q[i]; w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w; }
}
#else
C12381_HD void fp4_load_private(fp4& r, const fp4* p) { r = *p; }
#endif
C12381_HDN void f12t_mul_h(fp4& H, const fp4& y, const tri& t) {
    fp4 z, zc, zn, e, w;
    {   // (x + x')(y + y') first: its operands are sums, so x and y need not stay in registers while it runs
        fp4 sx, sy;
        {
            fp4 x, yv, xn, yn;
            slot_load(x, H);
            fp4_load_private(yv, &y);
            slot_load_role(xn, H, tri_next(t), t);
            tri_fetch_fp4(yn, yv, tri_next(t), t);
            fp4_addn(sx, x, xn); fp4_addn(sy, yv, yn);
        }
        C12381_PHASE();
        fp4_mul_core_raw(zc, sx, sy);                      // un-normalised: only enters e
        C12381_PHASE();
    }
    {   // x y: x comes back from the slot, which then takes zc; y is read again through a pointer the compiler cannot match
        // with the first read (it would otherwise keep all of y alive across the first product)
        fp4 x, yv;
        slot_unpark(x, H);
        slot_park(H, zc);
        const fp4* yp = &y;
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(yp));
#endif
        fp4_load_private(yv, yp);
        C12381_PHASE();
        fp4_mul_core(z, x, yv);
        C12381_PHASE();
    }
    slot_unpark(zc, H);
    tri_fetch_fp4(zn, z, tri_next(t), t);
    fp4_sub(e, zc, z); fp4_sub(e, e, zn); fp4_norm1(e, e);
    f12t_combine(w, z, zn, e, t);
    slot_store(H, w);
}
// w = x^2 (FP12_sqr :190-238 as six squarings: z_r = x_r^2, (x_r + x_{r+1})^2).
// Where the value whose neighbour coefficient is needed sits in the LDS slots, the neighbour's is read straight from ITS slot
// (slot_load_role: 14 ds_read_b128) instead of being shuffled out of registers (56 ds_bpermute_b32) — measured as neutral, kept for the shorter
// stream (profiles/r04_ab_slot_neighbour.txt).
template <bool FROM_SLOT>
C12381_HD void f12t_sqr_body_t(fp4& w, const fp4& x, const fp4& H, const tri& t) {
    fp4 xn, z, zc, zn, e, sx;
    if (FROM_SLOT) slot_load_role(xn, H, tri_next(t), t); else tri_fetch_fp4(xn, x, tri_next(t), t);
    fp4_sqr_core(z, x);
    fp4_add(sx, x, xn);                                    // limbs < 2^29 + slack: within the operand bound of the Fp2 products (host simulation asserts it)
    fp4_sqr_core_raw(zc, sx);                           // un-normalised: zc only enters e, which gets its own carry round
    tri_fetch_fp4(zn, z, tri_next(t), t);
    fp4_sub(e, zc, z); fp4_sub(e, e, zn); fp4_norm1(e, e);
    f12t_combine(w, z, zn, e, t);
}
C12381_HD void f12t_sqr_body(fp4& w, const fp4& x, const tri& t) { f12t_sqr_body_t<false>(w, x, x, t); }
C12381_HDN void f12t_sqr(fp4& w, const fp4& x, const tri& t) { fp4 xv = x, r; f12t_sqr_body(r, xv, t); w = r; }      // w may alias x
C12381_HDN void f12t_sqr_h(fp4& H, const tri& t) { fp4 x, r; slot_load(x, H); f12t_sqr_body_t<true>(r, x, H, t); slot_store(H, r); }
// Granger-Scott unitary squaring (FP12_usqr :147-186): one Fp4 squaring per lane.
//   w_a = 3 xa^2 - 2 conj(xa),  w_b = 3 s xc^2 + 2 conj(xb),  w_c = 3 xb^2 - 2 conj(xc)
C12381_HD void f12t_usqr_tail(fp4& w, const fp4& q, const fp4& x, bool reduce, const tri& t) {
    fp4 qq, sq, three, lin, c1, c2, r;
    const int src = t.role == 0 ? t.role : (t.role == 1 ? tri_next(t) : tri_prev(t));
    tri_fetch_fp4(qq, q, src, t);
    fp4_times_i(sq, qq);
    fp4_select(sq, t.role == 1, sq, qq);
    fp4_norm1(qq, sq);
    fp4_add(three, qq, qq); fp4_add(three, three, qq);
    fp4_conj(c1, x); fp4_nconj(c2, x);
    fp4_select(lin, t.role == 1, c1, c2);
    fp4_add(lin, lin, lin);
    fp4_add(r, three, lin);
    if (reduce) fp4_weak_reduce(w, r); else fp4_norm1(w, r);
}
C12381_HD void f12t_usqr_body(fp4& w, const fp4& x, bool reduce, const tri& t) {
    fp4 q;
    fp4_sqr_core_raw(q, x);                                // un-normalised: one carry round after the select in the tail
    f12t_usqr_tail(w, q, x, reduce, t);
}
C12381_HDN void f12t_usqr(fp4& w, const fp4& x, bool reduce, const tri& t) { fp4 xv = x, r; f12t_usqr_body(r, xv, reduce, t); w = r; }
// slot form: x is NOT kept in registers across the squaring — the linear part reads it from the slot again (14 LDS reads instead
// of 56 live registers: with them the routine spilled 7 dwords a call, 315 calls a pairing)
C12381_HDN void f12t_usqr_h(fp4& H, bool reduce, const tri& t) {
    fp4 q, x, r;
    slot_load(x, H);
    fp4_sqr_core_raw(q, x);
    C12381_PHASE();
    slot_unpark(x, H);
    f12t_usqr_tail(r, q, x, reduce, t);
    slot_store(H, r);
}
// ------------------------------------------------------------------ the cyclotomic squaring of the exponentiation loops (round 4)
// Scaled representation: the slot holds y = 3 x.  Granger-Scott's x <- 3 x^2 - 2 conj(x) becomes  y <- y^2 - 2 conj(y)  (an identity of
// polynomials, so it reproduces FP12_usqr on ANY input, unitary or not — f12t_pow_generic relies on that): no tripling.  The scale
// survives a product with an unscaled operand (3h * a = 3 (h a)); the exponentiation routines enter with f12t_scale3_h and leave
// with f12t_unscale3_h.
// Per lane, with ys the coefficient this lane's OUTPUT needs squared (role a: its own; roles b and c: each other's, read from the
// partner's slot) and y its own coefficient, u = ys.a, v = ys.b in Fp2, t3 = u v, Q0 = u^2 + xi v^2 = (u + v)(u + xi v) - (1 + xi) t3:
//   roles a, c:  w.a = Q0 - 2 y.a,       w.b = 2 t3 + 2 y.b
//   role b:      w.a = xi 2 t3 + 2 y.a,  w.b = Q0 - 2 y.b          (w_b = 3 s c^2 + 2 conj(b): s (Q0 + 2 t3 s) = 2 xi t3 + Q0 s)
// Both Fp2 values come out of reductions with their linear terms injected (fp_reduce_cols_inj), already normalised:
//   S = t3 + mu y.b - q p             mu = 1 (roles a, c) | 0 (role b);   2 S is w.b of roles a, c
//   V = (u + v)(u + xi v) - (1 + xi)(S - mu y.b) + lambda - q' p,   lambda = -2 y.a | -2 y.b:  w.a of roles a, c, w.b of role b
// (the mu y.b inside S cancels exactly in V: the identity is one of integers), and the remaining value — 2 S, or 2 (xi S + y.a - q'' p) on
// role b — is one exactly carried linear combination (fp_lincomb3p).  Every "- q p" takes the multiple of p nearest to the linear
// terms out again, so the value bound of y stays below 5.2 for ever: no weak reduction, no carry round, no select, no register
// shuffle (the old form: 3 q +- 2 conj(x) assembled from lazy sums behind two carry rounds, a weak reduction every second call, 56
// ds_bpermute and ~170 selects; 3 906 -> ~3 450 vector instructions per call, 315 calls per pairing).
C12381_HD void fp4_scale3(fp4& r, const fp4& x) {          // 3 x, weakly reduced: |value| < 1.6 p whatever came in
    fp4 t;
    fp4_add(t, x, x); fp4_add(t, t, x);
    fp4_weak_reduce(r, t);
}
C12381_HD void fp4_unscale3(fp4& r, const fp4& y) {        // y / 3
    fp inv3;
    fp_set_const(inv3, FP_INV3);
    fp_mul(r.a.a, y.a.a, inv3); fp_mul(r.a.b, y.a.b, inv3); fp_mul(r.b.a, y.b.a, inv3); fp_mul(r.b.b, y.b.b, inv3);
}
C12381_HD void f12t_scale3_h(fp4& H) { fp4 x, y; slot_load(x, H); fp4_scale3(y, x); slot_store(H, y); }
C12381_HD void f12t_unscale3_h(fp4& H) { fp4 x, y; slot_load(y, H); fp4_unscale3(x, y); slot_store(H, x); }
#if defined(__HIP_DEVICE_COMPILE__)
C12381_HD int32_t lane_opaque(int32_t v) { asm("" : "+v"(v)); return v; }     // a per-lane multiplier the optimiser must not fold into selects
#else
C12381_HD int32_t lane_opaque(int32_t v) { return v; }
#endif
C12381_HDN void f12t_usqr3_h(fp4& H, const tri& t) {
    const bool r1 = t.role == 1;
    // multipliers of the injected terms: constants in SGPRs, role-dependent ones in VGPRs (multiply-add operands, not selects)
    const int32_t c1 = fp_opaque_const(1), cm1 = fp_opaque_const(-1), cm2 = fp_opaque_const(-2);
    const int32_t mu = lane_opaque(r1 ? 0 : 1);
    const int32_t a1 = lane_opaque(r1 ? -2 : 2), a2 = lane_opaque(r1 ? 0 : -1), a3 = lane_opaque(r1 ? 0 : -2);
    const int32_t b1 = lane_opaque(r1 ? 0 : 1), b2 = lane_opaque(r1 ? -2 : 2), b3 = a3;
    fp2 S, V;
    fp2 t1, t2;
    {
        fp4 ys;
        slot_load_role(ys, H, t.role == 0 ? 0 : (r1 ? 2 : 1), t);
        C12381_BOUNDS({ const fp* c4[4] = {&ys.a.a, &ys.a.b, &ys.b.a, &ys.b.b};
                        for (const fp* c : c4) { if (c->vb > 200.0 || c->lb > 268435456.0 + 64) bounds_fail("f12t_usqr3_h input", c->vb, c->lb); } })
        {   // S = ys.a * ys.b + mu y.b - q p
            fp2 yb;
            slot_load_half(yb, H, 1);
            fp nab;
            fp_raw_neg(nab, ys.a.b);
            const int32_t nq_re = -fp_quot_top(mu * yb.a.l[NL - 1]), nq_im = -fp_quot_top(mu * yb.b.l[NL - 1]);
            fp_reduce_cols_inj(S.a, [&](int k, int64_t& acc) { fp_col_acc(acc, ys.a.a, ys.b.a, k); fp_col_acc(acc, nab, ys.b.b, k); },
                               [&](int i, int64_t& acc) { fp_inj(acc, yb.a, i, mu); fp_inj_p(acc, i, nq_re); });
            fp_reduce_cols_inj(S.b, [&](int k, int64_t& acc) { fp_col_acc(acc, ys.a.a, ys.b.b, k); fp_col_acc(acc, ys.a.b, ys.b.a, k); },
                               [&](int i, int64_t& acc) { fp_inj(acc, yb.b, i, mu); fp_inj_p(acc, i, nq_im); });
            // |mu y.b - q p| <= (0.5 + 2 / 106513 + 1e-5 |y.b| / p) p: the nearest multiple of p by the top limb (fp_quot_top)
            C12381_BOUNDS({ set_inj_bounds(S.a, ys.a.a.lb * ys.b.a.lb + ys.a.b.lb * ys.b.b.lb, ys.a.a.vb * ys.b.a.vb + ys.a.b.vb * ys.b.b.vb, 0.501,
                                           yb.a.lb + 8.0 * 268435456.0, "usqr3 S.a"); check_actual_vb(S.a, "usqr3 S.a value");
                            set_inj_bounds(S.b, ys.a.a.lb * ys.b.b.lb + ys.a.b.lb * ys.b.a.lb, ys.a.a.vb * ys.b.b.vb + ys.a.b.vb * ys.b.a.vb, 0.501,
                                           yb.b.lb + 8.0 * 268435456.0, "usqr3 S.b"); check_actual_vb(S.b, "usqr3 S.b value"); })
        }
        fp2 xv;
        fp2_add(t1, ys.a, ys.b);                               // u + v: limbs < 2^29
        fp2_mul_ip(xv, ys.b);
        fp2_add(xv, ys.a, xv);
        fp2_norm1(t2, xv);                                     // u + xi v, three terms: carried once
    }
    {   // V = t1 t2 - (2 + i)(S - mu y.b) + lambda - q p:   (2 + i)(a + b i) = (2a - b) + (a + 2b) i
        fp2 ya, yb;
        slot_load_half(ya, H, 0);
        slot_load_half(yb, H, 1);
        fp nt1b;
        fp_raw_neg(nt1b, t1.b);
        const int32_t top_re = cm2 * S.a.l[NL - 1] + S.b.l[NL - 1] + a1 * yb.a.l[NL - 1] + a2 * yb.b.l[NL - 1] + a3 * ya.a.l[NL - 1];
        const int32_t top_im = cm1 * S.a.l[NL - 1] + cm2 * S.b.l[NL - 1] + b1 * yb.a.l[NL - 1] + b2 * yb.b.l[NL - 1] + b3 * ya.b.l[NL - 1];
        const int32_t nq_re = -fp_quot_top(top_re), nq_im = -fp_quot_top(top_im);
        fp_reduce_cols_inj(V.a, [&](int k, int64_t& acc) { fp_col_acc(acc, t1.a, t2.a, k); fp_col_acc(acc, nt1b, t2.b, k); },
                           [&](int i, int64_t& acc) { fp_inj(acc, S.a, i, cm2); fp_inj(acc, S.b, i, c1); fp_inj(acc, yb.a, i, a1); fp_inj(acc, yb.b, i, a2);
                                                      fp_inj(acc, ya.a, i, a3); fp_inj_p(acc, i, nq_re); });
        fp_reduce_cols_inj(V.b, [&](int k, int64_t& acc) { fp_col_acc(acc, t1.a, t2.b, k); fp_col_acc(acc, t1.b, t2.a, k); },
                           [&](int i, int64_t& acc) { fp_inj(acc, S.a, i, cm1); fp_inj(acc, S.b, i, cm2); fp_inj(acc, yb.a, i, b1); fp_inj(acc, yb.b, i, b2);
                                                      fp_inj(acc, ya.b, i, b3); fp_inj_p(acc, i, nq_im); });
        // the injected sum minus the nearest multiple of p: eight unit multipliers' worth of low-limb error (8 / 106513) + the estimate's 1e-5
        C12381_BOUNDS({ const double inj_lb = 3.0 * (S.a.lb + S.b.lb) + 3.0 * (yb.a.lb + yb.b.lb) + 2.0 * (ya.a.lb + ya.b.lb) + 64.0 * 268435456.0;
                        set_inj_bounds(V.a, t1.a.lb * t2.a.lb + t1.b.lb * t2.b.lb, t1.a.vb * t2.a.vb + t1.b.vb * t2.b.vb, 0.502, inj_lb, "usqr3 V.a");
                        check_actual_vb(V.a, "usqr3 V.a value");
                        set_inj_bounds(V.b, t1.a.lb * t2.b.lb + t1.b.lb * t2.a.lb, t1.a.vb * t2.b.vb + t1.b.vb * t2.a.vb, 0.502, inj_lb, "usqr3 V.b");
                        check_actual_vb(V.b, "usqr3 V.b value"); })
        // the other output: 2 S on roles a, c;  2 (xi S + y.a - q p) on role b, xi S = (S.a - S.b) + (S.a + S.b) i
        const int32_t k2 = fp_opaque_const(2);
        const int32_t e_sb = lane_opaque(r1 ? -2 : 0), e_y = lane_opaque(r1 ? 2 : 0), f_sa = lane_opaque(r1 ? 2 : 0);
        const int32_t kq_re = -e_y * fp_quot_top(ya.a.l[NL - 1]), kq_im = -e_y * fp_quot_top(ya.b.l[NL - 1]);
        fp2 U;
        // |2 S| <= 2 VB(S);   role b: 2 (2 VB(S) + 0.501) with VB(S) of the uninjected form (mu = 0: the 0.501 of the declaration is not there)
        C12381_BOUNDS(const double sv = S.a.vb > S.b.vb ? S.a.vb : S.b.vb; const double vbu = r1 ? 2.0 * (2.0 * (sv - 0.501) + 0.501) : 2.0 * sv;)
#ifndef C12381_CHECK_BOUNDS
        const double vbu = 0;
#endif
        fp_lincomb3p(U.a, S.a, k2, S.b, e_sb, ya.a, e_y, kq_re, vbu);
        fp_lincomb3p(U.b, S.b, k2, S.a, f_sa, ya.b, e_y, kq_im, vbu);
        // placement by address: V is .a on roles a, c and .b on role b
        slot_store_half(H, r1 ? 1 : 0, V);
        slot_store_half(H, r1 ? 0 : 1, U);
    }
}
// FP12_conj :117-123
C12381_HD void f12t_conj(fp4& w, const fp4& x, const tri& t) {
    fp4 c1, c2;
    fp4_conj(c1, x); fp4_nconj(c2, x);
    fp4_select(w, t.role == 1, c2, c1);
}
C12381_HD void f12t_conj_h(fp4& H, const tri& t) { fp4 x, r; slot_load(x, H); f12t_conj(r, x, t); slot_store(H, r); }
// FP12_frob :867-880
C12381_HDN void f12t_frob(fp4& w, const fp4& x, const tri& t) {
    fp2 f, f2, f3, m;
    fp2_set_const(f, FROB_F_A, FROB_F_B);
    fp2_set_const(f2, FROB_F2_A, FROB_F2_B);
    fp2_set_const(f3, FROB_F3_A, FROB_F3_B);
    fp4 y, ym;
    fp4_frob(y, x, f3);
    fp2_select(m, t.role == 1, f, f2);
    fp4_pmul(ym, y, m);
    fp4_select(w, t.role == 0, y, ym);
}
// FP12_inv :627-664
C12381_HDN void f12t_inv(fp4& w, const fp4& x, const tri& t) {
    fp4 xn, xp, xa, xb, xc;
    tri_fetch_fp4(xn, x, tri_next(t), t);
    tri_fetch_fp4(xp, x, tri_prev(t), t);
    // role 0: (own,next,prev) = (a,b,c); role 1: (b,c,a); role 2: (c,a,b)
    fp4_select(xa, t.role == 0, x, xp); fp4_select(xa, t.role == 2, xn, xa);
    fp4_select(xb, t.role == 1, x, xn); fp4_select(xb, t.role == 2, xp, xb);
    fp4_select(xc, t.role == 2, x, xp); fp4_select(xc, t.role == 1, xn, xc);
    // f_0 = xa^2 - s xb xc,  f_1 = s xc^2 - xa xb,  f_2 = xb^2 - xa xc
    fp4 P, u, v, sq, cr, tmp, f;
    fp4_select(P, t.role == 0, xa, xb); fp4_select(P, t.role == 1, xc, P);
    fp4_select(u, t.role == 0, xb, xa);
    fp4_select(v, t.role == 1, xb, xc);
    fp4_sqr_core(sq, P);
    fp4_mul_core(cr, u, v);
    fp4_times_i(tmp, sq); fp4_norm1(tmp, tmp); fp4_select(sq, t.role == 1, tmp, sq);
    fp4_times_i(tmp, cr); fp4_norm1(tmp, tmp); fp4_select(cr, t.role == 0, tmp, cr);
    fp4_sub(f, sq, cr); fp4_norm1(f, f);
    // f3 = xa f0 + s (xc f1 + xb f2): this lane's term is P * f
    fp4 term, tn, tp, ta, tbc, f3, f3i;
    fp4_mul_core(term, P, f);
    tri_fetch_fp4(tn, term, tri_next(t), t);
    tri_fetch_fp4(tp, term, tri_prev(t), t);
    fp4_select(ta, t.role == 0, term, tp); fp4_select(ta, t.role == 2, tn, ta);
    fp4 o1, o2;
    fp4_select(o1, t.role == 0, tn, term);                 // the two terms that are not term_a
    fp4_select(o2, t.role == 1, tn, tp);
    fp4_add(tbc, o1, o2);
    fp4_times_i(tmp, tbc);
    fp4_add(f3, ta, tmp);
    fp4_norm1(f3, f3);
    fp4_inv(f3i, f3);
    fp4_mul_core(w, f, f3i);
}
// f *= line(l0, l1, l2)  (sparse M-type line, see fp12_mul_line).  One dense-by-sparse product per lane.
C12381_HD void f12t_mul_line_core(fp4& x, const fp2& l0, const fp2& l1, const fp2& l2, const tri& t) {
    fp4 la; la.a = l0; la.b = l1;
    fp4 p;
    fp2 q0, q1, qn0, qn1, ia, ib;
    fp2_mul(q0, x.a, l2);
    fp2_mul(q1, x.b, l2);
    fp4_mul_core_raw(p, x, la);                            // carried once, together with the terms added below
    tri_fetch_fp2(qn0, q0, tri_next(t), t);
    tri_fetch_fp2(qn1, q1, tri_next(t), t);
    // roles a, b: + ((1+i) qn0, (1+i) qn1);   role c: + ((1+i) qn1, qn0)
    fp2 pick;
    fp2_select(pick, t.role == 2, qn1, qn0);
    fp2_mul_ip(ia, pick);
    fp2_mul_ip(ib, qn1);
    fp2_select(ib, t.role == 2, qn0, ib);
    fp2_add(p.a, p.a, ia);
    fp2_add(p.b, p.b, ib);
    fp4_norm1(x, p);
}
// the same for a line whose s-coefficient is 1 (l1 = 1: the lines of a fixed G2 argument are stored divided by it, see
// miller_lines_precompute): x (l0 + s) = (xa l0 + (1+i) xb) + (xa + xb l0) s — two Fp2 products instead of Karatsuba's three.
// x passes through UNMULTIPLIED ((1+i) xb in .a, xa in .b): its value bound would double per line, so each reduction also takes the
// multiple of p nearest to its injected sum out again (fp_inj_p) — that replaces the weak reduction (~360 instructions) this routine
// used to end in.
C12381_HD void f12t_mul_line1_core(fp4& x, const fp2& l0, const fp2& l2, const tri& t) {
    const int32_t c1 = fp_opaque_const(1), cm1 = fp_opaque_const(-1);
    const bool r2 = t.role == 2;
    fp2 q0, q1, qn0, qn1;
    fp2_mul(q0, x.a, l2);
    fp2_mul(q1, x.b, l2);
    tri_fetch_fp2(qn0, q0, tri_next(t), t);
    tri_fetch_fp2(qn1, q1, tri_next(t), t);
    fp2 pick, other, wa, wb;
    fp2_select(pick, r2, qn1, qn0);
    fp2_select(other, r2, qn0, qn1);
    const int32_t o_cross = lane_opaque(r2 ? 0 : 1), o_ncross = lane_opaque(r2 ? 0 : -1);
    // .a = x.a l0 + xi x.b + xi pick - q p
    const int32_t ta_re = x.b.a.l[NL - 1] - x.b.b.l[NL - 1] + pick.a.l[NL - 1] - pick.b.l[NL - 1];
    const int32_t ta_im = x.b.a.l[NL - 1] + x.b.b.l[NL - 1] + pick.a.l[NL - 1] + pick.b.l[NL - 1];
    const int32_t nqa_re = -fp_quot_top(ta_re), nqa_im = -fp_quot_top(ta_im);
    // .b = x.b l0 + x.a + (other | xi other) - q p
    const int32_t tb_re = x.a.a.l[NL - 1] + other.a.l[NL - 1] + o_ncross * other.b.l[NL - 1];
    const int32_t tb_im = x.a.b.l[NL - 1] + other.b.l[NL - 1] + o_cross * other.a.l[NL - 1];
    const int32_t nqb_re = -fp_quot_top(tb_re), nqb_im = -fp_quot_top(tb_im);
    // the injected sums minus the nearest multiple of p: at most four unit multipliers of low-limb error (4 / 106513) + the estimate's 1e-5 per p
    fp2_mul_inj(wa, x.a, l0, [&](int i, int64_t& acc) { fp_inj(acc, x.b.a, i, c1); fp_inj(acc, x.b.b, i, cm1); fp_inj(acc, pick.a, i, c1); fp_inj(acc, pick.b, i, cm1);
                                                        fp_inj_p(acc, i, nqa_re); },
                [&](int i, int64_t& acc) { fp_inj(acc, x.b.a, i, c1); fp_inj(acc, x.b.b, i, c1); fp_inj(acc, pick.a, i, c1); fp_inj(acc, pick.b, i, c1);
                                           fp_inj_p(acc, i, nqa_im); },
                C12381_INJB(0.502, x.b.a.lb + x.b.b.lb + pick.a.lb + pick.b.lb + 64.0 * 268435456.0), C12381_INJB(0.502, x.b.a.lb + x.b.b.lb + pick.a.lb + pick.b.lb + 64.0 * 268435456.0));
    fp2_mul_inj(wb, x.b, l0, [&](int i, int64_t& acc) { fp_inj(acc, x.a.a, i, c1); fp_inj(acc, other.a, i, c1); fp_inj(acc, other.b, i, o_ncross); fp_inj_p(acc, i, nqb_re); },
                [&](int i, int64_t& acc) { fp_inj(acc, x.a.b, i, c1); fp_inj(acc, other.b, i, c1); fp_inj(acc, other.a, i, o_cross); fp_inj_p(acc, i, nqb_im); },
                C12381_INJB(0.502, x.a.a.lb + other.a.lb + other.b.lb + 64.0 * 268435456.0), C12381_INJB(0.502, x.a.b.lb + other.a.lb + other.b.lb + 64.0 * 268435456.0));
    x.a = wa; x.b = wb;
}
C12381_HDN void f12t_mul_line(fp4& x, const fp2& l0, const fp2& l1, const fp2& l2, const tri& t) { f12t_mul_line_core(x, l0, l1, l2, t); }
C12381_HDN void f12t_mul_line_h(fp4& H, const fp2& l0, const fp2& l1, const fp2& l2, const tri& t) {
    fp4 x;
    slot_load(x, H);
    f12t_mul_line_core(x, l0, l1, l2, t);
    slot_store(H, x);
}
// h <- a^x for unitary a, x < 0, computed IN the LDS slot h: the 63 squarings and 5 products move the running value with LDS
// instructions only; `a` is read from wherever it lives (private memory in the kernels), never written.
C12381_HDN void f12t_pow_x(fp4& h, const fp4& a, const tri& t) {
    {   // the running value is kept as y = 3 h (f12t_usqr3_h); a stays as it is: 3h * a = 3 (h a)
        fp4 av = a, y;
        fp4_scale3(y, av);
        slot_store(h, y);
    }
#pragma unroll 1
    for (int i = 62; i >= 0; --i) {
        C12381_FAIR_SHARE(i, h);
        f12t_usqr3_h(h, t);
        if ((BLS_X >> i) & 1ull) f12t_mul_h(h, a, t);
    }
    {
        fp4 y, x, c;
        slot_load(y, h);
        fp4_unscale3(x, y);
        f12t_conj(c, x, t);
        slot_store(h, c);
    }
}
// PAIR_fexp :629-755 as six steps (the work-queue kernels schedule them as separate tasks; state between steps: r, y1
// and — after step 4 — aux).  `h` is this lane's LDS slot: every product is h <- h * (operand in private memory).
//   0: easy part, y1 = r^3      1, 2: r <- r^(x-1)      3: r <- r^(x+p)      4: aux = r^x      5: r <- aux^x r^(p^2-1) y1
C12381_HDN void f12t_final_exp_step(int step, fp4& r, fp4& y1, fp4& aux, fp4& h, const tri& t) {
    fp4 t0;
    if (step == 0) {
        f12t_inv(t0, r, t);
        { fp4 c; f12t_conj(c, r, t); slot_store(h, c); }
        f12t_mul_h(h, t0, t);                                                 // conj(r) / r
        slot_load(r, h);
        f12t_frob(t0, r, t); f12t_frob(t0, t0, t);
        f12t_mul_h(h, t0, t);                                                 // ^(p^2 + 1)
        slot_load(r, h);
        f12t_usqr_h(h, false, t); f12t_mul_h(h, r, t);                        // r^3
        slot_load(y1, h);
    } else if (step <= 3) {
        f12t_pow_x(h, r, t);
        if (step == 3) f12t_frob(t0, r, t); else f12t_conj(t0, r, t);
        f12t_mul_h(h, t0, t);                                                 // r^(x-1) twice, then r^(x+p)
        slot_load(r, h);
    } else if (step == 4) {
        f12t_pow_x(h, r, t);
        slot_load(aux, h);                                                    // r^x
    } else {
        f12t_pow_x(h, aux, t);                                                // r^(x^2)
        f12t_frob(t0, r, t); f12t_frob(t0, t0, t);                            // r^(p^2)
        f12t_mul_h(h, t0, t);
        f12t_conj(t0, r, t);
        f12t_mul_h(h, t0, t);                                                 // ^(x^2+p^2-1)
        f12t_mul_h(h, y1, t);
        slot_load(r, h);
    }
}
C12381_HD void f12t_final_exp_ws(fp4& r, fp4& h, const tri& t) {
    fp4 y1, aux;
#pragma unroll 1
    for (int step = 0; step < 6; ++step) f12t_final_exp_step(step, r, y1, aux, h, t);
}
#if !defined(__HIP_DEVICE_COMPILE__)
// host simulation only: on the device the working slot has to be LDS (slot_load / slot_store)
inline void f12t_final_exp(fp4& r, const tri& t) {
    fp4 h;
    f12t_final_exp_ws(r, h, t);
}
#endif
C12381_HD void f12t_one(fp4& F, const tri& t);
// FP12_pow :736-774 on a triple, exponent e < 2^256 used AS GIVEN (the three lanes of a triple hold the same e): the
// reference's signed-digit ladder over (3e, e) with Granger-Scott squarings — a power only for unitary inputs, like
// there.  Triples hold different exponents, so both candidates are always formed and selected (fp12_pow_generic).
// H: the lane's slot, holds a on entry and the result on return.  The accumulator starts at 1 and the loop starts at the top
// bit position, so every triple of the wavefront runs the same 257 iterations: above an exponent's leading digit both steps
// are identities on 1 (usqr(1) = 1, 1 * 1 = 1), its leading digit is +1 (3e has its top bit one position above any bit of e)
// and turns the accumulator into a — from there on the sequence is FP12_pow's, for non-unitary inputs too.  Zero digits
// multiply by 1 (the value is unchanged; the bytes written are canonical either way).
// The pieces (the work-queue kernel runs the ladder in four tasks): the accumulator's start value 3 * 1, iterations hi .. lo of the ladder on the
// accumulator in the slot with the base `a` in private memory, and the way out (f12t_unscale3_h).
C12381_HD void f12t_pow_acc_init(fp4& H, const tri& t) {       // the accumulator in the scaled form 3 * acc (f12t_usqr3_h)
    fp4 one, y;
    f12t_one(one, t);
    fp4_scale3(y, one);
    slot_store(H, y);
}
C12381_HDN void f12t_pow_generic_range(fp4& H, const fp4& a, const uint32_t (&e)[8], int hi, int lo, const tri& t) {
    uint32_t e3[9];
    {
        uint64_t c = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) { c += (uint64_t)e[i] * 3u; e3[i] = (uint32_t)c; c >>= 32; }
        e3[8] = (uint32_t)c;
    }
    fp4 ac, one;
    f12t_conj(ac, a, t);
    f12t_one(one, t);
#pragma unroll 1
    for (int i = hi; i >= lo; --i) {
        C12381_FAIR_SHARE(i, H);
        f12t_usqr3_h(H, t);                                    // y <- y^2 - 2 conj(y) is FP12_usqr's polynomial on any input
        const int b3 = (int)((e3[i >> 5] >> (i & 31)) & 1u);
        const int b1 = i < 256 ? (int)((e[i >> 5] >> (i & 31)) & 1u) : 0;
        const int bt = b3 - b1;
        fp4 m;
        fp4_select(m, bt < 0, ac, a);
        fp4_select(m, bt == 0, one, m);
        f12t_mul_h(H, m, t);
    }
}
C12381_HD void f12t_pow_generic(fp4& H, const uint32_t (&e)[8], const tri& t) {
    fp4 a;
    slot_load(a, H);
    f12t_pow_acc_init(H, t);
    f12t_pow_generic_range(H, a, e, 257, 1, t);
    f12t_unscale3_h(H);
}
// ------------------------------------------------------------------ FP12_pow on the cyclotomic subgroup: fixed 4-bit windows
// x^(p^4 - p^2 + 1) = 1 — every pairing value: the easy part of the final exponentiation maps into that subgroup — makes the Granger-Scott
// squaring a true squaring and conj the inverse, so EVERY addition chain returns the field element FP12_pow :736-774 returns for (x, e),
// and the canonical bytes with it.  Only there: on any other input the reference's value depends on its own digit sequence, which
// f12t_pow_generic reproduces; the kernels test membership (f12t_is_cyclotomic, one product and four Frobenius maps) and take the windowed
// ladder only when every triple of the wavefront passes.  x = 0 passes the test as well and gives 0 (e != 0) or 1 (e = 0) on both routes.
//   table x^0 .. x^15 (14 products; `store(k, v)` / `load(r, k)`: one Fp4 per lane and entry, k may differ from lane to lane),
//   then 64 windows from the top: four squarings in the scaled form, one product with the digit's entry (x^0 = 1 for a zero digit, so all
//   triples run the same instructions) — 78 products against the 257 of the signed-digit ladder, the 256 squarings stay.
C12381_HD bool f12t_is_cyclotomic(fp4& H, const fp4& x, const tri& t) {
    fp4 f2, f4, r, d;
    f12t_frob(f2, x, t); f12t_frob(f2, f2, t);                 // x^(p^2)
    f12t_frob(f4, f2, t); f12t_frob(f4, f4, t);                // x^(p^4)
    slot_store(H, f4);
    f12t_mul_h(H, x, t);
    slot_load(r, H);
    fp4_sub(d, r, f2);
    const int mine = (fp_is_zero(d.a.a) & fp_is_zero(d.a.b) & fp_is_zero(d.b.a) & fp_is_zero(d.b.b)) ? 1 : 0;
    const int a = tri_fetch_int(mine, 0, t), b = tri_fetch_int(mine, 1, t), c = tri_fetch_int(mine, 2, t);
    return (a & b & c) != 0;
}
// H: the lane's slot (any content on entry, x^e on return); x: the base, in private memory.  In pieces for the work-queue kernel: the table,
// windows whi .. wlo (63 .. 0 from the top) on the accumulator in the slot, and f12t_pow_acc_init / f12t_unscale3_h around them.
template <class Store>
C12381_HD void f12t_pow_window_table(fp4& H, const fp4& x, const tri& t, Store store) {
    fp4 one, r;
    f12t_one(one, t);
    store(0, one); store(1, x);
    slot_store(H, x);
#pragma unroll 1
    for (int k = 2; k < 16; ++k) {
        f12t_mul_h(H, x, t);
        slot_load(r, H);
        store(k, r);
    }
}
template <class Load>
C12381_HD void f12t_pow_window_range(fp4& H, const uint32_t (&e)[8], int whi, int wlo, const tri& t, Load load) {
#pragma unroll 1
    for (int w = whi; w >= wlo; --w) {
        C12381_FAIR_SHARE(w, H);
        if (w != 63) { f12t_usqr3_h(H, t); f12t_usqr3_h(H, t); f12t_usqr3_h(H, t); f12t_usqr3_h(H, t); }      // the accumulator is 1 above the top window
        const int digit = (int)((e[w >> 3] >> ((w & 7) * 4)) & 15u);
        fp4 m;
        load(m, digit);
        f12t_mul_h(H, m, t);
    }
}
template <class Store, class Load>
C12381_HD void f12t_pow_window(fp4& H, const fp4& x, const uint32_t (&e)[8], const tri& t, Store store, Load load) {
    f12t_pow_window_table(H, x, t, store);
    f12t_pow_acc_init(H, t);
    f12t_pow_window_range(H, e, 63, 0, t, load);
    f12t_unscale3_h(H);
}
// FP12_isunity: every lane tests its own coefficient, the verdict is combined over the triple
C12381_HD bool f12t_is_one(const fp4& x, const tri& t) {
    fp d, one;
    fp_one(one);
    fp_sub(d, x.a.a, one);
    const bool first = t.role == 0 ? fp_is_zero(d) : fp_is_zero(x.a.a);
    const int mine = (first & fp_is_zero(x.a.b) & fp_is_zero(x.b.a) & fp_is_zero(x.b.b)) ? 1 : 0;
    const int a = tri_fetch_int(mine, 0, t), b = tri_fetch_int(mine, 1, t), c = tri_fetch_int(mine, 2, t);
    return (a & b & c) != 0;
}

// ------------------------------------------------------------------ Miller loop on a triple
// Doubling step: role 0/1/2 holds X/Y/Z of T in `tc`.  Three Fp2 products per lane (PAIR_double :40-78 +
// ECP2_dbl ecp2_BLS12381.cpp:358-409); the three line coefficients are then shared with all lanes.
C12381_HD void miller3_dbl_step_sel(fp2& tc, fp2& l0, fp2& l1, fp2& l2, const fp& sel, const tri& t);
C12381_HD void miller3_dbl_step_core(fp2& tc, fp2& l0, fp2& l1, fp2& l2, const fp& px, const fp& py, const tri& t) {
    fp sel;
    fp_select(sel, t.role == 0, px, py);
    miller3_dbl_step_sel(tc, l0, l1, l2, sel, t);
}
// sel: px on role 0, py on roles 1 and 2 (role 1's product is not used).
// Who computes what is chosen so that every product finds its operands on its own lane or in a value that is shared anyway:
// role b never needs Z, and the new Y and Z are produced by the lanes that keep them — six exchanges of an Fp2 value per step
// (Y, Y^2, 3b'Z^2, three line coefficients) instead of eight.
C12381_HD void miller3_dbl_step_sel(fp2& tc, fp2& l0, fp2& l1, fp2& l2, const fp& sel, const tri& t) {
    fp2 yt, s0, t0, t2b, z8, a, b, p2, u, y3p, p3, own, piece;
    tri_fetch_fp2(yt, tc, 1, t);                           // Y to everyone
    fp2_sqr(s0, tc);                                       // a: X^2, b: Y^2, c: Z^2
    tri_fetch_fp2(t0, s0, 1, t);                           // t0 = Y^2 to everyone
    {   // c: 3b' Z^2; shared afterwards
        fp2 tb;
        fp2_mul_b3(tb, s0);
        tri_fetch_fp2(t2b, tb, 2, t);
    }
    fp2_mul_small(z8, t0, 8);
    // round 2: a: X*Y, b: x3 = t2b * 8Y^2, c: t1 = Z*Y
    fp2_select(a, t.role == 1, t2b, tc);
    fp2_select(b, t.role == 1, z8, yt);
    fp2_mul(p2, a, b);
    // u = Y^2 - 9b' Z^2, y3' = Y^2 + 3b' Z^2
    fp2_dbl(u, t2b); fp2_add(u, u, t2b); fp2_sub(u, t0, u); fp2_norm1(u, u);
    fp2_add(y3p, t0, t2b);
    // round 3: a: u * xy, b: u * y3', c: t1 * z8
    fp2_select(a, t.role == 2, p2, u);
    fp2_select(b, t.role == 0, p2, y3p); fp2_select(b, t.role == 2, z8, b);
    fp2_mul(p3, a, b);
    // a: X3 = 2 u xy; b: Y3 = u y3' + x3; c: Z3 = t1 z8 — each on the lane that holds that coordinate
    fp2 xa, yc;
    fp2_dbl(xa, p3);
    fp2_add(yc, p3, p2);
    fp2_select(own, t.role == 0, xa, p3); fp2_select(own, t.role == 1, yc, own);
    // line pieces: a: l2 = 3 X^2 px, b: l1 = 3b'Z^2 - Y^2, c: l0 = -2 YZ (1+i) py
    fp2 cc, aa, pm, bb;
    fp2_dbl(cc, s0); fp2_add(cc, cc, s0);                  // a: 3X^2
    fp2_dbl(aa, p2); fp2_neg(aa, aa); fp2_mul_ip(aa, aa);  // c: -2YZ(1+i)
    fp2_select(a, t.role == 0, cc, aa);
    fp2_norm1(a, a);
    fp2_mul_fp(pm, a, sel);
    fp2_sub(bb, t2b, t0); fp2_norm1(bb, bb);
    fp2_select(piece, t.role == 1, bb, pm);
    tri_fetch_fp2(l2, piece, 0, t);
    tri_fetch_fp2(l0, piece, 2, t);
    tri_fetch_fp2(l1, piece, 1, t);
    tc = own;
}
C12381_HDN void miller3_dbl_step(fp2& tc, fp2& l0, fp2& l1, fp2& l2, const fp& px, const fp& py, const tri& t) {
    miller3_dbl_step_core(tc, l0, l1, l2, px, py, t);
}
// doubling step and the multiplication of f by its line in ONE out-of-line routine: the three line coefficients stay in
// registers instead of crossing two call boundaries through memory (42 stores + 42 loads per iteration).
// skip: the G1 argument of this pair is infinity — its line is replaced by 1 (PAIR_ate returns 1 for it, :448-449).
// F is this lane's LDS slot.
C12381_HDN void miller3_dbl_line(fp4& F, fp2& tc, const fp& px, const fp& py, bool skip, const tri& t) {
    fp2 l0, l1, l2, one2, zero2;
    miller3_dbl_step_core(tc, l0, l1, l2, px, py, t);
    fp2_one(one2); fp2_zero(zero2);
    fp2_select(l0, skip, one2, l0); fp2_select(l1, skip, zero2, l1); fp2_select(l2, skip, zero2, l2);
    fp4 x;
    slot_load(x, F);
    f12t_mul_line_core(x, l0, l1, l2, t);
    slot_store(F, x);
}
// ONE iteration of the loop of one pair without its addition step: f <- f^2, T <- 2T, f <- f * line — in one out-of-line routine.
// The running coordinate travels in registers (argument and return value), f and the G1 coordinate in the lane's slot: the
// iteration touches no private memory (as three routines it re-read tc, px, py and wrote tc back every iteration: 24 KB per lane
// and Miller loop, the largest part of the pairing kernels' traffic past L2, profiles/r02_traffic_split.txt).
// info = role | base << 2 | skip << 8.  The squaring forms (x + x')^2 first and reads x again from the slot for x^2, so that only
// one of them is alive with tc during an Fp4 squaring.
C12381_HD void miller3_iter_body(fp2& tc, fp4& F, int info) {
    tri t;
    t.role = info & 3; t.base = (info >> 2) & 63;
    const bool skip = ((info >> 8) & 1) != 0;
    {
        fp4 z, zc, zn, e, w;
        {
            fp4 x, xn, sx;
            slot_load(x, F);
            slot_load_role(xn, F, tri_next(t), t);
            fp4_add(sx, x, xn);                            // limbs < 2^29 + slack: within the operand bound of the Fp2 products
            C12381_PHASE();
            fp4_sqr_core_raw(zc, sx);                           // un-normalised: zc only enters e, which gets its own carry round
            C12381_PHASE();
        }
        {
            fp4 x;
            slot_unpark(x, F);
            fp4_sqr_core(z, x);
        }
        tri_fetch_fp4(zn, z, tri_next(t), t);
        fp4_sub(e, zc, z); fp4_sub(e, e, zn); fp4_norm1(e, e);
        f12t_combine(w, z, zn, e, t);
        slot_park(F, w);
    }
    C12381_PHASE();
    fp2 l0, l1, l2;
    {
        fp2 one2, zero2;
        fp sel;
        slot_psel_load(sel, F);
        miller3_dbl_step_sel(tc, l0, l1, l2, sel, t);
        fp2_one(one2); fp2_zero(zero2);
        fp2_select(l0, skip, one2, l0); fp2_select(l1, skip, zero2, l1); fp2_select(l2, skip, zero2, l2);
    }
    C12381_PHASE();
    {
        fp4 x;
        slot_unpark(x, F);
        f12t_mul_line_core(x, l0, l1, l2, t);
        slot_store(F, x);
    }
}
#if defined(__HIP_DEVICE_COMPILE__)
// the coordinate arrives as 28 scalar arguments (v0..v27) and leaves as one vector built in a single step: a 32-dword vector
// ARGUMENT stays one 1024-bit register tuple for as long as any element is alive, and the allocator spills such tuples whole
#define C12381_M3_28(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(17) X(18) X(19) X(20) X(21) \
    X(22) X(23) X(24) X(25) X(26) X(27)
#define C12381_M3_PARAM(i) int32_t a##i,
#define C12381_M3_TAKE(i) (i < NL ? tc.a.l[i % NL] : tc.b.l[i % NL]) = a##i;
#define C12381_M3_PASS(i) R[i],
C12381_HDN miller3_regs miller3_iter_regs(C12381_M3_28(C12381_M3_PARAM) int32_t info, int32_t slot) {
    fp2 tc;
    C12381_M3_28(C12381_M3_TAKE)
    fp4& F = *(fp4*)(c12381_lds_fp4*)(uintptr_t)(uint32_t)slot;
    miller3_iter_body(tc, F, info);
    return m3r_make(tc, info, slot);
}
C12381_HD miller3_regs miller3_iter(const miller3_regs& R) { return miller3_iter_regs(C12381_M3_28(C12381_M3_PASS) R[28], R[29]); }
#else
C12381_HDN miller3_regs miller3_iter(const miller3_regs& R) {
    miller3_regs Q = R;
    miller3_iter_body(Q.tc, *Q.F, Q.info);
    return Q;
}
#endif

// ------------------------------------------------------------------ Miller loop on a triple, in pieces
// (the kernels run it either whole or as two half-ranges of a work queue, see k_pair3.hip)
C12381_HD void miller3_q(g2p& Q, const fp2& qx, const fp2& qy, bool q_inf) {      // G2 infinity runs as (0:1:0), like PAIR_ate
    g2p inf;
    g2_set_inf(inf);
    Q.x = qx; Q.y = qy; fp2_one(Q.z);
    fp2_select(Q.x, q_inf, inf.x, Q.x); fp2_select(Q.y, q_inf, inf.y, Q.y); fp2_select(Q.z, q_inf, inf.z, Q.z);
}
C12381_HD void miller3_tc(fp2& tc, const g2p& Q, const tri& t) {                     // role 0/1/2 holds X/Y/Z of T
    fp2_select(tc, t.role == 0, Q.x, Q.y); fp2_select(tc, t.role == 2, Q.z, tc);
}
C12381_HD void f12t_one(fp4& F, const tri& t) {
    fp4 one4, zero4;
    fp4_zero(zero4); one4 = zero4; fp_one(one4.a.a);
    fp4_select(F, t.role == 0, one4, zero4);
}
// addition step T <- T +- Q and its line (the 5 iterations whose digit of 3|x| - |x| is non-zero), on the three lanes.
// The values are those of PAIR_add (pair_BLS12381.cpp:81-116) + ECP2_add (ecp2_BLS12381.cpp:413-502) with the second operand
// affine, Z2 in {1, 0} (0: the G2 argument is infinity and runs as (0 : 1 : 0) with the affine view (0, 1), like there):
//   line   l0 = (X1 - Z1 x2)(1+i) py,  l1 = (Y1 - Z1 y2) x2 - (X1 - Z1 x2) y2,  l2 = -(Y1 - Z1 y2) px
//   sum    t0 = X1 x2, t1 = Y1 y2, t2 = Z1 Z2, t3 = X1 y2 + Y1 x2, t4 = Y1 Z2 + Z1 y2, y3 = 3b'(X1 Z2 + Z1 x2),
//          X3 = t3 (t1 - 3b't2) - y3 t4,  Y3 = y3 3t0 + (t1 - 3b't2)(t1 + 3b't2),  Z3 = (t1 + 3b't2) t4 + 3t0 t3
// (the products by Z2 and the Karatsuba forms of t3, t4, y3 in the reference are these values).  Five rounds of one Fp2 product per
// lane — own coordinate times x2 / y2 twice, the two line products, two products of the own output coordinate — instead of the
// whole one-lane step (16 Fp2 products) replicated on every lane.
C12381_HD void miller3_add_line(fp4& F, fp2& tc, const fp& px, const fp& py, bool skip, const g2p& Q, bool neg, const tri& t) {
    fp2 X1, Y1, Z1, x2, y2, zero2, one2;
    fp2_zero(zero2); fp2_one(one2);
    tri_fetch_fp2(X1, tc, 0, t); tri_fetch_fp2(Y1, tc, 1, t); tri_fetch_fp2(Z1, tc, 2, t);
    x2 = Q.x;
    {
        fp2 ny;
        fp2_neg(ny, Q.y); fp2_norm1(ny, ny);
        fp2_select(y2, neg, ny, Q.y);
    }
    const bool z2_zero = fp2_is_zero(Q.z);
    // rounds 1, 2: own coordinate times (x2 | y2 | y2) and (y2 | x2 | x2)
    fp2 B, p1, p2;
    fp2_select(B, t.role == 0, x2, y2);
    fp2_mul(p1, tc, B);                                     // a: t0 = X1 x2, b: t1 = Y1 y2, c: zy = Z1 y2
    fp2_select(B, t.role == 0, y2, x2);
    fp2_mul(p2, tc, B);                                     // a: X1 y2,      b: Y1 x2,      c: zx = Z1 x2
    fp2 t0, t1, zy, xy, yx, zx;
    tri_fetch_fp2(t0, p1, 0, t); tri_fetch_fp2(t1, p1, 1, t); tri_fetch_fp2(zy, p1, 2, t);
    tri_fetch_fp2(xy, p2, 0, t); tri_fetch_fp2(yx, p2, 1, t); tri_fetch_fp2(zx, p2, 2, t);
    C12381_PHASE();
    // values every lane forms for itself (additions only)
    fp2 aa, cc, t3, t4, y3, t2, z3, t1m, t03, xz, yz, zz;
    fp2_sub(aa, X1, zx); fp2_norm1(aa, aa);
    fp2_sub(cc, Y1, zy); fp2_norm1(cc, cc);
    fp2_add(t3, xy, yx); fp2_norm1(t3, t3);
    fp2_select(xz, z2_zero, zero2, X1); fp2_select(yz, z2_zero, zero2, Y1); fp2_select(zz, z2_zero, zero2, Z1);
    fp2_add(t4, yz, zy); fp2_norm1(t4, t4);
    fp2_add(y3, xz, zx); fp2_mul_b3(y3, y3);
    fp2_norm1(t2, zz); fp2_mul_b3(t2, t2);
    fp2_add(z3, t1, t2); fp2_norm1(z3, z3);
    fp2_sub(t1m, t1, t2); fp2_norm1(t1m, t1m);
    fp2_mul_small(t03, t0, 3);
    // operands of rounds 4, 5 (the lane's own coordinate of T + Q) are picked now, so that the six shared values die here
    fp2 A4, B4, A5, B5;
    fp2_select(A4, t.role == 0, t3, y3);  fp2_select(A4, t.role == 2, z3, A4);
    fp2_select(B4, t.role == 0, t1m, t03); fp2_select(B4, t.role == 2, t4, B4);
    fp2_select(A5, t.role == 0, y3, t1m);  fp2_select(A5, t.role == 2, t03, A5);
    fp2_select(B5, t.role == 0, t4, z3);   fp2_select(B5, t.role == 2, t3, B5);
    C12381_PHASE();
    // round 3: a: aa y2, b: cc x2;  half products: a: aa (1+i) py, b: -cc px
    fp2 A, p3, aai, ncc, H, h;
    fp s;
    fp2_select(A, t.role == 1, cc, aa);
    fp2_select(B, t.role == 1, x2, y2);
    fp2_mul(p3, A, B);
    fp2_mul_ip(aai, aa); fp2_neg(ncc, cc);
    fp2_select(H, t.role == 1, ncc, aai);
    fp_select(s, t.role == 1, px, py);
    fp2_mul_fp(h, H, s);
    C12381_PHASE();
    // rounds 4, 5
    fp2 P4, P5, dif, sum;
    fp2_mul(P4, A4, B4);
    fp2_mul(P5, A5, B5);
    fp2_sub(dif, P4, P5); fp2_add(sum, P4, P5);
    fp2_select(tc, t.role == 0, dif, sum);
    C12381_PHASE();
    // the line
    fp2 l0, l1, l2, m, tl;
    tri_fetch_fp2(tl, p3, 0, t); tri_fetch_fp2(m, p3, 1, t);
    tri_fetch_fp2(l0, h, 0, t); tri_fetch_fp2(l2, h, 1, t);
    fp2_sub(l1, m, tl); fp2_norm1(l1, l1);
    fp2_select(l0, skip, one2, l0); fp2_select(l1, skip, zero2, l1); fp2_select(l2, skip, zero2, l2);
    f12t_mul_line_h(F, l0, l1, l2, t);
}
// everything iteration i does for ONE (P, Q) pair after the squaring: doubling step + line, and the addition step of
// the 5 iterations whose digit of 3|x| - |x| is non-zero.  skip: P is infinity, the pair contributes 1 (PAIR_ate :448-449).
C12381_HD void miller3_pair_step(fp4& F, fp2& tc, const fp& px, const fp& py, bool skip, const g2p& Q, int i, const tri& t) {
    constexpr unsigned __int128 N1 = (unsigned __int128)BLS_X;
    constexpr unsigned __int128 N3 = N1 * 3;
    miller3_dbl_line(F, tc, px, py, skip, t);
    const int bt = (int)((N3 >> i) & 1) - (int)((N1 >> i) & 1);
    if (bt != 0) miller3_add_line(F, tc, px, py, skip, Q, bt < 0, t);      // wave-uniform
}
// iterations hi .. lo (inclusive, 64 >= hi >= lo >= 1) of the loop for one pair / for two pairs sharing the squarings
// F: the lane's pair_slot (its psel field is set here)
C12381_HDN void miller3_range(fp4& F, fp2& tc, const fp& px, const fp& py, bool skip, const g2p& Q, int hi, int lo, const tri& t) {
    constexpr unsigned __int128 N1 = (unsigned __int128)BLS_X;
    constexpr unsigned __int128 N3 = N1 * 3;
    {
        fp sel;
        fp_select(sel, t.role == 0, px, py);
        slot_psel_store(F, sel);
    }
    const int info = t.role | (t.base << 2) | (skip ? 256 : 0);
    miller3_regs R = m3r_pack(tc, F, info);
#pragma unroll 1
    for (int i = hi; i >= lo; --i) {
        C12381_FAIR_SHARE(i, F);
        R = miller3_iter(R);
        const int bt = (int)((N3 >> i) & 1) - (int)((N1 >> i) & 1);
        if (bt != 0) {                                         // wave-uniform: the 5 addition steps
            m3r_tc(tc, R);
            miller3_add_line(F, tc, px, py, skip, Q, bt < 0, t);
            R = m3r_with_tc(R, tc);
        }
    }
    m3r_tc(tc, R);
}
C12381_HDN void miller3_range2(fp4& F, fp2& tc1, const fp& px1, const fp& py1, bool skip1, const g2p& Q1,
                               fp2& tc2, const fp& px2, const fp& py2, bool skip2, const g2p& Q2, int hi, int lo, const tri& t) {
#pragma unroll 1
    for (int i = hi; i >= lo; --i) {
        C12381_FAIR_SHARE(i, F);
        f12t_sqr_h(F, t);
        miller3_pair_step(F, tc1, px1, py1, skip1, Q1, i, t);
        miller3_pair_step(F, tc2, px2, py2, skip2, Q2, i, t);
    }
}
C12381_HD void miller3_q(g2p& Q, const fp2& qx, const fp2& qy, bool q_inf);

// iterations hi .. lo of the joint loop of K <= MAX_PROD pairs sharing the squarings (PAIR_double_ate pair_BLS12381.cpp:508-626
// generalised: f <- f^2 * l_1 * ... * l_K per iteration, so the value is the product of the K single-loop values as a field
// element).  The per-pair operands live in arrays: K is a run-time argument of the product kernels.
constexpr int MAX_PROD = 3;
struct miller3_pair { fp px, py; fp2 tc; g2p Q; bool skip; };
C12381_HDN void miller3_rangeK(fp4& F, miller3_pair* pr, int K, int hi, int lo, const tri& t) {
#pragma unroll 1
    for (int i = hi; i >= lo; --i) {
        C12381_FAIR_SHARE(i, F);
        f12t_sqr_h(F, t);
#pragma unroll 1
        for (int j = 0; j < K; ++j) miller3_pair_step(F, pr[j].tc, pr[j].px, pr[j].py, pr[j].skip, pr[j].Q, i, t);
    }
}


// ------------------------------------------------------------------ fixed G2 argument: precomputed lines
// The running point T and the coefficients of every line depend on Q only:  l0 = c0 * py,  l1 = c1,  l2 = c2 * px
// (miller_dbl_step / miller_add_step, pairing.hpp).  When a whole batch pairs against ONE Q (the public w and g2 of
// BBS+ verification) the 69 coefficient triples are computed once — by running the one-lane steps with P = (1, 1) —
// and an iteration costs four Fp multiplications per line instead of the G2 doubling.
constexpr int FQ_LINES = 69;                         // 64 doubling steps + 5 addition steps, in loop order
constexpr int FQ_LINE_DWORDS = 6 * NL;               // record stride: c0, c1, c2 (336 B = 21 16-byte words); normalised records use c0/c1, c2/c1 (224 B)
constexpr int FQ_FMT_WORD = FQ_LINES * FQ_LINE_DWORDS;  // tab[FQ_FMT_WORD]: 1 = every record is (c0/c1, c2/c1), 0 = raw (c0, c1, c2)
constexpr int FQ_TABLE_DWORDS = FQ_FMT_WORD + 4;
constexpr int fq_line_index(int i) {                 // index of iteration i's doubling line (its addition line follows)
    constexpr unsigned __int128 N1 = (unsigned __int128)BLS_X;
    constexpr unsigned __int128 N3 = N1 * 3;
    int k = 0;
    for (int j = 64; j > i; --j) k += 1 + ((((N3 >> j) & 1) != ((N1 >> j) & 1)) ? 1 : 0);
    return k;
}
C12381_HD void fq_store_line(int32_t* dst, const fp2& c0, const fp2& c1, const fp2& c2) {
    msm_store_pt(dst, c0.a, c0.b);
    msm_store_pt(dst + 2 * NL, c1.a, c1.b);
    msm_store_pt(dst + 4 * NL, c2.a, c2.b);
}
C12381_HD void fq_load_line(fp2& c0, fp2& c1, fp2& c2, const int32_t* src) {
    msm_load_pt(c0.a, c0.b, src);
    msm_load_pt(c1.a, c1.b, src + 2 * NL);
    msm_load_pt(c2.a, c2.b, src + 4 * NL);
}
// one lane: the whole coefficient table of Q (affine point of the twist; "infinity" runs as (0:1:0) with the affine
// view (0, 1), exactly as the running-point loop does — the coefficients are the same field elements either way)
C12381_HDN void miller_lines_precompute(int32_t* tab, const fp2& qx, const fp2& qy, bool q_inf = false, bool normalise = true) {
    g2p Q, T;
    miller3_q(Q, qx, qy, q_inf);
    T = Q;
    fp one;
    fp_one(one);
    constexpr unsigned __int128 N1 = (unsigned __int128)BLS_X;
    constexpr unsigned __int128 N3 = N1 * 3;
    int k = 0;
#pragma unroll 1
    for (int i = 64; i >= 1; --i) {
        fp2 l0, l1, l2, n0, n1, n2;
        miller_dbl_step(T, l0, l1, l2, one, one);
        fp2_norm1(n0, l0); fp2_norm1(n1, l1); fp2_norm1(n2, l2);
        fq_store_line(tab + (size_t)(k++) * FQ_LINE_DWORDS, n0, n1, n2);
        const int bt = (int)((N3 >> i) & 1) - (int)((N1 >> i) & 1);
        if (bt != 0) {
            g2p S = Q;
            if (bt < 0) g2_neg(S, Q);
            miller_add_step(T, S, l0, l1, l2, one, one);
            fp2_norm1(n0, l0); fp2_norm1(n1, l1); fp2_norm1(n2, l2);
            fq_store_line(tab + (size_t)(k++) * FQ_LINE_DWORDS, n0, n1, n2);
        }
    }
    // Normalised form: every line divided by its s-coefficient c1.  A factor in Fp2 of a line changes the Miller value by a
    // factor in Fp2, and (p^12 - 1)/r is a multiple of p^2 - 1 (r divides p^4 - p^2 + 1), so the final exponentiation removes
    // it: the GT value — the only thing the fixed-argument kernels output — is unchanged, and the line product needs four Fp2
    // products per lane instead of five (f12t_mul_line1_core).  One simultaneous inversion over the 69 coefficients; a table
    // with a vanishing c1 (G2 infinity: the addition lines are (0, 0, -px)) stays raw.
    if (!normalise) { tab[FQ_FMT_WORD] = 0; return; }      // A/B switch (C12381_FQ_RAW)
    fp2 pref[FQ_LINES];
    fp2 run;
    fp2_one(run);
#pragma unroll 1
    for (int j = 0; j < FQ_LINES; ++j) {
        fp2 c0, c1, c2, nr;
        fq_load_line(c0, c1, c2, tab + (size_t)j * FQ_LINE_DWORDS);
        fp2_mul(nr, run, c1);
        fp2_norm1(run, nr);
        pref[j] = run;
    }
    if (fp2_is_zero(run)) { tab[FQ_FMT_WORD] = 0; return; }
    fp2 inv;
    fp2_inv(inv, run);
#pragma unroll 1
    for (int j = FQ_LINES - 1; j >= 0; --j) {
        fp2 c0, c1, c2, ij, ni, a0, a2, n0, n2;
        fq_load_line(c0, c1, c2, tab + (size_t)j * FQ_LINE_DWORDS);
        if (j > 0) { fp2_mul(ij, inv, pref[j - 1]); } else { ij = inv; }
        fp2_norm1(ij, ij);
        fp2_mul(ni, inv, c1); fp2_norm1(inv, ni);
        fp2_mul(a0, c0, ij); fp2_mul(a2, c2, ij);
        fp2_norm1(n0, a0); fp2_norm1(n2, a2);
        msm_store_pt(tab + (size_t)j * FQ_LINE_DWORDS, n0.a, n0.b);
        msm_store_pt(tab + (size_t)j * FQ_LINE_DWORDS + 2 * NL, n2.a, n2.b);
    }
    tab[FQ_FMT_WORD] = 1;
}
// f *= line(tab[k]) evaluated at P = (px, py): role 0 forms l0 = c0 py, role 1 forms l2 = c2 px (role 2 repeats role 1), and
// the triple shares the two products — one Fp2-by-Fp product per lane instead of two.  Raw records (c0, c1, c2):
C12381_HDN void miller3_fixed_line_raw(fp4& F, const int32_t* tab, int k, const fp& px, const fp& py, bool skip, const tri& t) {
    fp2 c0, c1, c2, l0, l2, one2, zero2, cs, prod;
    fp ps;
    fq_load_line(c0, c1, c2, tab + (size_t)k * FQ_LINE_DWORDS);
    fp2_select(cs, t.role == 0, c0, c2);
    fp_select(ps, t.role == 0, py, px);
    fp2_mul_fp(prod, cs, ps);
    tri_fetch_fp2(l0, prod, 0, t);
    tri_fetch_fp2(l2, prod, 1, t);
    fp2_one(one2); fp2_zero(zero2);
    fp2_select(l0, skip, one2, l0); fp2_select(c1, skip, zero2, c1); fp2_select(l2, skip, zero2, l2);
    fp4 x;
    slot_load(x, F);
    f12t_mul_line_core(x, l0, c1, l2, t);
    slot_store(F, x);
}
// normalised records (c0/c1, c2/c1): the line is l0 + s + l2 (...).  A skipped pair (G1 argument at infinity) multiplies by s
// alone (l0 = l2 = 0) instead of by 1: a factor in Fp4 is removed by the final exponentiation just like one in Fp2 ((p^12 - 1)/r
// is a multiple of p^4 - 1), and f keeps going through the same reduction as every other lane's.
C12381_HDN void miller3_fixed_line1(fp4& F, const int32_t* tab, int k, const fp& px, const fp& py, bool skip, const tri& t) {
    fp2 l0, l2, cs, prod, zero2;
    fp ps;
    const int32_t* rec = tab + (size_t)k * FQ_LINE_DWORDS;
    msm_load_pt(cs.a, cs.b, rec + (t.role == 0 ? 0 : 2 * NL));      // role 0 needs c0 / c1, the others c2 / c1: one record, chosen by address
    fp_select(ps, t.role == 0, py, px);
    fp2_mul_fp(prod, cs, ps);
    tri_fetch_fp2(l0, prod, 0, t);
    tri_fetch_fp2(l2, prod, 1, t);
    fp2_zero(zero2);
    fp2_select(l0, skip, zero2, l0); fp2_select(l2, skip, zero2, l2);
    fp4 x;
    slot_load(x, F);
    f12t_mul_line1_core(x, l0, l2, t);
    slot_store(F, x);
}
C12381_HD void miller3_fixed_line(fp4& F, const int32_t* tab, bool norm, int k, const fp& px, const fp& py, bool skip, const tri& t) {
    if (norm) miller3_fixed_line1(F, tab, k, px, py, skip, t);       // wave-uniform: a property of the table
    else miller3_fixed_line_raw(F, tab, k, px, py, skip, t);
}
// iterations hi .. lo of the loop of one pair whose G2 argument is fixed (table tab)
C12381_HDN void miller3_range_fixed(fp4& F, const fp& px_, const fp& py_, bool skip, const int32_t* tab, int hi, int lo, const tri& t_) {
    constexpr unsigned __int128 N1 = (unsigned __int128)BLS_X;
    constexpr unsigned __int128 N3 = N1 * 3;
    int k = 0;
    tab = wave_uniform(tab);
    hi = wave_uniform(hi); lo = wave_uniform(lo);
    const bool norm = wave_uniform((int)tab[FQ_FMT_WORD]) != 0;
    const fp& px = *wave_uniform(&px_); const fp& py = *wave_uniform(&py_);
    const tri& t = *wave_uniform(&t_);
#pragma unroll 1
    for (int j = 64; j > hi; --j) k += 1 + ((((N3 >> j) & 1) != ((N1 >> j) & 1)) ? 1 : 0);
#pragma unroll 1
    for (int i = hi; i >= lo; --i) {
        C12381_FAIR_SHARE(i, F);
        f12t_sqr_h(F, t);
        miller3_fixed_line(F, tab, norm, k++, px, py, skip, t);
        if (((N3 >> i) & 1) != ((N1 >> i) & 1)) miller3_fixed_line(F, tab, norm, k++, px, py, skip, t);
    }
}
// iterations hi .. lo of the joint loop of two pairs whose G2 arguments are both fixed (tables tab1, tab2)
C12381_HDN void miller3_range2_fixed(fp4& F, const fp& px1_, const fp& py1_, bool skip1, const int32_t* tab1,
                                     const fp& px2_, const fp& py2_, bool skip2, const int32_t* tab2, int hi, int lo, const tri& t_) {
    constexpr unsigned __int128 N1 = (unsigned __int128)BLS_X;
    constexpr unsigned __int128 N3 = N1 * 3;
    int k = 0;
    tab1 = wave_uniform(tab1); tab2 = wave_uniform(tab2);
    hi = wave_uniform(hi); lo = wave_uniform(lo);
    const bool norm1 = wave_uniform((int)tab1[FQ_FMT_WORD]) != 0, norm2 = wave_uniform((int)tab2[FQ_FMT_WORD]) != 0;
    const fp& px1 = *wave_uniform(&px1_); const fp& py1 = *wave_uniform(&py1_);
    const fp& px2 = *wave_uniform(&px2_); const fp& py2 = *wave_uniform(&py2_);
    const tri& t = *wave_uniform(&t_);
#pragma unroll 1
    for (int j = 64; j > hi; --j) k += 1 + ((((N3 >> j) & 1) != ((N1 >> j) & 1)) ? 1 : 0);
#pragma unroll 1
    for (int i = hi; i >= lo; --i) {
        C12381_FAIR_SHARE(i, F);
        f12t_sqr_h(F, t);
        miller3_fixed_line(F, tab1, norm1, k, px1, py1, skip1, t);
        miller3_fixed_line(F, tab2, norm2, k, px2, py2, skip2, t);
        ++k;
        if (((N3 >> i) & 1) != ((N1 >> i) & 1)) {           // wave-uniform: addition step of this iteration
            miller3_fixed_line(F, tab1, norm1, k, px1, py1, skip1, t);
            miller3_fixed_line(F, tab2, norm2, k, px2, py2, skip2, t);
            ++k;
        }
    }
}

// f = conj(Miller_{|x|}(Q, P)) on a triple.  Returns this lane's coefficient.
C12381_HDN void miller3_loop(fp4& F, const fp& px, const fp& py, bool p_inf, const fp2& qx, const fp2& qy, bool q_inf, const tri& t) {
    g2p Q;
    fp2 tc;
    miller3_q(Q, qx, qy, q_inf);
    miller3_tc(tc, Q, t);
    { fp4 one; f12t_one(one, t); slot_store(F, one); }
    miller3_range(F, tc, px, py, p_inf, Q, 64, 1, t);
    f12t_conj_h(F, t);
}
// Two Miller loops with SHARED squarings: F = conj(M(Q1, P1) * M(Q2, P2)) — the product the reference forms from two
// pair_ate results (liner_pair.hpp:339-350) costs one Fp12 squaring per iteration instead of two.  As a field element
// the result is exactly the product of the two single-loop values (f <- f^2 * l distributes over the product).
C12381_HDN void miller3_loop2(fp4& F, const fp& px1, const fp& py1, bool p_inf1, const fp2& qx1, const fp2& qy1, bool q_inf1,
                              const fp& px2, const fp& py2, bool p_inf2, const fp2& qx2, const fp2& qy2, bool q_inf2, const tri& t) {
    g2p Q1, Q2;
    fp2 tc1, tc2;
    miller3_q(Q1, qx1, qy1, q_inf1); miller3_q(Q2, qx2, qy2, q_inf2);
    miller3_tc(tc1, Q1, t); miller3_tc(tc2, Q2, t);
    { fp4 one; f12t_one(one, t); slot_store(F, one); }
    miller3_range2(F, tc1, px1, py1, p_inf1, Q1, tc2, px2, py2, p_inf2, Q2, 64, 1, t);
    f12t_conj_h(F, t);
}

}  // namespace c12381
