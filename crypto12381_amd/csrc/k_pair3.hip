// Three-lanes-per-pairing kernels (pairing3.hpp): each triple of lanes holds one Fp12 (one Fp4 coefficient per lane)
// and one G2 point (one projective coordinate per lane); 21 pairings per wavefront.
//   pair3_kernel       Miller loop + final exponentiation -> 576-byte GT
//   pair3_eq_kernel    e(a1,a2) == e(b1,b2)
#include "kernels_common.hpp"
#include "pairing3.hpp"
#include "fixed_base.hpp"

// Groups left to the tenth-length queue tasks at the end of a launch.  Measured on MI355X (profiles/r02_ab_queued_groups.txt):
// 2^16 pairings (3121 groups, 2048 resident wavefronts): 2048 queued 21.8 ms, 1024 queued 19.4 ms, 512 queued 20.1 ms;
// 2^18 BBS+ verifications (12484 groups): 2048 queued 82.0 ms, 1024 queued 82.4 ms, 512 queued 88.9 ms, 256 queued 97.3 ms —
// the more whole groups a wavefront runs, the further apart the wavefronts finish: a third of the groups, between half a
// grid and two grids (re-measured with the normalised line tables: 12484 groups, 2048 queued 83.2 ms, 4096 80.5, 6144 80.4, all 82.1).
// tuning runs: C12381_QUEUE_GROUPS = number of groups that go through the queue (c12381_set_queue_groups; 0 = the rule below)
__device__ int g_queue_groups_override = 0;
__device__ __forceinline__ size_t queue_direct_groups(size_t ngroups, size_t nwaves) {
    if (ngroups <= nwaves) return 0;
    size_t queued = ngroups / 3;
    if (queued < nwaves / 2) queued = nwaves / 2;
    if (queued > 2 * nwaves) queued = 2 * nwaves;             // 12484 groups (2^18 BBS+): 2048 queued 83.2 ms, 4096 80.5, 6144 80.4, all 82.1 (r02_ab_queued_groups.txt)
    const int ov = g_queue_groups_override;
    if (ov > 0) queued = (size_t)ov < ngroups ? (size_t)ov : ngroups;
    return ngroups - queued;
}
// Miller-loop tasks per group in the queue kernels (the 64 iterations in equal parts)
constexpr int MILLER_TASKS_PER_GROUP = 4;       // 2 and 8 measured: 2 loses 4 %, 8 is within the spread of 4 (profiles/r04_ab_queue_priority.txt, r05_ab_miller_variants.txt)
static_assert(64 % MILLER_TASKS_PER_GROUP == 0, "the Miller-loop tasks divide the 64 iterations");
constexpr int MILLER_ITERS_PER_TASK = 64 / MILLER_TASKS_PER_GROUP;
// (Issue priorities for the two kinds of work — s_setprio around the whole groups / the tasks — were measured and are not in the tree: whole groups
// above tasks doubles the launch, a task holder that cannot issue stalls everyone behind its hand-over; the other two orders change nothing, the
// SIMD's age order already does that.  profiles/r04_ab_queue_priority.txt)
namespace c12381 {
int set_queue_groups_override(int v) { return hipMemcpyToSymbol(HIP_SYMBOL(g_queue_groups_override), &v, sizeof(int)) == hipSuccess ? 0 : -1; }
}

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

// One pair_slot per lane in LDS (pairing3.hpp: the Fp4 + one Fp, 304-byte stride): the running Miller value and the accumulator
// of the exponentiations by x live here, so the out-of-line tower routines read and write their hot operand at LDS latency.
// 76 KB per workgroup, two workgroups per CU.
typedef pair_slot fp4_slot;

__device__ __forceinline__ void gt_load_coeff(fp4& x, const uint8_t* p576, int role) {
    const uint8_t* p = p576 + (role == 0 ? 384 : (role == 1 ? 192 : 0));
    uint32_t raw[12];
    load_raw48(raw, p); fp_from_raw48(x.b.b, raw);
    load_raw48(raw, p + 48); fp_from_raw48(x.b.a, raw);
    load_raw48(raw, p + 96); fp_from_raw48(x.a.b, raw);
    load_raw48(raw, p + 144); fp_from_raw48(x.a.a, raw);
}

// ---- work-queue variant: state of a group of 21 pairings between two phases, as rows of 64 x 16 bytes (one per lane)
constexpr int ST_ROWS_F = 14, ST_ROWS_TC = 7;                 // Fp4 = 56 dwords, Fp2 = 28 dwords
constexpr int GT_POW_TAB_ROWS = 16 * ST_ROWS_F;               // gt3_op_kernel's table of x^0 .. x^15 per wavefront
static_assert(GT_POW_TAB_BYTES_PER_WAVE == (size_t)GT_POW_TAB_ROWS * 64 * 16, "kernels.hpp: size of the power table per wavefront");
constexpr int ST_F = 0, ST_TC1 = ST_ROWS_F, ST_TC2 = ST_TC1 + ST_ROWS_TC, ST_Y1 = ST_TC2 + ST_ROWS_TC;

template <class T, int ROWS>
__device__ __forceinline__ void st_store(uint4* rows, unsigned lane, const T& x) {
    static_assert(sizeof(T) == ROWS * 16, "state row count");
    const int32_t* w = reinterpret_cast<const int32_t*>(&x);
#pragma unroll
    for (int r = 0; r < ROWS; ++r) rows[(size_t)r * 64 + lane] = make_uint4((uint32_t)w[4 * r], (uint32_t)w[4 * r + 1], (uint32_t)w[4 * r + 2], (uint32_t)w[4 * r + 3]);
}
template <class T, int ROWS>
__device__ __forceinline__ void st_load(T& x, const uint4* rows, unsigned lane) {
    static_assert(sizeof(T) == ROWS * 16, "state row count");
    int32_t* w = reinterpret_cast<int32_t*>(&x);
#pragma unroll
    for (int r = 0; r < ROWS; ++r) { const uint4 v = rows[(size_t)r * 64 + lane]; w[4 * r] = (int32_t)v.x; w[4 * r + 1] = (int32_t)v.y; w[4 * r + 2] = (int32_t)v.z; w[4 * r + 3] = (int32_t)v.w; }
}

// ---- the same state WITHOUT cache maintenance (round 5): every 32-bit word travels in an 8-byte word together with a 32-bit tag — launch epoch and
// the phase that wrote it; a reader that sees another tag in ANY word reads again.  The accesses are agent-scope coherent (sc1: write-through /
// read past this XCD's L2 — the eight XCDs' L2s are not coherent with each other for ordinary accesses: with an L1 invalidation alone the results
// were wrong, profiles/r05_ab_fence_free_handover.txt), and a word validates itself, so no release / acquire pair is needed: none of the
// buffer_wbl2 sc1 / buffer_inv sc1 the fences of the 16-byte form cost per task — a write-back and an invalidation of the whole XCD's L2 under
// 256 resident wavefronts whose spills and private operands live there (tasks 5-7 % longer, the whole groups running beside them 2-3 %).
// Two tagged words per lane and instruction: buffer_load / buffer_store_dwordx4 with sc1 (as relaxed agent-scope ATOMIC 8-byte accesses —
// global_load / store_dwordx2 sc1 — the same traffic took ~600 cycles of issue per instruction, 65-105 K cycles per task to arrive and 20-48 K to
// leave: they are not coalesced).  An aligned 8-byte half of a lane's 16-byte access is never torn (one lane's 16 bytes move in one transaction),
// and tearing BETWEEN the halves is harmless, each carries its own tag.  Rows of 64 lanes x 16 bytes; one row per pair of 32-bit words.
constexpr int ST_DW_F = 56, ST_DW_TC = 28;                    // 32-bit words of an Fp4 / Fp2
constexpr int STW_F = 0, STW_TC1 = ST_DW_F / 2, STW_TC2 = STW_TC1 + ST_DW_TC / 2, STW_Y1 = STW_TC2 + ST_DW_TC / 2, STW_ROWS = STW_Y1 + ST_DW_F / 2;      // 84 rows of 1 KB
static_assert((size_t)STW_ROWS * 1024 == PAIR_QUEUE_STATE_BYTES, "kernels.hpp: size of a queued group's state block");
constexpr int STW_AUX = 16 | (int)0x80000000u;                // cache policy of the buffer instructions: sc1, volatile (never merged or hoisted)
typedef int32_t stw_v4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t st_tag(uint32_t epoch, unsigned int writer_phase) { return (epoch << 4) | (writer_phase + 1u); }
__device__ __forceinline__ __amdgpu_buffer_rsrc_t stw_rsrc(const void* rows) {          // rows: wave-uniform
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(rows), 0, 0x7fffffff, 0x00020000);
}
template <class T, int DW>
__device__ __forceinline__ void stw_store(void* rows, unsigned lane, const T& x, uint32_t tag) {
    static_assert(sizeof(T) == DW * 4 && DW % 2 == 0, "state word count");
    const int32_t* w = reinterpret_cast<const int32_t*>(&x);
    const __amdgpu_buffer_rsrc_t rs = stw_rsrc(rows);
#pragma unroll
    for (int r = 0; r < DW / 2; ++r) {
        stw_v4 v;
        v.x = w[2 * r]; v.y = (int32_t)tag; v.z = w[2 * r + 1]; v.w = (int32_t)tag;
        __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)((r * 64 + lane) * 16), 0, STW_AUX);
    }
}
// false: some word still carried another tag after spin_limit re-reads (the writer never finished): the caller poisons the group
template <class T, int DW>
__device__ __forceinline__ bool stw_load(T& x, const void* rows, unsigned lane, uint32_t tag, int spin_limit, unsigned int* reread = nullptr) {
    static_assert(sizeof(T) == DW * 4 && DW % 2 == 0, "state word count");
    int32_t* w = reinterpret_cast<int32_t*>(&x);
    const __amdgpu_buffer_rsrc_t rs = stw_rsrc(rows);
    int spins = 0;
    for (;;) {
        bool ok = true;
#pragma unroll
        for (int r = 0; r < DW / 2; ++r) {
            const stw_v4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)((r * 64 + lane) * 16), 0, STW_AUX);
            w[2 * r] = v.x; w[2 * r + 1] = v.z;
            ok = ok && (uint32_t)v.y == tag && (uint32_t)v.w == tag;
        }
        if (__builtin_amdgcn_ballot_w64(!ok) == 0) return true;           // wave-uniform
        if (reread) ++*reread;
        if (spin_limit < 0 || ++spins > spin_limit) return false;
        __builtin_amdgcn_s_sleep(64);
    }
}

// ---- hand-over protocol of the work queue.  flags[g] = number of finished phases of group g, bit 31 = the group is
// POISONED: a wavefront gave up waiting for its predecessor (bounded spin), so the group's state is not to be trusted.
// A poisoned task skips its arithmetic and passes the mark on (successors then start at once instead of spinning
// through their own bound), the last phase writes 0xff to every output of the group and raises bad_flag[1], which the
// host reports as C12381_E_INTERNAL — a library-internal failure never looks like valid output or like a bad input point.
// Publishing is an atomic max, so a predecessor that was merely slow cannot clear the mark afterwards.
constexpr unsigned int Q_POISON = 0x80000000u;
// spin_limit < 0 (tests only, C12381_PAIR_SPIN_LIMIT): every wait is treated as timed out
__device__ __forceinline__ bool queue_wait(unsigned int* flags, size_t g, unsigned int p, int spin_limit) {
    if (p == 0) return false;
    int spins = 0;
    for (;;) {
        const unsigned int v = (unsigned int)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&flags[g], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT));
        if (spin_limit < 0 || (v & Q_POISON)) return true;
        if (v >= p) return false;
        if (++spins > spin_limit) return true;
        __builtin_amdgcn_s_sleep(64);
    }
}
__device__ __forceinline__ void queue_publish(unsigned int* flags, size_t g, unsigned int p, bool poisoned, unsigned lane) {
    __threadfence();
    if (lane == 0) __hip_atomic_fetch_max(&flags[g], (poisoned ? Q_POISON : 0u) | (p + 1u), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
// The forms for state that travels in tagged words (stw_store / stw_load): the flag only says "worth looking" and carries the poison mark, the data
// validates itself — relaxed accesses, no cache maintenance, and the publisher does not wait for its stores either (21-48 K cycles per task while
// they drained): a reader that arrives before the last word re-reads.
__device__ __forceinline__ bool queue_wait_rlx(unsigned int* flags, size_t g, unsigned int p, int spin_limit) {
    if (p == 0) return false;
    int spins = 0;
    for (;;) {
        const unsigned int v = (unsigned int)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&flags[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        if (spin_limit < 0 || (v & Q_POISON)) return true;
        if (v >= p) return false;
        if (++spins > spin_limit) return true;
        __builtin_amdgcn_s_sleep(64);
    }
}
__device__ __forceinline__ void queue_publish_rlx(unsigned int* flags, size_t g, unsigned int p, bool poisoned, unsigned lane) {
    if (lane == 0) __hip_atomic_fetch_max(&flags[g], (poisoned ? Q_POISON : 0u) | (p + 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void gt_poison(uint8_t* o576, int role) {
    uint4* q = reinterpret_cast<uint4*>(o576 + (role == 0 ? 384 : (role == 1 ? 192 : 0)));
#pragma unroll
    for (int j = 0; j < 12; ++j) q[j] = make_uint4(~0u, ~0u, ~0u, ~0u);
}

// Diagnostic (experiments runs, C12381_PAIR_STAMPS): what ONE wavefront of a queue kernel spent its launch on, in shader-clock cycles
// (s_memtime) — whole groups, queue tasks from start to publish (state loads and stores included), hand-over waits (claim to start) —
// written once at exit as 12 words per wavefront (tools/queue_wave_stats.py).  Everything is wave-uniform (SGPRs); null pointer = off.
// Inside a task: loaded() after the state has arrived (the tag comparison has consumed it), computed() before the state is stored.
struct queue_wave_stats {
    unsigned long long* out;
    unsigned long long t_entry = 0, whole = 0, task = 0, wait = 0, t0 = 0, t1 = 0, load = 0, store = 0;
    unsigned int n_whole = 0, n_task = 0, reread = 0;
    __device__ __forceinline__ explicit queue_wave_stats(unsigned long long* wstats) : out(wstats) { if (out) t_entry = __builtin_amdgcn_s_memtime(); }
    __device__ __forceinline__ void mark() { if (out) t0 = __builtin_amdgcn_s_memtime(); }
    __device__ __forceinline__ void whole_done() { if (out) { whole += __builtin_amdgcn_s_memtime() - t0; ++n_whole; } }
    __device__ __forceinline__ void wait_done() { if (out) { const unsigned long long t = __builtin_amdgcn_s_memtime(); wait += t - t0; t0 = t; t1 = t; } }
    __device__ __forceinline__ void loaded() { if (out) { const unsigned long long t = __builtin_amdgcn_s_memtime(); load += t - t1; t1 = t; } }
    __device__ __forceinline__ void computed() { if (out) t1 = __builtin_amdgcn_s_memtime(); }
    __device__ __forceinline__ void task_done() { if (out) { const unsigned long long t = __builtin_amdgcn_s_memtime(); task += t - t0; store += t - t1; ++n_task; } }
    __device__ __forceinline__ void finish() {
        if (!out || (threadIdx.x & 63u) != 0) return;
        unsigned long long* o = out + 12 * (((size_t)blockIdx.x * BLOCK + threadIdx.x) >> 6);
        o[0] = t_entry; o[1] = __builtin_amdgcn_s_memtime(); o[2] = whole; o[3] = n_whole; o[4] = task; o[5] = n_task; o[6] = wait;
        o[7] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) | ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32);      // HW_ID | XCC_ID
        o[8] = load; o[9] = store; o[10] = reread;
    }
};

}  // namespace

namespace c12381 {

// One whole group of 21 pairings on this wavefront: Miller loop(s) + final exponentiation + output (lane element index i, shadow
// lanes inactive).  EQ: e(a1, a2) == e(b1, b2)  <=>  e(a1, a2) * e(-b1, b2) == 1: one joint Miller loop (shared squarings), one
// final exponentiation.  liner_pair.hpp:339-350 forms ate(a) * conj(ate(b)) from two separate loops; after the final
// exponentiation both are e(a) / e(b) (conj and negating the G1 argument both invert the pairing value; the Miller values differ
// by factors in Fp6, which the easy part kills), so the boolean is the same for all curve points, infinity arguments included.
template <bool EQ>
__device__ __forceinline__ void pair3_whole_group(size_t i, bool active, const uint8_t* a1, const uint8_t* a2, const uint8_t* b1, const uint8_t* b2,
                                                  size_t b2_stride, uint8_t* out, int* bad_flag, fp4& H, const tri& t) {
    fp px, py; fp2 qx, qy; bool pinf, qinf, ok, okb = true;
    pair_inputs(px, py, pinf, qx, qy, qinf, ok, a1 + 96 * i, a2 + 192 * i);
    if (!ok) { pinf = true; qinf = true; }
    if (EQ) {
        fp px2, py2; fp2 qx2, qy2; bool pinf2, qinf2;
        pair_inputs(px2, py2, pinf2, qx2, qy2, qinf2, okb, b1 + 96 * i, b2 + b2_stride * i);
        if (!okb) { pinf2 = true; qinf2 = true; }
        {
            fp ny;
            fp_neg(ny, py2);
            fp_norm1(py2, ny);
        }
        miller3_loop2(H, px, py, pinf, qx, qy, qinf, px2, py2, pinf2, qx2, qy2, qinf2, t);
    } else {
        miller3_loop(H, px, py, pinf, qx, qy, qinf, t);
    }
    fp4 F;
    slot_load(F, H);
    f12t_final_exp_ws(F, H, t);
    const bool valid = ok && okb;
    if (EQ) {
        const bool one = f12t_is_one(F, t);
        if (active && t.role == 0) {
            if (!valid) *bad_flag = 1;
            out[i] = valid ? (one ? 1 : 0) : 0xff;
        }
    } else if (active) {
        if (!valid) { *bad_flag = 1; gt_poison(out + 576 * i, t.role); }
        else gt_store_coeff(out + 576 * i, F, t.role);
    }
}

__global__ void __launch_bounds__(BLOCK, 2) pair3_kernel(size_t n, const uint8_t* g1, const uint8_t* g2, uint8_t* gt, int* bad_flag) {
    if ((((size_t)blockIdx.x * BLOCK + threadIdx.x) >> 6) * TRI_PER_WAVE >= n) return;      // whole wavefront idle
    tri t; size_t i; bool active;
    tri_setup(t, i, active, n);
    __shared__ fp4_slot slots[BLOCK];
    slot_fair_set(slots[threadIdx.x].v, 8);                      // plain grid: the two wavefronts of a SIMD take turns (pairing3.hpp C12381_FAIR_SHARE)
    pair3_whole_group<false>(i, active, g1, g2, nullptr, nullptr, 0, gt, bad_flag, slots[threadIdx.x].v, t);
}

__global__ void __launch_bounds__(BLOCK, 2) pair3_eq_kernel(size_t n, const uint8_t* a1, const uint8_t* a2, const uint8_t* b1, const uint8_t* b2,
                                                         size_t b2_stride, uint8_t* out, int* bad_flag, const int32_t* skip_if) {
    if (skip_if && skip_if[HDR_VALID] != 0) return;          // the fixed-G2 path serves this batch (pair3_prod_fixed_queue_kernel)
    if ((((size_t)blockIdx.x * BLOCK + threadIdx.x) >> 6) * TRI_PER_WAVE >= n) return;
    tri t; size_t i; bool active;
    tri_setup(t, i, active, n);
    __shared__ fp4_slot slots[BLOCK];
    slot_fair_set(slots[threadIdx.x].v, 8);                      // plain grid: the two wavefronts of a SIMD take turns (pairing3.hpp C12381_FAIR_SHARE)
    pair3_whole_group<true>(i, active, a1, a2, b1, b2, b2_stride, out, bad_flag, slots[threadIdx.x].v, t);
}

// gt[i] = prod_{j < k} e(g1[j * n + i], g2[j * n + i]),  k = 1 .. MAX_PROD: pair(a, b) * pair(c, d) [* pair(e, f)] as the headers
// form it (liner_pair.hpp:291-303 -> pair_double_ate pair_BLS12381.cpp:508-626, then PAIR_fexp) with ONE joint Miller loop
// (shared squarings) and ONE final exponentiation.  miller_only: stop before the final exponentiation — the value is then
// the reference's Miller value (the product of the single-loop values; a G1 argument at infinity contributes 1, :532-541).
__global__ void __launch_bounds__(BLOCK, 2) pair3_prod_kernel(size_t n, int k, const uint8_t* g1, const uint8_t* g2, uint8_t* gt, int* bad_flag, int miller_only) {
    if ((((size_t)blockIdx.x * BLOCK + threadIdx.x) >> 6) * TRI_PER_WAVE >= n) return;      // whole wavefront idle
    tri t; size_t i; bool active;
    tri_setup(t, i, active, n);
    __shared__ fp4_slot slots[BLOCK];
    slot_fair_set(slots[threadIdx.x].v, 8);                      // plain grid: the two wavefronts of a SIMD take turns (pairing3.hpp C12381_FAIR_SHARE)
    fp4& H = slots[threadIdx.x].v;
    miller3_pair pr[MAX_PROD];
    bool ok = true;
#pragma unroll 1
    for (int j = 0; j < k; ++j) {
        fp2 qx, qy; bool pinf, qinf, okj;
        pair_inputs(pr[j].px, pr[j].py, pinf, qx, qy, qinf, okj, g1 + 96 * ((size_t)j * n + i), g2 + 192 * ((size_t)j * n + i));
        if (!okj) { pinf = true; qinf = true; }
        ok = ok && okj;
        pr[j].skip = pinf;
        miller3_q(pr[j].Q, qx, qy, qinf);
        miller3_tc(pr[j].tc, pr[j].Q, t);
    }
    { fp4 one; f12t_one(one, t); slot_store(H, one); }
    miller3_rangeK(H, pr, k, 64, 1, t);
    f12t_conj_h(H, t);
    fp4 F;
    slot_load(F, H);
    if (!miller_only) f12t_final_exp_ws(F, H, t);
    if (active) {
        if (!ok) { *bad_flag = 1; gt_poison(gt + 576 * i, t.role); }
        else gt_store_coeff(gt + 576 * i, F, t.role);
    }
}

// ------------------------------------------------------------------ work-queue kernels
// A pairing is ~3.4 M instructions per wavefront and every wavefront task is equally long, so a plain grid finishes in
// whole "rounds": 3121 wavefronts (2^16 pairings) on 2048 resident slots take two full double rounds although they
// are 1.52 rounds of work.  Here each group of 21 pairings is TEN tasks — four quarters of the Miller loop and the six
// steps of the final exponentiation (f12t_final_exp_step) — handed out through one atomic counter to a grid that just
// fills the machine; a wavefront that finishes takes the next task, so the tail is a tenth as long.  Tasks are numbered phase-major
// and a task of phase p waits (spins on the group's flag) only for a task of phase p-1, which never waits for anything
// of phase >= p: no cycle, every wavefront reaches the end of the queue.  The spin is bounded as a last line of defence
// (queue_wait above: a time-out poisons the group instead of letting it run on stale state).
template <bool EQ>
__device__ __forceinline__ void pair3_queue_body(size_t n, const uint8_t* a1, const uint8_t* a2, const uint8_t* b1, const uint8_t* b2, size_t b2_stride,
                                                 uint8_t* out, int* bad_flag, uint4* state, unsigned int* flags, unsigned int* counter, int spin_limit, uint32_t epoch,
                                                 fp4& H, unsigned long long* stamps = nullptr, unsigned long long* wstats = nullptr) {
    queue_wave_stats ws(wstats);
    uint8_t* const stw = reinterpret_cast<uint8_t*>(state);
    const unsigned lane = threadIdx.x & 63u;
    const unsigned trip = lane / 3u;
    tri t;
    t.role = lane == 63u ? 0 : (int)(lane - 3u * trip);
    t.base = lane == 63u ? 63 : (int)(3u * trip);
    const size_t ngroups = (n + TRI_PER_WAVE - 1) / TRI_PER_WAVE;
    constexpr unsigned int MILLER_TASKS = MILLER_TASKS_PER_GROUP, TASKS = MILLER_TASKS + 6;
    // Hybrid schedule.  Wavefronts first claim WHOLE groups (counter[1]: no hand-over, no wait on a slower partner, the state stays
    // in registers and in the LDS slot) until only the last third of the groups (queue_direct_groups) is left; those go through the queue in tenth-length
    // tasks (counter[0]), which is what evens out the end of the launch: whole groups finish up to a group-time apart (the two
    // wavefronts of a SIMD do not share it evenly, profiles/r02_queue_phase_times.txt), ~5 small tasks per wavefront absorb that.
    // 2^16 pairings on 2048 resident wavefronts: 2081 whole groups + 1040 queued ones (22.6 -> 20.4 ms with a static split).
    // ndirect = 0 when the batch fits the grid (queue forced on for a small batch: tests of the queue path).
    const size_t nwaves = (size_t)gridDim.x * (BLOCK / 64);
    const size_t ndirect = queue_direct_groups(ngroups, nwaves);
    const unsigned long long ts_entry = stamps ? __builtin_amdgcn_s_memtime() : 0ull;
    for (;;) {
        const unsigned int gc = atomicAdd(counter + 1, lane == 0 ? 1u : 0u);
        const size_t g = (size_t)(unsigned int)__builtin_amdgcn_readfirstlane((int)gc);
        if (g >= ndirect) break;
        const size_t e = g * TRI_PER_WAVE + (lane == 63u ? TRI_PER_WAVE - 1 : trip);
        const unsigned long long ts_g0 = stamps ? __builtin_amdgcn_s_memtime() : 0ull;
        ws.mark();
        pair3_whole_group<EQ>(e < n ? e : n - 1, lane < 63u && e < n, a1, a2, b1, b2, b2_stride, out, bad_flag, H, t);
        ws.whole_done();
        if (stamps && lane == 0) {                             // diagnostic: whole groups behind the queued groups' tasks (entry 10 nq + g)
            unsigned long long* o = stamps + 4 * ((ngroups - ndirect) * (size_t)TASKS + g);
            o[0] = ts_entry; o[1] = ts_g0; o[2] = __builtin_amdgcn_s_memtime();
            o[3] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) | ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32);
        }
    }
    const size_t nq = ngroups - ndirect;                       // queued groups: ndirect .. ngroups - 1
    const size_t ntasks = nq * TASKS;
    for (;;) {
        // lane 0 claims a task; readfirstlane makes the number a scalar, so phase / group and every branch on them are
        // wave-uniform for the compiler too (a broadcast by shuffle leaves them "divergent": the loop was then
        // restructured per lane set, lanes 1..63 re-entered it with task 0 and never left)
        // (every lane issues the atomic, lanes 1..63 add zero: no divergent branch in front of the scalarisation)
        const unsigned int claimed = atomicAdd(counter, lane == 0 ? 1u : 0u);
        const unsigned int task = (unsigned int)__builtin_amdgcn_readfirstlane((int)claimed);
        if ((size_t)task >= ntasks) break;
        const unsigned int p = (unsigned int)(task / nq);
        const size_t g = ndirect + task % nq;
        const size_t e = g * TRI_PER_WAVE + (lane == 63u ? TRI_PER_WAVE - 1 : trip);
        const bool active = lane < 63u && e < n;
        const size_t i = e < n ? e : n - 1;          // inactive lanes shadow the last element: same instruction stream
        // diagnostic time stamps (C12381_PAIR_STAMPS, tools/queue_phase_times.py): claim, start after the wait, end — per task
        unsigned long long ts_claim = 0, ts_start = 0;
        if (stamps) ts_claim = __builtin_amdgcn_s_memtime();
        ws.mark();
        bool poisoned = queue_wait_rlx(flags, g, p, spin_limit);
        ws.wait_done();
        if (stamps) ts_start = __builtin_amdgcn_s_memtime();
        uint8_t* st = wave_uniform(stw + (g - ndirect) * PAIR_QUEUE_STATE_BYTES);        // only the queued groups own a state block (pair_queue_setup)
        const uint32_t tag_in = st_tag(epoch, p - 1u), tag_out = st_tag(epoch, p);     // F always comes from the phase before
        if (!poisoned && p < MILLER_TASKS) {
            fp px, py, px2, py2; fp2 qx, qy, qx2, qy2; bool pinf, qinf, pinf2 = true, qinf2 = true, ok, okb = true;
            pair_inputs(px, py, pinf, qx, qy, qinf, ok, a1 + 96 * i, a2 + 192 * i);
            if (!ok) { pinf = true; qinf = true; }
            g2p Q, Q2;
            fp2 tc, tc2;
            miller3_q(Q, qx, qy, qinf);
            if (EQ) {
                pair_inputs(px2, py2, pinf2, qx2, qy2, qinf2, okb, b1 + 96 * i, b2 + b2_stride * i);
                if (!okb) { pinf2 = true; qinf2 = true; }
                fp ny;
                fp_neg(ny, py2);
                fp_norm1(py2, ny);                       // e(a1, a2) * e(-b1, b2), see pair3_eq_kernel
                miller3_q(Q2, qx2, qy2, qinf2);
            }
            if (p == 0) {
                miller3_tc(tc, Q, t);
                if (EQ) miller3_tc(tc2, Q2, t);
                fp4 one;
                f12t_one(one, t);
                slot_store(H, one);
            } else {
                fp4 f;
                bool got = stw_load<fp4, ST_DW_F>(f, st + STW_F * 1024, lane, tag_in, spin_limit);
                got = stw_load<fp2, ST_DW_TC>(tc, st + STW_TC1 * 1024, lane, tag_in, spin_limit) && got;
                if (EQ) got = stw_load<fp2, ST_DW_TC>(tc2, st + STW_TC2 * 1024, lane, tag_in, spin_limit) && got;
                poisoned = !got;
                slot_store(H, f);
            }
            ws.loaded();
            if (!poisoned) {
                const int hi = 64 - MILLER_ITERS_PER_TASK * (int)p, lo = hi - (MILLER_ITERS_PER_TASK - 1);
                if (EQ) miller3_range2(H, tc, px, py, pinf, Q, tc2, px2, py2, pinf2, Q2, hi, lo, t);
                else miller3_range(H, tc, px, py, pinf, Q, hi, lo, t);
                if (p == MILLER_TASKS - 1) f12t_conj_h(H, t);
                ws.computed();
                {
                    fp4 f;
                    slot_load(f, H);
                    stw_store<fp4, ST_DW_F>(st + STW_F * 1024, lane, f, tag_out);
                }
                if (p < MILLER_TASKS - 1) {
                    stw_store<fp2, ST_DW_TC>(st + STW_TC1 * 1024, lane, tc, tag_out);
                    if (EQ) stw_store<fp2, ST_DW_TC>(st + STW_TC2 * 1024, lane, tc2, tag_out);
                }
            }
        } else if (!poisoned) {
            const int step = (int)(p - MILLER_TASKS);
            fp4 r, y1, aux;                              // aux shares the rows of the (finished) running points
            bool got = stw_load<fp4, ST_DW_F>(r, st + STW_F * 1024, lane, tag_in, spin_limit);
            if (step >= 1) got = stw_load<fp4, ST_DW_F>(y1, st + STW_Y1 * 1024, lane, st_tag(epoch, MILLER_TASKS), spin_limit) && got;       // written by step 0
            if (step == 5) got = stw_load<fp4, ST_DW_F>(aux, st + STW_TC1 * 1024, lane, st_tag(epoch, MILLER_TASKS + 4u), spin_limit) && got;  // written by step 4
            poisoned = !got;
            ws.loaded();
            if (!poisoned) {
                f12t_final_exp_step(step, r, y1, aux, H, t);
                ws.computed();
                if (step < 5) {
                    stw_store<fp4, ST_DW_F>(st + STW_F * 1024, lane, r, tag_out);
                    if (step == 0) stw_store<fp4, ST_DW_F>(st + STW_Y1 * 1024, lane, y1, tag_out);
                    if (step == 4) stw_store<fp4, ST_DW_F>(st + STW_TC1 * 1024, lane, aux, tag_out);
                } else {
                    // validity of this lane's inputs (cheap next to the arithmetic; keeps the state slab free of flags)
                    fp px, py; fp2 qx, qy; bool pinf, qinf, ok, okb = true;
                    pair_inputs(px, py, pinf, qx, qy, qinf, ok, a1 + 96 * i, a2 + 192 * i);
                    if (EQ) pair_inputs(px, py, pinf, qx, qy, qinf, okb, b1 + 96 * i, b2 + b2_stride * i);
                    const bool valid = ok && okb;
                    if (EQ) {
                        const bool one = f12t_is_one(r, t);
                        if (active && t.role == 0) {
                            if (!valid) *bad_flag = 1;
                            out[e] = valid ? (one ? 1 : 0) : 0xff;
                        }
                    } else if (active) {
                        if (!valid) { *bad_flag = 1; gt_poison(out + 576 * e, t.role); }
                        else gt_store_coeff(out + 576 * e, r, t.role);
                    }
                }
            }
        }
        if (poisoned && p == TASKS - 1 && active) {        // the group's state was never completed: 0xff outputs, C12381_E_INTERNAL
            bad_flag[1] = 1;
            if (EQ) { if (t.role == 0) out[e] = 0xff; } else gt_poison(out + 576 * e, t.role);
        }
        queue_publish_rlx(flags, g, p, poisoned, lane);
        ws.task_done();
        if (stamps && lane == 0) {
            unsigned long long* o = stamps + 4 * (size_t)task;
            o[0] = ts_claim; o[1] = ts_start; o[2] = __builtin_amdgcn_s_memtime();
            o[3] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) | ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32);      // HW_ID | XCC_ID
        }
    }
    ws.finish();
}

__global__ void __launch_bounds__(BLOCK, 2) pair3_queue_kernel(size_t n, const uint8_t* g1, const uint8_t* g2, uint8_t* gt, int* bad_flag, uint4* state,
                                                            unsigned int* flags, unsigned int* counter, int spin_limit, unsigned int epoch, unsigned long long* stamps, unsigned long long* wstats) {
    __shared__ fp4_slot slots[BLOCK];
    slot_fair_set(slots[threadIdx.x].v, 0);
    pair3_queue_body<false>(n, g1, g2, nullptr, nullptr, 0, gt, bad_flag, state, flags, counter, spin_limit, epoch, slots[threadIdx.x].v, stamps, wstats);
}
__global__ void __launch_bounds__(BLOCK, 2) pair3_eq_queue_kernel(size_t n, const uint8_t* a1, const uint8_t* a2, const uint8_t* b1, const uint8_t* b2,
                                                               size_t b2_stride, uint8_t* out, int* bad_flag, uint4* state, unsigned int* flags,
                                                               unsigned int* counter, const int32_t* skip_if, int spin_limit, unsigned int epoch) {
    if (skip_if && skip_if[HDR_VALID] != 0) return;
    __shared__ fp4_slot slots[BLOCK];
    slot_fair_set(slots[threadIdx.x].v, 0);
    pair3_queue_body<true>(n, a1, a2, b1, b2, b2_stride, out, bad_flag, state, flags, counter, spin_limit, epoch, slots[threadIdx.x].v);
}

// ------------------------------------------------------------------ the split forms through the same queue
// pair_ate alone (MILLER = true: four quarter-loop tasks per group, output = the Miller value) and the final exponentiation alone
// (MILLER = false: its six steps, input = 576-byte Fp12 values) for batches of more than one machine round: the plain grids of
// miller3_kernel / gt3_op_kernel run 2^16 elements (1.52 rounds of equally long wavefront tasks) as two full rounds.  Same hybrid
// schedule and hand-over protocol as pair3_queue_body; a separate body, so that the pairing kernels' code is untouched.
template <bool MILLER>
__device__ __forceinline__ void split3_queue_body(size_t n, const uint8_t* in1, const uint8_t* in2, uint8_t* out, int* bad_flag, uint4* state,
                                                  unsigned int* flags, unsigned int* counter, int spin_limit, uint32_t epoch, fp4& H,
                                                  unsigned long long* wstats) {
    queue_wave_stats ws(wstats);
    uint8_t* const stw = reinterpret_cast<uint8_t*>(state);
    const unsigned lane = threadIdx.x & 63u;
    const unsigned trip = lane / 3u;
    tri t;
    t.role = lane == 63u ? 0 : (int)(lane - 3u * trip);
    t.base = lane == 63u ? 63 : (int)(3u * trip);
    const size_t ngroups = (n + TRI_PER_WAVE - 1) / TRI_PER_WAVE;
    constexpr unsigned int TASKS = MILLER ? (unsigned)MILLER_TASKS_PER_GROUP : 6u;
    const size_t nwaves = (size_t)gridDim.x * (BLOCK / 64);
    const size_t ndirect = queue_direct_groups(ngroups, nwaves);
    for (;;) {                                                 // whole groups first
        const unsigned int gc = atomicAdd(counter + 1, lane == 0 ? 1u : 0u);
        const size_t g = (size_t)(unsigned int)__builtin_amdgcn_readfirstlane((int)gc);
        if (g >= ndirect) break;
        const size_t e = g * TRI_PER_WAVE + (lane == 63u ? TRI_PER_WAVE - 1 : trip);
        const bool active = lane < 63u && e < n;
        const size_t i = e < n ? e : n - 1;
        ws.mark();
        if (MILLER) {
            fp px, py; fp2 qx, qy; bool pinf, qinf, ok;
            pair_inputs(px, py, pinf, qx, qy, qinf, ok, in1 + 96 * i, in2 + 192 * i);
            if (!ok) { pinf = true; qinf = true; }
            miller3_loop(H, px, py, pinf, qx, qy, qinf, t);
            if (active) {
                if (!ok) { *bad_flag = 1; gt_poison(out + 576 * i, t.role); }
                else { fp4 F; slot_load(F, H); gt_store_coeff(out + 576 * i, F, t.role); }
            }
        } else {
            fp4 r;
            gt_load_coeff(r, in1 + 576 * i, t.role);
            f12t_final_exp_ws(r, H, t);
            if (active) gt_store_coeff(out + 576 * i, r, t.role);
        }
        ws.whole_done();
    }
    const size_t nq = ngroups - ndirect;
    const size_t ntasks = nq * TASKS;
    for (;;) {
        const unsigned int claimed = atomicAdd(counter, lane == 0 ? 1u : 0u);
        const unsigned int task = (unsigned int)__builtin_amdgcn_readfirstlane((int)claimed);
        if ((size_t)task >= ntasks) break;
        const unsigned int p = (unsigned int)(task / nq);
        const size_t g = ndirect + task % nq;
        const size_t e = g * TRI_PER_WAVE + (lane == 63u ? TRI_PER_WAVE - 1 : trip);
        const bool active = lane < 63u && e < n;
        const size_t i = e < n ? e : n - 1;
        ws.mark();
        bool poisoned = queue_wait_rlx(flags, g, p, spin_limit);
        ws.wait_done();
        uint8_t* st = wave_uniform(stw + (g - ndirect) * PAIR_QUEUE_STATE_BYTES);
        const uint32_t tag_in = st_tag(epoch, p - 1u), tag_out = st_tag(epoch, p);
        if (!poisoned && MILLER) {
            fp px, py; fp2 qx, qy; bool pinf, qinf, ok;
            pair_inputs(px, py, pinf, qx, qy, qinf, ok, in1 + 96 * i, in2 + 192 * i);
            if (!ok) { pinf = true; qinf = true; }
            g2p Q;
            fp2 tc;
            miller3_q(Q, qx, qy, qinf);
            if (p == 0) {
                miller3_tc(tc, Q, t);
                fp4 one;
                f12t_one(one, t);
                slot_store(H, one);
            } else {
                fp4 f;
                bool got = stw_load<fp4, ST_DW_F>(f, st + STW_F * 1024, lane, tag_in, spin_limit);
                got = stw_load<fp2, ST_DW_TC>(tc, st + STW_TC1 * 1024, lane, tag_in, spin_limit) && got;
                poisoned = !got;
                slot_store(H, f);
            }
            ws.loaded();
            if (!poisoned) {
                const int hi = 64 - MILLER_ITERS_PER_TASK * (int)p, lo = hi - (MILLER_ITERS_PER_TASK - 1);
                miller3_range(H, tc, px, py, pinf, Q, hi, lo, t);
                ws.computed();
                if (p == TASKS - 1) {
                    f12t_conj_h(H, t);
                    if (active) {
                        if (!ok) { *bad_flag = 1; gt_poison(out + 576 * e, t.role); }
                        else { fp4 F; slot_load(F, H); gt_store_coeff(out + 576 * e, F, t.role); }
                    }
                } else {
                    fp4 f;
                    slot_load(f, H);
                    stw_store<fp4, ST_DW_F>(st + STW_F * 1024, lane, f, tag_out);
                    stw_store<fp2, ST_DW_TC>(st + STW_TC1 * 1024, lane, tc, tag_out);
                }
            }
        } else if (!poisoned) {
            const int step = (int)p;
            fp4 r, y1, aux;
            bool got = true;
            if (step == 0) gt_load_coeff(r, in1 + 576 * i, t.role);
            else got = stw_load<fp4, ST_DW_F>(r, st + STW_F * 1024, lane, tag_in, spin_limit, &ws.reread);
            if (step >= 1) got = stw_load<fp4, ST_DW_F>(y1, st + STW_Y1 * 1024, lane, st_tag(epoch, 0u), spin_limit) && got;      // written by step 0
            if (step == 5) got = stw_load<fp4, ST_DW_F>(aux, st + STW_TC1 * 1024, lane, st_tag(epoch, 4u), spin_limit) && got;    // written by step 4
            poisoned = !got;
            ws.loaded();
            if (!poisoned) {
                f12t_final_exp_step(step, r, y1, aux, H, t);
                ws.computed();
                if (step < 5) {
                    stw_store<fp4, ST_DW_F>(st + STW_F * 1024, lane, r, tag_out);
                    if (step == 0) stw_store<fp4, ST_DW_F>(st + STW_Y1 * 1024, lane, y1, tag_out);
                    if (step == 4) stw_store<fp4, ST_DW_F>(st + STW_TC1 * 1024, lane, aux, tag_out);
                } else if (active) gt_store_coeff(out + 576 * e, r, t.role);
            }
        }
        if (poisoned && p == TASKS - 1 && active) { bad_flag[1] = 1; gt_poison(out + 576 * e, t.role); }
        queue_publish_rlx(flags, g, p, poisoned, lane);
        ws.task_done();
    }
    ws.finish();
}
__global__ void __launch_bounds__(BLOCK, 2) miller3_queue_kernel(size_t n, const uint8_t* g1, const uint8_t* g2, uint8_t* out, int* bad_flag, uint4* state,
                                                              unsigned int* flags, unsigned int* counter, int spin_limit, unsigned int epoch, unsigned long long* wstats) {
    __shared__ fp4_slot slots[BLOCK];
    slot_fair_set(slots[threadIdx.x].v, 4);                      // a quarter of the younger wavefront's iterations at raised priority (pairing3.hpp C12381_FAIR_SHARE)
    split3_queue_body<true>(n, g1, g2, out, bad_flag, state, flags, counter, spin_limit, epoch, slots[threadIdx.x].v, wstats);
}
__global__ void __launch_bounds__(BLOCK, 2) fexp3_queue_kernel(size_t n, const uint8_t* in576, uint8_t* out, int* bad_flag, uint4* state,
                                                            unsigned int* flags, unsigned int* counter, int spin_limit, unsigned int epoch, unsigned long long* wstats) {
    __shared__ fp4_slot slots[BLOCK];
    slot_fair_set(slots[threadIdx.x].v, 4);
    split3_queue_body<false>(n, in576, nullptr, out, bad_flag, state, flags, counter, spin_limit, epoch, slots[threadIdx.x].v, wstats);
}

// ------------------------------------------------------------------ both G2 arguments fixed for the batch
// Coefficient table of one G2 point (pairing3.hpp, 69 lines) with the header of the fixed-base tables (k_fixed.hip):
// header[HDR_VALID] = valid (on the twist, not infinity, in G2), header[HDR_REBUILD] = rebuild requested by fixed_cache_check_kernel.
// (One working lane, but the launch bounds of every kernel in this file: the out-of-line field routines are compiled
// once for all their callers, and a kernel that allowed one wave per SIMD would hand them a 512-register budget.)
// need_g2 != 0: valid only for elements of G2 other than infinity (BBS+ rewrite); need_g2 == 0: any point of the twist
// and infinity (plain pairing against one Q: the table holds exactly the lines the running-point loop would compute).
// header[HDR_RULE] remembers which rule the cached flag was computed under.
// need_g2: bit 0 = the point has to be in G2, bit 2 = keep the records raw (A/B switch C12381_FQ_RAW)
__global__ void __launch_bounds__(BLOCK, 2) g2_lines_table_kernel(const uint8_t* q192, int32_t* buf, int need_g2_flags) {
    const int need_g2 = need_g2_flags & 1;
    const bool raw = (need_g2_flags & 4) != 0;
    if (buf[HDR_REBUILD] == 0 && buf[HDR_RULE] == need_g2 + 1) return;    // cached table is current
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    g2p Q;
    bool inf, ok;
    g2_parse192(Q.x, Q.y, inf, ok, q192);
    fp2_one(Q.z);
    const bool valid = need_g2 ? (ok && !inf && g2_in_subgroup(Q)) : ok;
    buf[HDR_VALID] = valid ? 1 : 0;
    buf[HDR_RULE] = need_g2 + 1;
    if (valid) miller_lines_precompute(buf + HDR_DWORDS, Q.x, Q.y, inf, !raw);
}
// gate[HDR_VALID] = a valid and b valid (the table-driven kernels run), (gate + GATE_OTHER)[HDR_VALID] = the opposite (the generic kernels run)
__global__ void __launch_bounds__(BLOCK, 2) gate_and_kernel(int32_t* gate, const int32_t* a, const int32_t* b) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int both = (a[HDR_VALID] != 0 && b[HDR_VALID] != 0) ? 1 : 0;
    gate[HDR_VALID] = both;
    gate[GATE_OTHER + HDR_VALID] = both ? 0 : 1;          // read as (gate + GATE_OTHER)[HDR_VALID] by kernels that skip on "generic"
}
// ok[i] = [ e(a_i, W) * e(c_i, G) == 1 ] with W, G given by their coefficient tables.  Work queue as above (ten tasks
// per group); the Miller tasks carry only F.  Runs only when run_if[HDR_VALID] != 0.
// TWO: product of two pairings, boolean output; otherwise one pairing per element against the table tabw, GT output
// (an invalid table — Q not on the twist — poisons every output and raises the flag).
template <bool TWO>
__device__ __forceinline__ void pair3_fixed_queue_body(size_t n, const uint8_t* a96, const uint8_t* c96, const int32_t* tabw, const int32_t* tabg,
                                                       uint8_t* out, int* bad_flag, uint4* state, unsigned int* flags, unsigned int* counter,
                                                       int spin_limit, uint32_t epoch, bool table_ok, fp4& H) {
    uint8_t* const stw = reinterpret_cast<uint8_t*>(state);
    const unsigned lane = threadIdx.x & 63u;
    const unsigned trip = lane / 3u;
    tri t;
    t.role = lane == 63u ? 0 : (int)(lane - 3u * trip);
    t.base = lane == 63u ? 63 : (int)(3u * trip);
    const size_t ngroups = (n + TRI_PER_WAVE - 1) / TRI_PER_WAVE;
    constexpr unsigned int MILLER_TASKS = MILLER_TASKS_PER_GROUP, TASKS = MILLER_TASKS + 6;
    // whole groups first, the last third of the groups through the queue (see pair3_queue_body)
    const size_t nwaves = (size_t)gridDim.x * (BLOCK / 64);
    const size_t ndirect = queue_direct_groups(ngroups, nwaves);
    for (;;) {
        const unsigned int gc = atomicAdd(counter + 1, lane == 0 ? 1u : 0u);
        const size_t g = (size_t)(unsigned int)__builtin_amdgcn_readfirstlane((int)gc);
        if (g >= ndirect) break;
        const size_t e = g * TRI_PER_WAVE + (lane == 63u ? TRI_PER_WAVE - 1 : trip);
        const bool active = lane < 63u && e < n;
        const size_t i = e < n ? e : n - 1;
        fp ax, ay, cx, cy; bool ainf, cinf = true, oka, okc = true;
        g1_parse96(ax, ay, ainf, oka, a96 + 96 * i);
        if (!oka || !table_ok) ainf = true;
        if (TWO) {
            g1_parse96(cx, cy, cinf, okc, c96 + 96 * i);
            if (!okc) cinf = true;
        }
        { fp4 one; f12t_one(one, t); slot_store(H, one); }
        if (TWO) miller3_range2_fixed(H, ax, ay, ainf, tabw, cx, cy, cinf, tabg, 64, 1, t);
        else miller3_range_fixed(H, ax, ay, ainf, tabw, 64, 1, t);
        f12t_conj_h(H, t);
        fp4 F;
        slot_load(F, H);
        f12t_final_exp_ws(F, H, t);
        if (TWO) {
            const bool one = f12t_is_one(F, t);
            const bool valid = oka && okc;
            if (active && t.role == 0) {
                if (!valid) *bad_flag = 1;
                out[e] = valid ? (one ? 1 : 0) : 0xff;
            }
        } else if (active) {
            if (!(oka && table_ok)) { *bad_flag = 1; gt_poison(out + 576 * e, t.role); }
            else gt_store_coeff(out + 576 * e, F, t.role);
        }
    }
    const size_t nq = ngroups - ndirect;
    const size_t ntasks = nq * TASKS;
    for (;;) {
        const unsigned int claimed = atomicAdd(counter, lane == 0 ? 1u : 0u);
        const unsigned int task = (unsigned int)__builtin_amdgcn_readfirstlane((int)claimed);
        if ((size_t)task >= ntasks) break;
        const unsigned int p = (unsigned int)(task / nq);
        const size_t g = ndirect + task % nq;
        const size_t e = g * TRI_PER_WAVE + (lane == 63u ? TRI_PER_WAVE - 1 : trip);
        const bool active = lane < 63u && e < n;
        const size_t i = e < n ? e : n - 1;
        bool poisoned = queue_wait_rlx(flags, g, p, spin_limit);
        uint8_t* st = wave_uniform(stw + (g - ndirect) * PAIR_QUEUE_STATE_BYTES);        // only the queued groups own a state block (pair_queue_setup)
        const uint32_t tag_in = st_tag(epoch, p - 1u), tag_out = st_tag(epoch, p);
        fp ax, ay, cx, cy; bool ainf, cinf = true, oka, okc = true;
        if (!poisoned && (p < MILLER_TASKS || p == TASKS - 1)) {
            g1_parse96(ax, ay, ainf, oka, a96 + 96 * i);
            if (!oka || !table_ok) ainf = true;
            if (TWO) {
                g1_parse96(cx, cy, cinf, okc, c96 + 96 * i);
                if (!okc) cinf = true;
            }
        }
        if (!poisoned && p < MILLER_TASKS) {
            {
                fp4 f;
                if (p == 0) f12t_one(f, t); else poisoned = !stw_load<fp4, ST_DW_F>(f, st + STW_F * 1024, lane, tag_in, spin_limit);
                slot_store(H, f);
            }
            if (!poisoned) {
                const int hi = 64 - MILLER_ITERS_PER_TASK * (int)p, lo = hi - (MILLER_ITERS_PER_TASK - 1);
                if (TWO) miller3_range2_fixed(H, ax, ay, ainf, tabw, cx, cy, cinf, tabg, hi, lo, t);
                else miller3_range_fixed(H, ax, ay, ainf, tabw, hi, lo, t);
                if (p == MILLER_TASKS - 1) f12t_conj_h(H, t);
                fp4 f;
                slot_load(f, H);
                stw_store<fp4, ST_DW_F>(st + STW_F * 1024, lane, f, tag_out);
            }
        } else if (!poisoned) {
            const int step = (int)(p - MILLER_TASKS);
            fp4 r, y1, aux;
            bool got = stw_load<fp4, ST_DW_F>(r, st + STW_F * 1024, lane, tag_in, spin_limit);
            if (step >= 1) got = stw_load<fp4, ST_DW_F>(y1, st + STW_Y1 * 1024, lane, st_tag(epoch, MILLER_TASKS), spin_limit) && got;       // written by step 0
            if (step == 5) got = stw_load<fp4, ST_DW_F>(aux, st + STW_TC1 * 1024, lane, st_tag(epoch, MILLER_TASKS + 4u), spin_limit) && got;  // written by step 4
            poisoned = !got;
            if (!poisoned) {
                f12t_final_exp_step(step, r, y1, aux, H, t);
                if (step < 5) {
                    stw_store<fp4, ST_DW_F>(st + STW_F * 1024, lane, r, tag_out);
                    if (step == 0) stw_store<fp4, ST_DW_F>(st + STW_Y1 * 1024, lane, y1, tag_out);
                    if (step == 4) stw_store<fp4, ST_DW_F>(st + STW_TC1 * 1024, lane, aux, tag_out);
                } else if (TWO) {
                    const bool one = f12t_is_one(r, t);
                    const bool valid = oka && okc;
                    if (active && t.role == 0) {
                        if (!valid) *bad_flag = 1;
                        out[e] = valid ? (one ? 1 : 0) : 0xff;
                    }
                } else if (active) {
                    if (!(oka && table_ok)) { *bad_flag = 1; gt_poison(out + 576 * e, t.role); }
                    else gt_store_coeff(out + 576 * e, r, t.role);
                }
            }
        }
        if (poisoned && p == TASKS - 1 && active) {        // the group's state was never completed: 0xff outputs, C12381_E_INTERNAL
            bad_flag[1] = 1;
            if (TWO) { if (t.role == 0) out[e] = 0xff; } else gt_poison(out + 576 * e, t.role);
        }
        queue_publish_rlx(flags, g, p, poisoned, lane);
    }
}
__global__ void __launch_bounds__(BLOCK, 2) pair3_prod_fixed_queue_kernel(size_t n, const uint8_t* a96, const uint8_t* c96, const int32_t* tabw,
                                                                       const int32_t* tabg, uint8_t* out, int* bad_flag, uint4* state,
                                                                       unsigned int* flags, unsigned int* counter, const int32_t* run_if, int spin_limit,
                                                                       unsigned int epoch) {
    if (run_if[HDR_VALID] == 0) return;
    __shared__ fp4_slot slots[BLOCK];
    slot_fair_set(slots[threadIdx.x].v, 0);
    pair3_fixed_queue_body<true>(n, a96, c96, tabw, tabg, out, bad_flag, state, flags, counter, spin_limit, epoch, true, slots[threadIdx.x].v);
}
// gt[i] = e(P_i, Q) for ONE Q given by its coefficient table (header at `buf`, lines behind it)
__global__ void __launch_bounds__(BLOCK, 2) pair3_fixed_queue_kernel(size_t n, const uint8_t* g1_96, const int32_t* buf, uint8_t* gt, int* bad_flag,
                                                                  uint4* state, unsigned int* flags, unsigned int* counter, int spin_limit, unsigned int epoch) {
    __shared__ fp4_slot slots[BLOCK];
    slot_fair_set(slots[threadIdx.x].v, 0);
    pair3_fixed_queue_body<false>(n, g1_96, nullptr, buf + HDR_DWORDS, nullptr, gt, bad_flag, state, flags, counter, spin_limit, epoch, buf[HDR_VALID] != 0, slots[threadIdx.x].v);
}

// ------------------------------------------------------------------ split pairing and GT arithmetic on triples
// pair_ate alone: the Miller value (not canonical as a pairing value, but the same field element as the reference's)
__global__ void __launch_bounds__(BLOCK, 2) miller3_kernel(size_t n, const uint8_t* g1, const uint8_t* g2, uint8_t* out, int* bad_flag) {
    if ((((size_t)blockIdx.x * BLOCK + threadIdx.x) >> 6) * TRI_PER_WAVE >= n) return;
    tri t; size_t i; bool active;
    tri_setup(t, i, active, n);
    fp px, py; fp2 qx, qy; bool pinf, qinf, ok;
    pair_inputs(px, py, pinf, qx, qy, qinf, ok, g1 + 96 * i, g2 + 192 * i);
    if (!ok) { if (active) *bad_flag = 1; pinf = true; qinf = true; }
    __shared__ fp4_slot slots[BLOCK];
    slot_fair_set(slots[threadIdx.x].v, 8);                      // plain grid: the two wavefronts of a SIMD take turns (pairing3.hpp C12381_FAIR_SHARE)
    fp4& H = slots[threadIdx.x].v;
    miller3_loop(H, px, py, pinf, qx, qy, qinf, t);
    if (active) {
        if (!ok) { uint4* q = reinterpret_cast<uint4*>(out + 576 * i + (t.role == 0 ? 384 : (t.role == 1 ? 192 : 0))); for (int j = 0; j < 12; ++j) q[j] = make_uint4(~0u, ~0u, ~0u, ~0u); }
        else { fp4 F; slot_load(F, H); gt_store_coeff(out + 576 * i, F, t.role); }
    }
}
// op 0: a*b (FP12_mul), 1: conj(a), 2: a^e (FP12_pow, e = 32-byte exponent used as given), 3: final exponentiation
// pow_tab (op 2; may be null): 16 Fp4 rows of 64 lanes per wavefront of THIS launch (GT_POW_TAB_ROWS x 64 x 16 bytes each) — the table of
// the windowed ladder, taken when every triple of the wavefront holds a member of the cyclotomic subgroup (pairing3.hpp f12t_pow_window)
__global__ void __launch_bounds__(BLOCK, 2) gt3_op_kernel(int op, size_t n, const uint8_t* a, const uint8_t* b, uint8_t* out, uint4* pow_tab) {
    if ((((size_t)blockIdx.x * BLOCK + threadIdx.x) >> 6) * TRI_PER_WAVE >= n) return;
    tri t; size_t i; bool active;
    tri_setup(t, i, active, n);
    __shared__ fp4_slot slots[BLOCK];
    slot_fair_set(slots[threadIdx.x].v, 8);                      // plain grid: the two wavefronts of a SIMD take turns (pairing3.hpp C12381_FAIR_SHARE)
    fp4& H = slots[threadIdx.x].v;
    fp4 x, r;
    gt_load_coeff(x, a + 576 * i, t.role);
    if (op == 0) { fp4 y; gt_load_coeff(y, b + 576 * i, t.role); f12t_mul(r, x, y, t); }
    else if (op == 1) { f12t_conj(r, x, t); }
    else if (op == 2) {
        uint32_t raw[8], e[8];
        load_raw32(raw, b + 32 * i); scalar_from_raw32(e, raw);
        bool windows = false;
        if (pow_tab) {                                             // wave-uniform
            const bool member = f12t_is_cyclotomic(H, x, t);
            windows = __builtin_amdgcn_ballot_w64(active && !member) == 0;
        }
        if (windows) {
            const unsigned lane = threadIdx.x & 63u;
            uint4* tab = wave_uniform(pow_tab + (((size_t)blockIdx.x * BLOCK + threadIdx.x) >> 6) * (size_t)(GT_POW_TAB_ROWS * 64));
            f12t_pow_window(H, x, e, t,
                            [&](int k, const fp4& v) { st_store<fp4, ST_ROWS_F>(tab + (size_t)k * (ST_ROWS_F * 64), lane, v); },
                            [&](fp4& m, int k) { st_load<fp4, ST_ROWS_F>(m, tab + (size_t)k * (ST_ROWS_F * 64), lane); });
        } else { slot_store(H, x); f12t_pow_generic(H, e, t); }
        slot_load(r, H);
    }
    else { r = x; f12t_final_exp_ws(r, H, t); }
    if (active) gt_store_coeff(out + 576 * i, r, t.role);
}
// The power for batches of more than one machine round: hybrid schedule and hand-over protocol of split3_queue_body, FIVE tasks per queued
// group — 0: membership test, route, table (windowed route), 1..4: a quarter of the ladder each (16 windows, or 64 / 64 / 64 / 65 iterations
// of the reference's digit sequence).  State between tasks: the accumulator (scaled form) and the route (word 0 of the group's second state
// area).  Tables: pow_tab holds one per wavefront of the grid (whole groups), then one per queued group.
constexpr unsigned int GT_POW_TASKS = 5;
__device__ __forceinline__ void gt3_pow_whole(fp4& H, const fp4& x, const uint32_t (&e)[8], bool active, uint4* tab, unsigned lane, const tri& t) {
    const bool member = f12t_is_cyclotomic(H, x, t);
    if (__builtin_amdgcn_ballot_w64(active && !member) == 0) {
        f12t_pow_window(H, x, e, t,
                        [&](int k, const fp4& v) { st_store<fp4, ST_ROWS_F>(tab + (size_t)k * (ST_ROWS_F * 64), lane, v); },
                        [&](fp4& m, int k) { st_load<fp4, ST_ROWS_F>(m, tab + (size_t)k * (ST_ROWS_F * 64), lane); });
    } else { slot_store(H, x); f12t_pow_generic(H, e, t); }
}
__global__ void __launch_bounds__(BLOCK, 2) gt3_pow_queue_kernel(size_t n, const uint8_t* a, const uint8_t* b, uint8_t* out, int* bad_flag, uint4* pow_tab,
                                                              uint4* state, unsigned int* flags, unsigned int* counter, int spin_limit) {
    __shared__ fp4_slot slots[BLOCK];
    fp4& H = slots[threadIdx.x].v;
    slot_fair_set(H, 0);
    const unsigned lane = threadIdx.x & 63u;
    const unsigned trip = lane / 3u;
    tri t;
    t.role = lane == 63u ? 0 : (int)(lane - 3u * trip);
    t.base = lane == 63u ? 63 : (int)(3u * trip);
    const size_t ngroups = (n + TRI_PER_WAVE - 1) / TRI_PER_WAVE;
    constexpr int ROWS = ST_Y1 + ST_ROWS_F;
    const size_t nwaves = (size_t)gridDim.x * (BLOCK / 64);
    const size_t wave = ((size_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
    const size_t ndirect = queue_direct_groups(ngroups, nwaves);
    uint4* own_tab = wave_uniform(pow_tab + wave * (size_t)(GT_POW_TAB_ROWS * 64));
    for (;;) {                                                 // whole groups first
        const unsigned int gc = atomicAdd(counter + 1, lane == 0 ? 1u : 0u);
        const size_t g = (size_t)(unsigned int)__builtin_amdgcn_readfirstlane((int)gc);
        if (g >= ndirect) break;
        const size_t el = g * TRI_PER_WAVE + (lane == 63u ? TRI_PER_WAVE - 1 : trip);
        const bool active = lane < 63u && el < n;
        const size_t i = el < n ? el : n - 1;
        fp4 x, r;
        uint32_t raw[8], e[8];
        gt_load_coeff(x, a + 576 * i, t.role);
        load_raw32(raw, b + 32 * i); scalar_from_raw32(e, raw);
        gt3_pow_whole(H, x, e, active, own_tab, lane, t);
        slot_load(r, H);
        if (active) gt_store_coeff(out + 576 * i, r, t.role);
    }
    const size_t nq = ngroups - ndirect;
    const size_t ntasks = nq * GT_POW_TASKS;
    for (;;) {
        const unsigned int claimed = atomicAdd(counter, lane == 0 ? 1u : 0u);
        const unsigned int task = (unsigned int)__builtin_amdgcn_readfirstlane((int)claimed);
        if ((size_t)task >= ntasks) break;
        const unsigned int p = (unsigned int)(task / nq);
        const size_t gq = task % nq, g = ndirect + gq;
        const size_t el = g * TRI_PER_WAVE + (lane == 63u ? TRI_PER_WAVE - 1 : trip);
        const bool active = lane < 63u && el < n;
        const size_t i = el < n ? el : n - 1;
        const bool poisoned = queue_wait(flags, g, p, spin_limit);
        uint4* st = state + gq * (size_t)ROWS * 64;
        uint4* tab = wave_uniform(pow_tab + (nwaves + gq) * (size_t)(GT_POW_TAB_ROWS * 64));
        if (poisoned) {
            if (p == GT_POW_TASKS - 1 && active) { bad_flag[1] = 1; gt_poison(out + 576 * el, t.role); }
        } else {
            fp4 x;
            uint32_t raw[8], e[8];
            gt_load_coeff(x, a + 576 * i, t.role);
            load_raw32(raw, b + 32 * i); scalar_from_raw32(e, raw);
            if (p == 0) {
                const bool member = f12t_is_cyclotomic(H, x, t);
                const bool windows = __builtin_amdgcn_ballot_w64(active && !member) == 0;
                if (windows) f12t_pow_window_table(H, x, t, [&](int k, const fp4& v) { st_store<fp4, ST_ROWS_F>(tab + (size_t)k * (ST_ROWS_F * 64), lane, v); });
                f12t_pow_acc_init(H, t);
                if (lane == 0) st[ST_TC1 * 64] = make_uint4(windows ? 1u : 0u, 0u, 0u, 0u);
            } else {
                fp4 acc;
                st_load<fp4, ST_ROWS_F>(acc, st + ST_F * 64, lane);
                slot_store(H, acc);
                const bool windows = __builtin_amdgcn_readfirstlane((int)st[ST_TC1 * 64].x) != 0;
                if (windows) {
                    const int whi = 63 - 16 * ((int)p - 1);
                    f12t_pow_window_range(H, e, whi, whi - 15, t, [&](fp4& m, int k) { st_load<fp4, ST_ROWS_F>(m, tab + (size_t)k * (ST_ROWS_F * 64), lane); });
                } else {
                    const int hi = 257 - 64 * ((int)p - 1), lo = p == GT_POW_TASKS - 1 ? 1 : hi - 63;
                    f12t_pow_generic_range(H, x, e, hi, lo, t);
                }
            }
            if (p == GT_POW_TASKS - 1) {
                f12t_unscale3_h(H);
                fp4 r;
                slot_load(r, H);
                if (active) gt_store_coeff(out + 576 * el, r, t.role);
            } else {
                fp4 acc;
                slot_load(acc, H);
                st_store<fp4, ST_ROWS_F>(st + ST_F * 64, lane, acc);
            }
        }
        queue_publish(flags, g, p, poisoned, lane);
    }
}
__global__ void __launch_bounds__(BLOCK, 2) gt3_is_unity_kernel(size_t n, const uint8_t* a, uint8_t* out) {
    if ((((size_t)blockIdx.x * BLOCK + threadIdx.x) >> 6) * TRI_PER_WAVE >= n) return;
    tri t; size_t i; bool active;
    tri_setup(t, i, active, n);
    fp4 x;
    gt_load_coeff(x, a + 576 * i, t.role);
    const bool one = f12t_is_one(x, t);
    if (active && t.role == 0) out[i] = one ? 1 : 0;
}

}  // namespace c12381
