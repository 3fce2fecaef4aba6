// Context, workspaces and the C ABI of the batched BLS12-381 backend for MI355X (gfx950).
// Public interface and reference citations: include/c12381_hip.h.  Kernels: kernels.hpp (k_g1.hip, k_g2gt.hip, k_pair3.hip).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/iterator/transform_iterator.hpp>
#include <new>
#include <thread>
#include <vector>

#include "../../include/c12381_hip.h"
#include "fp.hpp"
#include "g1.hpp"
#include "g2.hpp"
#include "msm.hpp"
#include "fixed_base.hpp"
#include "pairing3.hpp"
#include "kernels.hpp"

using namespace c12381;

namespace {
// Tuning and diagnostic switches exist only in builds with -DC12381_EXPERIMENTS (crypto12381_amd/lib/libc12381_hip_exp.so: tools/, A/B
// runs, tests/test_gpu_variants.py).  The default library reads NO environment variable and contains neither the superseded
// one-lane pairing kernels nor the forced-failure hooks: a stray variable in a caller's environment cannot select another path.
#ifdef C12381_EXPERIMENTS
inline const char* tuning_env(const char* name) { return std::getenv(name); }
#else
inline const char* tuning_env(const char*) { return nullptr; }
#endif
// elements per scalar-mul launch, in machine rounds (one round = the lanes resident at the kernel's occupancy: 256 CUs x 4 SIMDs x
// 64 lanes x waves per SIMD).  EIGHT rounds per launch, not one: a SIMD serves its oldest wavefront first and the younger one only
// fills its stalls (csrc/microbench/issue_mix.hip), so in a launch of exactly the resident size the older workgroup of each CU runs at
// full speed, leaves, and the younger one then runs ALONE with every wait for its table records exposed (SQ_WAIT_ANY 12 % per
// wavefront: + 6 % on the launch).  With more rounds per launch the dispatcher puts a new workgroup beside the one that is left, and only
// the last round runs alone: G1 26.8 -> 25.8-26.1 ms per 2^20, G2 8.04 -> 7.66 ms per 2^17 (profiles/r03_ab_chunk_rounds.txt).
// Table slabs: 2816 B per lane — 2.95 GB for a full G1 launch of 2^20 points, 2.95 GB for a full G2 launch of 2^19 (two lanes per point);
// smaller batches allocate for their own size.
int g_queue_groups_host = 0;                    // experiments builds: C12381_QUEUE_GROUPS (the device copy is set alongside, k_pair3.hip)
constexpr size_t CHUNK_ROUNDS = 8;
constexpr size_t G1_CHUNK = (size_t)65536 * G1_OCC * CHUNK_ROUNDS;
constexpr size_t G2_CHUNK = (size_t)32768 * G2H_OCC * CHUNK_ROUNDS;
constexpr int FLAG_WORDS = 4;                    // device status words (read_flag)
// terms per bucket-method pass (2 * n * windows sort items < 2^31); C12381_MSM_MAX_TERMS lowers it so that tests reach
// the multi-part path with small inputs
const size_t MSM_MAX_TERMS = [] {
    const char* e = tuning_env("C12381_MSM_MAX_TERMS");
    const size_t v = e ? (size_t)std::strtoull(e, nullptr, 10) : 0;
    return v >= 64 && v < ((size_t)1 << 26) ? v : (size_t)1 << 26;
}();
}  // namespace

// ====================================================================== host side
struct c12381_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipStream_t side = nullptr;           // rare fix-up passes run here, overlapped with the next chunk on `stream`
    hipEvent_t ev_side = nullptr;
    std::vector<hipEvent_t> ev_chunk;     // one per chunk of a scalar-mul batch (main -> side dependencies)
    std::vector<hipStream_t> sort_streams; // further streams for the segment sorts of the bucket product (created on first use)
    std::vector<hipEvent_t> sort_events;
    char err[256] = {0};
    enum { WS_TAB, WS_PROJ, WS_PREF, WS_IN0, WS_IN1, WS_OUT, WS_RED0, WS_RED1, WS_BBS_Q, WS_BBS_B, WS_BBS_IN, WS_BBS_WIRE, WS_BBS_WIRE_IN,
           WS_PAIR_ST, WS_POW_ST, WS_FQ_W, WS_FQ_G, WS_FQ_GATE, WS_FQ_P, WS_FB_G2, WS_FB_G1_0, WS_FB_G1_1, WS_FB_G1_2, WS_FB_G1_3, WS_MSM_PTS, WS_MSM_K0, WS_MSM_K1, WS_MSM_V0, WS_MSM_V1, WS_MSM_TMP, WS_MSM_RNG, WS_MSM_BK, WS_MSM_ORD, WS_MSM_OVF, WS_DEC1, WS_DEC2, WS_GT_POW, WS_COUNT };
    void* ws[WS_COUNT] = {nullptr};
    size_t ws_bytes[WS_COUNT] = {0};
    int* d_flag = nullptr;
    int* h_flag = nullptr;          // pinned
    // optional per-kernel timing (HIP events on the context's stream), see c12381_profile()
    bool profiling = false;
    struct ev_pair { hipEvent_t a, b; int kind; };
    std::vector<ev_pair> events;
    // diagnostic (experiments builds, C12381_PAIR_STAMPS): per-task time stamps of the last queue pairing launch, on this context's device
    unsigned long long* stamps = nullptr;
    size_t stamps_tasks = 0;
    // ... followed by 12 words per wavefront of the grid (k_pair3.hip queue_wave_stats) for the last queue launch of pairings, Miller loops or
    // final exponentiations; c12381_sync() writes both regions to the file
    static constexpr size_t STAMP_WAVES = 4096;
    // launch counter of the work-queue kernels whose state travels in tagged words (k_pair3.hip stw_store): 28 bits, never 0
    uint32_t queue_epoch = 0;
};

namespace {

int fail(c12381_ctx* c, hipError_t e, const char* what) {
    std::snprintf(c->err, sizeof c->err, "%s: %s", what, hipGetErrorString(e));
    return C12381_E_HIP;
}
#define HIPCK(c, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail((c), e_, #call); } while (0)

int ensure(c12381_ctx* c, int slot, size_t bytes) {
    if (c->ws_bytes[slot] >= bytes) return 0;
    if (c->ws[slot]) { HIPCK(c, hipFree(c->ws[slot])); c->ws[slot] = nullptr; c->ws_bytes[slot] = 0; }
    hipError_t e = hipMalloc(&c->ws[slot], bytes);
    if (e != hipSuccess) { std::snprintf(c->err, sizeof c->err, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); return C12381_E_NOMEM; }
    c->ws_bytes[slot] = bytes;
    return 0;
}
inline unsigned grid_for(size_t n) { return (unsigned)((n + BLOCK - 1) / BLOCK); }
inline size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

int bind(c12381_ctx* c) {
    if (!c) return C12381_E_ARG;
    HIPCK(c, hipSetDevice(c->device));
    return 0;
}

// HIP-event bracket around a dominant-kernel launch (kind: 0 = g1_mul_kernel, 1 = g1_finish_kernel, ...)
struct timed {
    c12381_ctx* c; int idx = -1;
    timed(c12381_ctx* c_, int kind) : c(c_) {
        if (!c->profiling) return;
        c12381_ctx::ev_pair p; p.kind = kind;
        if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) return;
        (void)hipEventRecord(p.a, c->stream);
        c->events.push_back(p); idx = (int)c->events.size() - 1;
    }
    ~timed() { if (idx >= 0) (void)hipEventRecord(c->events[idx].b, c->stream); }
};

// scalar multiplication of n elements into the projective SoA workspace (stride = padded n)
// (results land at proj[proj_off + i]; pt_stride 96 = per-lane points, 0 = one broadcast point)
// in_g1: the caller asserts every point lies in G1 (C12381_F_IN_SUBGROUP): the [r]phi(P) terms of scalars below x^2 are then
// the point at infinity and the kernel skips them — a membership test as long as a scalar multiplication per such lane
int g1_mul_to_proj(c12381_ctx* c, size_t n, const uint8_t* d_pts, const uint8_t* d_sc, size_t stride, size_t pt_stride = 96,
                   size_t proj_off = 0, const int32_t* skip_if = nullptr, bool in_g1 = false) {
    const size_t chunk = n < G1_CHUNK ? round_up(n, 64) : G1_CHUNK;
    int rc;
    if ((rc = ensure(c, c12381_ctx::WS_TAB, (size_t)G1_TAB_DWORDS * chunk * 4))) return rc;
    if ((rc = ensure(c, c12381_ctx::WS_PROJ, (size_t)3 * NL * stride * 4))) return rc;
    for (size_t off = 0; off < n; off += chunk) {
        const size_t m = n - off < chunk ? n - off : chunk;
        timed tm(c, 0);
        // small_term: the reference's [r]phi(P) for scalars below x^2, inside the kernel (k_g1.hip); in_g1 callers have none to add
        hipLaunchKernelGGL(g1_mul_kernel, dim3(grid_for(m)), dim3(BLOCK), 0, c->stream, m, d_pts + pt_stride * off, pt_stride, d_sc + 32 * off,
                           (int32_t*)c->ws[c12381_ctx::WS_TAB], (int32_t*)c->ws[c12381_ctx::WS_PROJ], stride, proj_off + off, c->d_flag, skip_if,
                           in_g1 ? 0 : 1);
        HIPCK(c, hipGetLastError());
    }
    return 0;
}
// Lanes of the simultaneous inversion (g1_finish_kernel / g2_finish_kernel: lane t converts elements t, t + T, ...): one lane per element up
// to one machine round of single wavefronts (65 536 lanes), then up to FINISH_M elements per lane.  The kernels are latency-bound — an
// inversion is a chain of 28 K dependent instructions whatever the number of lanes running it — so idle SIMDs are cheaper than long lanes:
// 2^18 G2 elements at 16 per lane took 0.50 ms (a quarter wavefront per SIMD, 16 x 15 products behind each inversion).
static size_t finish_lanes(size_t n) {
    size_t per = (n + 65535) / 65536;
    if (per > (size_t)FINISH_M) per = FINISH_M;
    if (per < 1) per = 1;
    const size_t T = round_up((n + per - 1) / per, 64);
    return T > n ? n : T;
}
int g1_finish(c12381_ctx* c, size_t n, const int32_t* proj, size_t stride, uint8_t* d_out, int fmt) {
    int rc;
    if ((rc = ensure(c, c12381_ctx::WS_PREF, (size_t)NL * stride * 4))) return rc;
    const size_t T = finish_lanes(n);
    timed tm(c, 1);
    hipLaunchKernelGGL(g1_finish_kernel, dim3(grid_for(T)), dim3(BLOCK), 0, c->stream, n, proj, stride,
                       (int32_t*)c->ws[c12381_ctx::WS_PREF], d_out, fmt, T);
    HIPCK(c, hipGetLastError());
    return 0;
}
// Status words raised by the kernels since the last read: [0] an input point was not on the curve (its outputs are 0xff),
// [1] a library-internal failure (a work-queue hand-over timed out: the affected outputs are 0xff as well).  Every host
// entry point ends here, so a word raised by an earlier asynchronous _dev call is reported by the next host call or
// c12381_sync() on the same context, whichever comes first — _dev callers separate logical operations with c12381_sync().
int read_flag(c12381_ctx* c) {
    HIPCK(c, hipMemcpyAsync(c->h_flag, c->d_flag, FLAG_WORDS * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipMemsetAsync(c->d_flag, 0, FLAG_WORDS * sizeof(int), c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    if (c->h_flag[1]) {
        std::snprintf(c->err, sizeof c->err, "internal: a pairing work-queue hand-over timed out; the affected outputs are 0xff");
        return C12381_E_INTERNAL;
    }
    return c->h_flag[0] ? C12381_E_POINT : 0;
}
// stage host buffers: copies up to three inputs in, runs body, copies output back
struct staged {
    uint8_t *in0 = nullptr, *in1 = nullptr, *out = nullptr;
};
int stage_in(c12381_ctx* c, staged& s, const void* h0, size_t b0, const void* h1, size_t b1, size_t bout) {
    int rc;
    if ((rc = ensure(c, c12381_ctx::WS_IN0, round_up(b0 ? b0 : 16, 256)))) return rc;
    if ((rc = ensure(c, c12381_ctx::WS_IN1, round_up(b1 ? b1 : 16, 256)))) return rc;
    if ((rc = ensure(c, c12381_ctx::WS_OUT, round_up(bout ? bout : 16, 256)))) return rc;
    s.in0 = (uint8_t*)c->ws[c12381_ctx::WS_IN0]; s.in1 = (uint8_t*)c->ws[c12381_ctx::WS_IN1]; s.out = (uint8_t*)c->ws[c12381_ctx::WS_OUT];
    if (h0 && b0) HIPCK(c, hipMemcpyAsync(s.in0, h0, b0, hipMemcpyHostToDevice, c->stream));
    if (h1 && b1) HIPCK(c, hipMemcpyAsync(s.in1, h1, b1, hipMemcpyHostToDevice, c->stream));
    return 0;
}
int stage_out(c12381_ctx* c, const staged& s, void* hout, size_t bout) {
    HIPCK(c, hipMemcpyAsync(hout, s.out, bout, hipMemcpyDeviceToHost, c->stream));
    return 0;
}

// ev_chunk[0]: the event the bucket product forks its side-stream work from (the scalar-multiplication batches use the same vector per chunk)
static int ensure_fork_event(c12381_ctx* c) {
    if (c->ev_chunk.empty()) {
        hipEvent_t e;
        HIPCK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        c->ev_chunk.push_back(e);
    }
    return 0;
}
// window width: msm_window_bits(n), or C12381_MSM_C = 4..16 (tuning runs)
static int msm_c(size_t n) {
    static const int forced = [] { const char* e = tuning_env("C12381_MSM_C"); const int v = e ? std::atoi(e) : 0; return v >= 4 && v <= 16 ? v : 0; }();
    return forced ? forced : msm_window_bits(n);
}
constexpr int MSM_SORT_STREAMS = 2;      // streams the window segments are sorted on (>= 2: the context's and its side stream; three or four measure the same, profiles/r04_ab_msm_front2.txt)
static_assert(MSM_SORT_STREAMS >= 2 && MSM_SORT_STREAMS <= 8, "MSM_SORT_STREAMS");
// the unsorted value of entry x of a window segment, as the sort's input iterator reads it (msm_entry_value)
struct msm_value_fn { uint32_t n; __host__ __device__ uint32_t operator()(uint32_t x) const { return msm_entry_value(x, n); } };
// Bucket-method MSM (msm.hpp): prep -> radix sort -> bucket sums -> window reduction -> Horner -> affine.
int g1_msm_pippenger(c12381_ctx* c, size_t n, const uint8_t* pts, const uint8_t* sc, uint8_t* out, int fmt, int in_fmt = 96) {
    const int cb = msm_c(n), W = msm_windows(cb);
    // nbk digit buckets + ONE more (index nbk, key W << cb): the points whose scalar is below x^2 (msm.hpp)
    const size_t E = msm_entries(n, W), nb = (size_t)1 << cb, nbk = nb * W, nbx = nbk + 1;
    int rc;
    if ((rc = ensure(c, c12381_ctx::WS_MSM_PTS, (size_t)2 * n * MSM_PT_STRIDE * 4))) return rc;
    if ((rc = ensure(c, c12381_ctx::WS_MSM_K0, E * 4))) return rc;
    if ((rc = ensure(c, c12381_ctx::WS_MSM_K1, E * 4))) return rc;
    if ((rc = ensure(c, c12381_ctx::WS_MSM_V0, E * 4))) return rc;
    if ((rc = ensure(c, c12381_ctx::WS_MSM_V1, E * 4))) return rc;
    if ((rc = ensure(c, c12381_ctx::WS_MSM_RNG, (nbx + 1) * 8))) return rc;
    if ((rc = ensure(c, c12381_ctx::WS_MSM_BK, nbx * G1_ENT_DWORDS * 4))) return rc;
    int32_t* pts2 = (int32_t*)c->ws[c12381_ctx::WS_MSM_PTS];
    uint32_t *k0 = (uint32_t*)c->ws[c12381_ctx::WS_MSM_K0], *k1 = (uint32_t*)c->ws[c12381_ctx::WS_MSM_K1];
    uint32_t *v0 = (uint32_t*)c->ws[c12381_ctx::WS_MSM_V0], *v1 = (uint32_t*)c->ws[c12381_ctx::WS_MSM_V1];
    uint32_t* lo = (uint32_t*)c->ws[c12381_ctx::WS_MSM_RNG];
    uint32_t* hi = lo + nbx + 1;
    int32_t* bk = (int32_t*)c->ws[c12381_ctx::WS_MSM_BK];
    // Sort by digit inside every window segment (msm_prep_one lays the entries out window by window): bits [0, cb) only — two
    // 8-bit passes for cb = 16.  rocPRIM's radix sort is called directly.  From 2^15 terms on: once per segment, on 16-BIT keys (the digit
    // alone: the window is the position) and with the unsorted values supplied by an iterator (msm_entry_value: they are positional too) —
    // round 4: 6 instead of 8 bytes read and written per entry and pass, no value array written by the preparation.  Below that: one
    // call over all entries with the window bits in 32-bit keys (a handful of launches instead of 3 per segment).
    const bool per_window = n >= ((size_t)1 << 15) && cb <= 16;
    if (per_window) {
        uint16_t *q0 = (uint16_t*)k0, *q1 = (uint16_t*)k1;
        hipLaunchKernelGGL(msm_prep16_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, c->stream, n, pts, in_fmt, sc, cb, W, pts2, q0, c->d_flag);
        HIPCK(c, hipGetLastError());
        auto vin = rocprim::make_transform_iterator(rocprim::counting_iterator<uint32_t>(0), msm_value_fn{(uint32_t)n});
        size_t tmp_bytes = 0, tb = 0;
        HIPCK(c, rocprim::radix_sort_pairs(nullptr, tmp_bytes, q0, q1, vin, v1, 2 * n, 0, cb, c->stream));
        HIPCK(c, rocprim::radix_sort_pairs(nullptr, tb, q0, q1, vin, v1, n, 0, 1, c->stream));
        if (tb > tmp_bytes) tmp_bytes = tb;
        // The segments are sorted alternately on the context's stream and on the side stream (own temporary storage each): one sort is
        // six small launches around two passes that reach 1.7 TB/s, two at a time fill the gaps and the memory system better.
        const size_t tmp_half = round_up(tmp_bytes + 256, 256);
        if ((rc = ensure(c, c12381_ctx::WS_MSM_TMP, MSM_SORT_STREAMS * tmp_half))) return rc;
        uint8_t* tmp = (uint8_t*)c->ws[c12381_ctx::WS_MSM_TMP];
        if ((rc = ensure_fork_event(c))) return rc;
        while ((int)c->sort_streams.size() < MSM_SORT_STREAMS - 2) {       // beyond the context's stream and its side stream
            hipStream_t st; hipEvent_t e;
            HIPCK(c, hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
            c->sort_streams.push_back(st);
            HIPCK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
            c->sort_events.push_back(e);
        }
        auto sort_stream = [&](int k) { return k == 0 ? c->stream : (k == 1 ? c->side : c->sort_streams[(size_t)k - 2]); };
        HIPCK(c, hipEventRecord(c->ev_chunk[0], c->stream));               // the keys are written
        for (int k = 1; k < MSM_SORT_STREAMS; ++k) HIPCK(c, hipStreamWaitEvent(sort_stream(k), c->ev_chunk[0], 0));
        for (int w = 0; w <= W; ++w) {                                     // segment W: the small-scalar entries, one key bit
            const size_t off = (size_t)2 * w * n;
            size_t sz = tmp_bytes;
            const int k = w % MSM_SORT_STREAMS;
            HIPCK(c, rocprim::radix_sort_pairs(tmp + (size_t)k * tmp_half, sz, q0 + off, q1 + off, vin, v1 + off, w < W ? 2 * n : n, 0, w < W ? cb : 1, sort_stream(k)));
        }
        for (int k = 1; k < MSM_SORT_STREAMS; ++k) {
            hipEvent_t e = k == 1 ? c->ev_side : c->sort_events[(size_t)k - 2];
            HIPCK(c, hipEventRecord(e, sort_stream(k)));
            HIPCK(c, hipStreamWaitEvent(c->stream, e, 0));
        }
        HIPCK(c, hipMemsetAsync(lo, 0, (nbx + 1) * 8, c->stream));
        hipLaunchKernelGGL(msm_ranges16_kernel, dim3(grid_for((2 * n + MSM_RANGES_PER_THREAD - 1) / MSM_RANGES_PER_THREAD), W + 1), dim3(BLOCK), 0, c->stream,
                           n, (const uint16_t*)q1, cb, W, lo, hi);
        HIPCK(c, hipGetLastError());
    } else {
        hipLaunchKernelGGL(msm_prep_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, c->stream, n, pts, in_fmt, sc, cb, W, pts2, k0, v0, c->d_flag);
        HIPCK(c, hipGetLastError());
        size_t tmp_bytes = 0;
        int end_bit = cb;
        while ((1 << (end_bit - cb)) <= W) ++end_bit;              // keys < (W + 1) << cb
        HIPCK(c, rocprim::radix_sort_pairs(nullptr, tmp_bytes, k0, k1, v0, v1, E, 0, end_bit, c->stream));
        if ((rc = ensure(c, c12381_ctx::WS_MSM_TMP, tmp_bytes + 256))) return rc;
        size_t sz = tmp_bytes;
        HIPCK(c, rocprim::radix_sort_pairs(c->ws[c12381_ctx::WS_MSM_TMP], sz, k0, k1, v0, v1, E, 0, end_bit, c->stream));
        HIPCK(c, hipMemsetAsync(lo, 0, (nbx + 1) * 8, c->stream));
        hipLaunchKernelGGL(msm_ranges_kernel, dim3(grid_for(E)), dim3(BLOCK), 0, c->stream, E, k1, cb, W, lo, hi);
        HIPCK(c, hipGetLastError());
    }
    // buckets in order of decreasing run length (k0 / v0 are free again after the first sort; the sorted size keys go
    // to k1, which the ranges kernel has finished with)
    const size_t key_cap = E > nbx ? E : nbx;
    if (key_cap > E) {
        if ((rc = ensure(c, c12381_ctx::WS_MSM_K0, key_cap * 4))) return rc;
        if ((rc = ensure(c, c12381_ctx::WS_MSM_K1, key_cap * 4))) return rc;
        if ((rc = ensure(c, c12381_ctx::WS_MSM_V0, key_cap * 4))) return rc;
        k0 = (uint32_t*)c->ws[c12381_ctx::WS_MSM_K0]; k1 = (uint32_t*)c->ws[c12381_ctx::WS_MSM_K1]; v0 = (uint32_t*)c->ws[c12381_ctx::WS_MSM_V0];
    }
    if ((rc = ensure(c, c12381_ctx::WS_MSM_ORD, nbx * 4))) return rc;
    uint32_t* order = (uint32_t*)c->ws[c12381_ctx::WS_MSM_ORD];
    // overflow bookkeeping for runs longer than MSM_RUN_CAP (k_g1.hip): counters | segment list | cut-bucket list | partial sums
    const uint32_t run_cap = msm_run_cap(n, cb);
    const size_t ovf_cap = E / (run_cap / 2) + 1;
    const size_t o_seg = 256, o_big = o_seg + round_up(ovf_cap * 8, 256), o_part = o_big + round_up(ovf_cap * 16, 256);
    if ((rc = ensure(c, c12381_ctx::WS_MSM_OVF, o_part + ovf_cap * G1_ENT_DWORDS * 4))) return rc;
    uint8_t* ovf = (uint8_t*)c->ws[c12381_ctx::WS_MSM_OVF];
    uint32_t* ovf_cnt = (uint32_t*)ovf;
    uint2* ovf_seg = (uint2*)(ovf + o_seg);
    uint4* ovf_big = (uint4*)(ovf + o_big);
    int32_t* ovf_part = (int32_t*)(ovf + o_part);
    HIPCK(c, hipMemsetAsync(ovf_cnt, 0, 16, c->stream));
    // The small-scalar bucket (index nbk) and the [r]phi(S) it owes, on the side stream while this one goes on to the bucket sums: ranges and
    // sorted values are final here.  ovf_cnt[2] = "term written"; longer buckets are left to the bucket kernel and msm_small_term_kernel.
    int32_t* small_term = (int32_t*)(ovf + 64);
    if ((rc = ensure_fork_event(c))) return rc;
    HIPCK(c, hipEventRecord(c->ev_chunk[0], c->stream));
    HIPCK(c, hipStreamWaitEvent(c->side, c->ev_chunk[0], 0));
    hipLaunchKernelGGL(msm_small_early_kernel, dim3(1), dim3(64), 0, c->side, (const uint32_t*)lo, (const uint32_t*)hi, (uint32_t)nbk, MSM_SMALL_EARLY_MAX,
                       (const uint32_t*)v1, (const int32_t*)pts2, small_term, ovf_cnt + 2);
    HIPCK(c, hipGetLastError());
    hipLaunchKernelGGL(msm_sizes_kernel, dim3(grid_for(nbx)), dim3(BLOCK), 0, c->stream, nbx, lo, hi, k0, v0, run_cap, ovf_cnt, ovf_seg, ovf_big, MSM_SMALL_EARLY_MAX);
    HIPCK(c, hipGetLastError());
    {   // run-length keys are below 2^bits(cap): one or two passes instead of four
        int kb = 1;
        while (((uint32_t)1 << kb) <= run_cap) ++kb;
        size_t tmp2 = 0;
        HIPCK(c, rocprim::radix_sort_pairs(nullptr, tmp2, k0, k1, v0, order, nbx, 0, kb, c->stream));
        if ((rc = ensure(c, c12381_ctx::WS_MSM_TMP, tmp2 + 256))) return rc;
        HIPCK(c, rocprim::radix_sort_pairs(c->ws[c12381_ctx::WS_MSM_TMP], tmp2, k0, k1, v0, order, nbx, 0, kb, c->stream));
    }
    {
        timed tm(c, 5);
        hipLaunchKernelGGL(msm_bucket_kernel, dim3(grid_for(nbx)), dim3(BLOCK), 0, c->stream, nbx, lo, hi, v1, pts2, bk, order, run_cap, MSM_SMALL_EARLY_MAX);
        HIPCK(c, hipGetLastError());
    }
    // uniform scalars register no overflow segment: both grids leave after reading the counters
    hipLaunchKernelGGL(msm_overflow_kernel, dim3(grid_for(ovf_cap)), dim3(BLOCK), 0, c->stream, (const uint32_t*)ovf_cnt, (const uint2*)ovf_seg, lo, hi, v1, pts2,
                       ovf_part, run_cap);
    HIPCK(c, hipGetLastError());
    hipLaunchKernelGGL(msm_overflow_combine_kernel, dim3(ovf_cap < 4096 ? (unsigned)((ovf_cap + 3) / 4) : 1024u), dim3(BLOCK), 0, c->stream,
                       (const uint32_t*)ovf_cnt, (const uint4*)ovf_big, (const int32_t*)ovf_part, bk);
    HIPCK(c, hipGetLastError());
    // a small-scalar bucket too long for the early kernel: its term from the bucket sum, on the side stream beside the window reductions
    // (returns at once when the early kernel has written the term)
    HIPCK(c, hipEventRecord(c->ev_chunk[0], c->stream));
    HIPCK(c, hipStreamWaitEvent(c->side, c->ev_chunk[0], 0));
    hipLaunchKernelGGL(msm_small_term_kernel, dim3(1), dim3(64), 0, c->side, (const int32_t*)(bk + nbk * G1_ENT_DWORDS), small_term, (const uint32_t*)(ovf_cnt + 2));
    HIPCK(c, hipGetLastError());
    HIPCK(c, hipEventRecord(c->ev_side, c->side));
    const uint32_t chunks = (uint32_t)((nb + MSM_CHUNK - 1) / MSM_CHUNK);
    size_t cur_n = (size_t)W * chunks, cur_stride = round_up(cur_n, 64);
    if ((rc = ensure(c, c12381_ctx::WS_RED0, (size_t)3 * NL * cur_stride * 4))) return rc;
    hipLaunchKernelGGL(msm_wreduce_kernel, dim3(grid_for(cur_n)), dim3(BLOCK), 0, c->stream, W, (uint32_t)nb, chunks, bk,
                       (int32_t*)c->ws[c12381_ctx::WS_RED0], cur_stride);
    HIPCK(c, hipGetLastError());
    // per-window sums: element index = chunk * W + w; every level folds 64 points of a window per wavefront
    const int32_t* cur = (const int32_t*)c->ws[c12381_ctx::WS_RED0];
    int slot = c12381_ctx::WS_RED1;
    while (cur_n > (size_t)W) {
        const size_t groups = cur_n / W, out_groups = (groups + 63) / 64;
        const size_t m = (size_t)W * out_groups, m_stride = round_up(m, 64);
        if ((rc = ensure(c, slot, (size_t)3 * NL * m_stride * 4))) return rc;
        hipLaunchKernelGGL(g1_wave_reduce_kernel, dim3(grid_for(m * 64)), dim3(BLOCK), 0, c->stream, groups, W, cur, cur_stride, (int32_t*)c->ws[slot], m_stride);
        HIPCK(c, hipGetLastError());
        cur = (const int32_t*)c->ws[slot]; cur_n = m; cur_stride = m_stride;
        slot = slot == c12381_ctx::WS_RED0 ? c12381_ctx::WS_RED1 : c12381_ctx::WS_RED0;
    }
    if ((rc = ensure(c, c12381_ctx::WS_PROJ, (size_t)3 * NL * 64 * 4))) return rc;
    HIPCK(c, hipStreamWaitEvent(c->stream, c->ev_side, 0));
    hipLaunchKernelGGL(msm_horner_kernel, dim3(1), dim3(64), 0, c->stream, cur, cur_stride, W, cb, (int32_t*)c->ws[c12381_ctx::WS_PROJ], (size_t)64,
                       (const int32_t*)small_term);
    HIPCK(c, hipGetLastError());
    return g1_finish(c, 1, (const int32_t*)c->ws[c12381_ctx::WS_PROJ], 64, out, fmt);
}
// C12381_MSM=naive forces the n-scalar-muls + tree-sum path (A/B measurements); default: buckets from 2 terms on (both
// paths equal the reference's chain of multiply() calls for every input; the bucket path is the faster one at every size)
static bool msm_use_buckets(size_t n) {
    static const int mode = [] { const char* e = tuning_env("C12381_MSM"); return e ? (e[0] == 'n' ? 1 : (e[0] == 'b' ? 2 : 0)) : 0; }();
    if (mode == 1) return false;
    return n >= 2;
}
}  // namespace

extern "C" {

int c12381_version(void) { return (0 << 16) | 1; }

int c12381_create(int device, c12381_ctx** out) {
    if (!out) return C12381_E_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) return C12381_E_HIP;
    c12381_ctx* c = new (std::nothrow) c12381_ctx;
    if (!c) return C12381_E_NOMEM;
    c->device = device;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return C12381_E_HIP; }
    c->own_stream = true;
    if (hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_side, hipEventDisableTiming) != hipSuccess) { c12381_destroy(c); return C12381_E_HIP; }
    if (hipMalloc((void**)&c->d_flag, FLAG_WORDS * sizeof(int)) != hipSuccess || hipHostMalloc((void**)&c->h_flag, FLAG_WORDS * sizeof(int)) != hipSuccess ||
        hipMemset(c->d_flag, 0, FLAG_WORDS * sizeof(int)) != hipSuccess) { c12381_destroy(c); return C12381_E_HIP; }
    if (const char* e = tuning_env("C12381_QUEUE_GROUPS")) { g_queue_groups_host = std::atoi(e); set_queue_groups_override(g_queue_groups_host); }      // tuning runs only
    *out = c;
    return 0;
}

void c12381_destroy(c12381_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (auto& p : c->events) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    for (int i = 0; i < c12381_ctx::WS_COUNT; ++i) if (c->ws[i]) (void)hipFree(c->ws[i]);
    if (c->stamps) (void)hipFree(c->stamps);
    if (c->d_flag) (void)hipFree(c->d_flag);
    if (c->h_flag) (void)hipHostFree(c->h_flag);
    for (hipStream_t st : c->sort_streams) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
    for (hipEvent_t e : c->sort_events) (void)hipEventDestroy(e);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    if (c->side) (void)hipStreamDestroy(c->side);
    for (hipEvent_t e : c->ev_chunk) (void)hipEventDestroy(e);
    if (c->ev_side) (void)hipEventDestroy(c->ev_side);
    delete c;
}

const char* c12381_last_error(const c12381_ctx* c) { return c ? c->err : "null context"; }

// Workspaces grow to the largest call a context has served and stay (a GT power of 2^16 elements leaves 1.4 GB of tables, 2^18 BBS+ verifications
// a 172 MB state slab, a 2^20 scalar multiplication its 2.95 GB table slab): a long-lived context that has finished with the large batches hands
// them back here; the next call allocates what it needs again.
int c12381_trim(c12381_ctx* c) {
    int rc = bind(c); if (rc) return rc;
    HIPCK(c, hipStreamSynchronize(c->stream));
    if (c->side) HIPCK(c, hipStreamSynchronize(c->side));
    for (hipStream_t st : c->sort_streams) HIPCK(c, hipStreamSynchronize(st));
    for (int i = 0; i < c12381_ctx::WS_COUNT; ++i) {
        if (!c->ws[i]) continue;
        // cached tables whose headers say "valid" live in some of these slots: freeing them only costs a rebuild on the next call that needs them
        HIPCK(c, hipFree(c->ws[i]));
        c->ws[i] = nullptr; c->ws_bytes[i] = 0;
    }
    return 0;
}

int c12381_set_stream(c12381_ctx* c, void* hip_stream) {
    int rc = bind(c); if (rc) return rc;
    HIPCK(c, hipStreamSynchronize(c->stream));
    if (c->own_stream) { HIPCK(c, hipStreamDestroy(c->stream)); c->own_stream = false; }
    if (hip_stream) { c->stream = (hipStream_t)hip_stream; }
    else { HIPCK(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)); c->own_stream = true; }
    return 0;
}

static void pair_stamps_dump(c12381_ctx* c);
int c12381_sync(c12381_ctx* c) {
    int rc = bind(c); if (rc) return rc;
    rc = read_flag(c);
    pair_stamps_dump(c);
    return rc;
}

// Ordering against the caller's other streams without blocking the host (include/c12381_hip.h "Stream ordering").  Side-stream work of
// earlier calls is always joined back into the context's stream by the call that started it (ev_side), so the context's stream alone
// carries the completion of everything launched so far.
int c12381_wait_event(c12381_ctx* c, void* hip_event) {
    int rc = bind(c); if (rc) return rc;
    if (!hip_event) return C12381_E_ARG;
    HIPCK(c, hipStreamWaitEvent(c->stream, (hipEvent_t)hip_event, 0));
    return 0;
}
int c12381_record_event(c12381_ctx* c, void* hip_event) {
    int rc = bind(c); if (rc) return rc;
    if (!hip_event) return C12381_E_ARG;
    HIPCK(c, hipEventRecord((hipEvent_t)hip_event, c->stream));
    return 0;
}

#ifdef C12381_EXPERIMENTS
// experiments builds only (not declared in include/c12381_hip.h): start the clock probe on a stream of its own; `out` = 2 n device words
extern "C" int c12381_exp_clock_probe(c12381_ctx* c, unsigned long long* out, int n, int gap) {
    int rc = bind(c); if (rc) return rc;
    if (!out || n <= 0 || gap < 0) return C12381_E_ARG;
    static hipStream_t probe_stream = nullptr;          // its own stream: G1 / MSM work waits for the context's side stream
    if (!probe_stream) HIPCK(c, hipStreamCreateWithFlags(&probe_stream, hipStreamNonBlocking));
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(BLOCK), 0, probe_stream, out, n, gap);
    HIPCK(c, hipGetLastError());
    return 0;
}
#endif
int c12381_profile(c12381_ctx* c, int enable) {
    int rc = bind(c); if (rc) return rc;
    HIPCK(c, hipStreamSynchronize(c->stream));
    for (auto& p : c->events) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    c->events.clear();
    c->profiling = enable != 0;
    return 0;
}
int c12381_profile_read(c12381_ctx* c, int kind, double* total_ms, uint64_t* launches) {
    int rc = bind(c); if (rc) return rc;
    if (!total_ms || !launches) return C12381_E_ARG;
    HIPCK(c, hipStreamSynchronize(c->stream));
    double ms = 0; uint64_t cnt = 0;
    for (auto& p : c->events) {
        if (p.kind != kind) continue;
        float t = 0;
        HIPCK(c, hipEventElapsedTime(&t, p.a, p.b));
        ms += t; ++cnt;
    }
    *total_ms = ms; *launches = cnt;
    return 0;
}

// ---------------------------------------------------------------- Fp
int c12381_fp_op_batch_dev(c12381_ctx* c, int op, size_t n, const uint8_t* a, const uint8_t* b, uint8_t* out) {
    int rc = bind(c); if (rc) return rc;
    if (op < 0 || op > 5 || !a || !out || (op <= 2 && !b)) return C12381_E_ARG;
    if (n == 0) return 0;
    hipLaunchKernelGGL(fp_op_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, c->stream, op, n, a, op <= 2 ? b : nullptr, out);
    HIPCK(c, hipGetLastError());
    return 0;
}
int c12381_fp_op_batch(c12381_ctx* c, int op, size_t n, const uint8_t* a, const uint8_t* b, uint8_t* out) {
    int rc = bind(c); if (rc) return rc;
    if (op < 0 || op > 5 || !a || !out || (op <= 2 && !b)) return C12381_E_ARG;
    if (n == 0) return 0;
    staged s;
    if ((rc = stage_in(c, s, a, 48 * n, op <= 2 ? b : nullptr, op <= 2 ? 48 * n : 0, 48 * n))) return rc;
    if ((rc = c12381_fp_op_batch_dev(c, op, n, s.in0, s.in1, s.out))) return rc;
    if ((rc = stage_out(c, s, out, 48 * n))) return rc;
    return read_flag(c);
}
int c12381_fp_mulchain_dev(c12381_ctx* c, size_t n, int iters, const uint8_t* a, const uint8_t* b, uint8_t* out) {
    int rc = bind(c); if (rc) return rc;
    if (!a || !b || !out || iters < 0) return C12381_E_ARG;
    if (n == 0) return 0;
    hipLaunchKernelGGL(fp_mulchain_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, c->stream, n, iters, a, b, out);
    HIPCK(c, hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------- G1
// C12381_F_COMPRESSED_IN: pts are n x 49 bytes (the serialized form, g1_point.hpp:87-111 -> ECP_fromOctet): from_bytes -> multiply ->
// to_bytes in ONE kernel — the square root runs in the kernel's prologue; a rejected encoding is a lane of 0xff + C12381_E_POINT
int c12381_g1_mul_batch_flags_dev(c12381_ctx* c, size_t n, const uint8_t* pts, const uint8_t* sc, uint8_t* out, int fmt, unsigned flags) {
    int rc = bind(c); if (rc) return rc;
    if (!pts || !sc || !out || (fmt != 49 && fmt != 96) || (flags & ~(unsigned)(C12381_F_IN_SUBGROUP | C12381_F_COMPRESSED_IN))) return C12381_E_ARG;
    if (n == 0) return 0;
    const size_t stride = round_up(n, 64);
    if ((rc = g1_mul_to_proj(c, n, pts, sc, stride, (flags & C12381_F_COMPRESSED_IN) ? 49 : 96, 0, nullptr, (flags & C12381_F_IN_SUBGROUP) != 0))) return rc;
    return g1_finish(c, n, (const int32_t*)c->ws[c12381_ctx::WS_PROJ], stride, out, fmt);
}
int c12381_g1_mul_batch_dev(c12381_ctx* c, size_t n, const uint8_t* pts, const uint8_t* sc, uint8_t* out, int fmt) {
    return c12381_g1_mul_batch_flags_dev(c, n, pts, sc, out, fmt, 0u);
}
int c12381_g1_mul_batch_flags(c12381_ctx* c, size_t n, const uint8_t* pts, const uint8_t* sc, uint8_t* out, int fmt, unsigned flags) {
    int rc = bind(c); if (rc) return rc;
    if (!pts || !sc || !out || (fmt != 49 && fmt != 96) || (flags & ~(unsigned)(C12381_F_IN_SUBGROUP | C12381_F_COMPRESSED_IN))) return C12381_E_ARG;
    if (n == 0) return 0;
    staged s;
    if ((rc = stage_in(c, s, pts, ((flags & C12381_F_COMPRESSED_IN) ? 49 : 96) * n, sc, 32 * n, (size_t)fmt * n))) return rc;
    if ((rc = c12381_g1_mul_batch_flags_dev(c, n, s.in0, s.in1, s.out, fmt, flags))) return rc;
    if ((rc = stage_out(c, s, out, (size_t)fmt * n))) return rc;
    return read_flag(c);
}
int c12381_g1_mul_batch(c12381_ctx* c, size_t n, const uint8_t* pts, const uint8_t* sc, uint8_t* out, int fmt) {
    return c12381_g1_mul_batch_flags(c, n, pts, sc, out, fmt, 0u);
}
int c12381_g1_add_batch(c12381_ctx* c, size_t n, const uint8_t* a, const uint8_t* b, uint8_t* out, int fmt) {
    int rc = bind(c); if (rc) return rc;
    if (!a || !b || !out || (fmt != 49 && fmt != 96)) return C12381_E_ARG;
    if (n == 0) return 0;
    staged s;
    if ((rc = stage_in(c, s, a, 96 * n, b, 96 * n, (size_t)fmt * n))) return rc;
    const size_t stride = round_up(n, 64);
    if ((rc = ensure(c, c12381_ctx::WS_PROJ, (size_t)3 * NL * stride * 4))) return rc;
    hipLaunchKernelGGL(g1_add_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, c->stream, n, s.in0, s.in1, (int32_t*)c->ws[c12381_ctx::WS_PROJ], stride,
                       c->d_flag);
    HIPCK(c, hipGetLastError());
    if ((rc = g1_finish(c, n, (const int32_t*)c->ws[c12381_ctx::WS_PROJ], stride, s.out, fmt))) return rc;
    if ((rc = stage_out(c, s, out, (size_t)fmt * n))) return rc;
    return read_flag(c);
}

// MSM: the bucket method (g1_msm_pippenger); a single term (or C12381_MSM=naive) takes n independent GLV scalar
// multiplications followed by a tree sum of the projective results (the reference's Π is also n full scalar-muls,
// g1_point.hpp:389-401).  Both equal the reference's chain for every input.  Only the final point is canonical.
// out = sum of n affine points (no scalars): the header's product over G1Point values (a chain of add(point1&, point1&),
// src/miracl_core_interface.cpp:129-132 -> ECP_add) and the combine step of a product that was sharded over GPUs:
// lift + tree sum + one affine conversion — tens of microseconds for the 8 partial points of a node, where the bucket method's
// fixed stages cost 2.3 ms.
int c12381_g1_sum_dev(c12381_ctx* c, size_t n, const uint8_t* pts, uint8_t* out, int fmt) {
    int rc = bind(c); if (rc) return rc;
    if (!out || (n && !pts) || (fmt != 49 && fmt != 96)) return C12381_E_ARG;
    if (n == 0) { HIPCK(c, hipMemsetAsync(out, 0, fmt, c->stream)); return 0; }
    const size_t stride = round_up(n, 64);
    if ((rc = ensure(c, c12381_ctx::WS_PROJ, (size_t)3 * NL * stride * 4))) return rc;
    hipLaunchKernelGGL(g1_lift_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, c->stream, n, pts, (int32_t*)c->ws[c12381_ctx::WS_PROJ], stride, c->d_flag);
    HIPCK(c, hipGetLastError());
    const int32_t* cur = (const int32_t*)c->ws[c12381_ctx::WS_PROJ];
    size_t cur_n = n, cur_stride = stride;
    int slot = c12381_ctx::WS_RED0;
    while (cur_n > 1) {
        const size_t m = cur_n > 4096 ? round_up(cur_n / 32, 64) : (cur_n > 64 ? 64 : 1);
        const size_t m_stride = round_up(m, 64);
        if ((rc = ensure(c, slot, (size_t)3 * NL * m_stride * 4))) return rc;
        hipLaunchKernelGGL(g1_reduce_kernel, dim3(grid_for(m)), dim3(BLOCK), 0, c->stream, cur_n, cur, cur_stride, m, (int32_t*)c->ws[slot], m_stride);
        HIPCK(c, hipGetLastError());
        cur = (const int32_t*)c->ws[slot]; cur_n = m; cur_stride = m_stride;
        slot = slot == c12381_ctx::WS_RED0 ? c12381_ctx::WS_RED1 : c12381_ctx::WS_RED0;
    }
    return g1_finish(c, 1, cur, cur_stride, out, fmt);
}
int c12381_g1_sum(c12381_ctx* c, size_t n, const uint8_t* pts, uint8_t* out, int fmt) {
    int rc = bind(c); if (rc) return rc;
    if (!out || (n && !pts) || (fmt != 49 && fmt != 96)) return C12381_E_ARG;
    staged s;
    if ((rc = stage_in(c, s, pts, 96 * n, nullptr, 0, (size_t)fmt))) return rc;
    if ((rc = c12381_g1_sum_dev(c, n, s.in0, s.out, fmt))) return rc;
    if ((rc = stage_out(c, s, out, (size_t)fmt))) return rc;
    return read_flag(c);
}
int c12381_g1_msm_flags_dev(c12381_ctx* c, size_t n, const uint8_t* pts, const uint8_t* sc, uint8_t* out, int fmt, unsigned flags) {
    int rc = bind(c); if (rc) return rc;
    if (!out || (n && (!pts || !sc)) || (fmt != 49 && fmt != 96) || (flags & ~(unsigned)C12381_F_COMPRESSED_IN)) return C12381_E_ARG;
    const int in_fmt = (flags & C12381_F_COMPRESSED_IN) ? 49 : 96;        // compressed terms are decoded by the preparation kernel
    if (n == 0) { HIPCK(c, hipMemsetAsync(out, 0, fmt, c->stream)); return 0; }
    if (n > MSM_MAX_TERMS) {
        // the sort works on 32-bit item counts and (term, half) values: larger products are cut into parts whose
        // partial points (affine, WS_BBS_B as a small staging slot) are summed by c12381_g1_sum_dev
        const size_t parts = (n + MSM_MAX_TERMS - 1) / MSM_MAX_TERMS;
        if ((rc = ensure(c, c12381_ctx::WS_BBS_B, round_up(parts * 96, 256)))) return rc;
        uint8_t* pp = (uint8_t*)c->ws[c12381_ctx::WS_BBS_B];
        for (size_t p = 0; p < parts; ++p) {
            const size_t lo = p * MSM_MAX_TERMS, m = n - lo < MSM_MAX_TERMS ? n - lo : MSM_MAX_TERMS;
            if ((rc = g1_msm_pippenger(c, m, pts + (size_t)in_fmt * lo, sc + 32 * lo, pp + 96 * p, 96, in_fmt))) return rc;
        }
        return c12381_g1_sum_dev(c, parts, pp, out, fmt);
    }
    if (msm_use_buckets(n)) return g1_msm_pippenger(c, n, pts, sc, out, fmt, in_fmt);
    const size_t stride = round_up(n, 64);
    if ((rc = g1_mul_to_proj(c, n, pts, sc, stride, (size_t)in_fmt))) return rc;
    const int32_t* cur = (const int32_t*)c->ws[c12381_ctx::WS_PROJ];
    size_t cur_n = n, cur_stride = stride;
    int slot = c12381_ctx::WS_RED0;
    while (cur_n > 1) {
        size_t m = cur_n > 4096 ? round_up(cur_n / 32, 64) : (cur_n > 64 ? 64 : 1);
        const size_t m_stride = round_up(m, 64);
        if ((rc = ensure(c, slot, (size_t)3 * NL * m_stride * 4))) return rc;
        hipLaunchKernelGGL(g1_reduce_kernel, dim3(grid_for(m)), dim3(BLOCK), 0, c->stream, cur_n, cur, cur_stride, m, (int32_t*)c->ws[slot], m_stride);
        HIPCK(c, hipGetLastError());
        cur = (const int32_t*)c->ws[slot]; cur_n = m; cur_stride = m_stride;
        slot = slot == c12381_ctx::WS_RED0 ? c12381_ctx::WS_RED1 : c12381_ctx::WS_RED0;
    }
    return g1_finish(c, 1, cur, cur_stride, out, fmt);
}
int c12381_g1_msm_dev(c12381_ctx* c, size_t n, const uint8_t* pts, const uint8_t* sc, uint8_t* out, int fmt) {
    return c12381_g1_msm_flags_dev(c, n, pts, sc, out, fmt, 0u);
}
int c12381_g1_msm_flags(c12381_ctx* c, size_t n, const uint8_t* pts, const uint8_t* sc, uint8_t* out, int fmt, unsigned flags) {
    int rc = bind(c); if (rc) return rc;
    if (!out || (n && (!pts || !sc)) || (fmt != 49 && fmt != 96) || (flags & ~(unsigned)C12381_F_COMPRESSED_IN)) return C12381_E_ARG;
    staged s;
    if ((rc = stage_in(c, s, pts, ((flags & C12381_F_COMPRESSED_IN) ? 49 : 96) * n, sc, 32 * n, (size_t)fmt))) return rc;
    if ((rc = c12381_g1_msm_flags_dev(c, n, s.in0, s.in1, s.out, fmt, flags))) return rc;
    if ((rc = stage_out(c, s, out, (size_t)fmt))) return rc;
    return read_flag(c);
}
int c12381_g1_msm(c12381_ctx* c, size_t n, const uint8_t* pts, const uint8_t* sc, uint8_t* out, int fmt) {
    return c12381_g1_msm_flags(c, n, pts, sc, out, fmt, 0u);
}
// sum_of_products(point1&, int n, point1*, const big*) exactly as the boundary defines it (-> ECP_muln, a plain Pippenger): the sum
// of the TRUE multiples [k_i mod r]P_i for any curve points.  On G1 it equals c12381_g1_msm — use that for throughput; this entry
// exists so that the seam function has the reference's value for every input (n plain ladders + tree sum; the seam is scalar).
int c12381_g1_sum_of_products_dev(c12381_ctx* c, size_t n, const uint8_t* pts, const uint8_t* sc, uint8_t* out, int fmt) {
    int rc = bind(c); if (rc) return rc;
    if (!out || (n && (!pts || !sc)) || (fmt != 49 && fmt != 96)) return C12381_E_ARG;
    if (n == 0) { HIPCK(c, hipMemsetAsync(out, 0, fmt, c->stream)); return 0; }
    const size_t stride = round_up(n, 64);
    if ((rc = ensure(c, c12381_ctx::WS_PROJ, (size_t)3 * NL * stride * 4))) return rc;
    hipLaunchKernelGGL(g1_mul_plain_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, c->stream, n, pts, sc, (int32_t*)c->ws[c12381_ctx::WS_PROJ], stride, c->d_flag);
    HIPCK(c, hipGetLastError());
    const int32_t* cur = (const int32_t*)c->ws[c12381_ctx::WS_PROJ];
    size_t cur_n = n, cur_stride = stride;
    int slot = c12381_ctx::WS_RED0;
    while (cur_n > 1) {
        const size_t m = cur_n > 4096 ? round_up(cur_n / 32, 64) : (cur_n > 64 ? 64 : 1);
        const size_t m_stride = round_up(m, 64);
        if ((rc = ensure(c, slot, (size_t)3 * NL * m_stride * 4))) return rc;
        hipLaunchKernelGGL(g1_reduce_kernel, dim3(grid_for(m)), dim3(BLOCK), 0, c->stream, cur_n, cur, cur_stride, m, (int32_t*)c->ws[slot], m_stride);
        HIPCK(c, hipGetLastError());
        cur = (const int32_t*)c->ws[slot]; cur_n = m; cur_stride = m_stride;
        slot = slot == c12381_ctx::WS_RED0 ? c12381_ctx::WS_RED1 : c12381_ctx::WS_RED0;
    }
    return g1_finish(c, 1, cur, cur_stride, out, fmt);
}
int c12381_g1_sum_of_products(c12381_ctx* c, size_t n, const uint8_t* pts, const uint8_t* sc, uint8_t* out, int fmt) {
    int rc = bind(c); if (rc) return rc;
    if (!out || (n && (!pts || !sc)) || (fmt != 49 && fmt != 96)) return C12381_E_ARG;
    staged s;
    if ((rc = stage_in(c, s, pts, 96 * n, sc, 32 * n, (size_t)fmt))) return rc;
    if ((rc = c12381_g1_sum_of_products_dev(c, n, s.in0, s.in1, s.out, fmt))) return rc;
    if ((rc = stage_out(c, s, out, (size_t)fmt))) return rc;
    return read_flag(c);
}
// One host process driving several GPUs: terms are split contiguously over the contexts, every
// context runs its local MSM on its own device from its own host thread, and the partial points (96 B each) are
// summed on the first context — the elliptic-curve "all-reduce" has no RCCL reduction op, the payload is ngpu x 96 B.
// (One process per GPU with torch.distributed does the same through an all-gather: crypto12381_amd/distributed.py.)
int c12381_g1_msm_multi(c12381_ctx** ctxs, int ngpu, size_t n, const uint8_t* pts, const uint8_t* sc, uint8_t* out, int fmt) {
    if (!ctxs || ngpu <= 0 || !out || (n && (!pts || !sc)) || (fmt != 49 && fmt != 96)) return C12381_E_ARG;
    for (int g = 0; g < ngpu; ++g) if (!ctxs[g]) return C12381_E_ARG;
    if (ngpu == 1) return c12381_g1_msm(ctxs[0], n, pts, sc, out, fmt);
    std::vector<uint8_t> partial((size_t)96 * ngpu, 0);
    std::vector<int> rcs(ngpu, 0);
    std::vector<std::thread> th;
    for (int g = 0; g < ngpu; ++g) {
        const size_t lo = n * (size_t)g / (size_t)ngpu, hi = n * (size_t)(g + 1) / (size_t)ngpu;
        th.emplace_back([=, &partial, &rcs] { rcs[g] = c12381_g1_msm(ctxs[g], hi - lo, pts + 96 * lo, sc + 32 * lo, &partial[(size_t)96 * g], 96); });
    }
    for (auto& t : th) t.join();
    for (int g = 0; g < ngpu; ++g) if (rcs[g]) return rcs[g];
    return c12381_g1_sum(ctxs[0], (size_t)ngpu, partial.data(), out, fmt);
}

// ---------------------------------------------------------------- G2
// finish = true: the kernel leaves projective results in WS_PROJ (the caller has sized it: 6 NL x round_up(n, 64) dwords)
// and g2_finish converts them with one inversion per FINISH_M elements; false: per-lane conversion straight to `out`.
static int g2_mul_dev_strided(c12381_ctx* c, size_t n, const uint8_t* pts, size_t pt_stride, const uint8_t* sc, uint8_t* out, int fmt,
                              const int32_t* skip_if = nullptr, bool finish = false, bool in_g2 = false);
static int g2_finish(c12381_ctx* c, size_t n, uint8_t* d_out, int fmt) {
    int rc;
    const size_t stride = round_up(n, 64);
    if ((rc = ensure(c, c12381_ctx::WS_PREF, (size_t)2 * NL * stride * 4))) return rc;
    const size_t T = finish_lanes(n);
    hipLaunchKernelGGL(g2_finish_kernel, dim3(grid_for(T)), dim3(BLOCK), 0, c->stream, n, (const int32_t*)c->ws[c12381_ctx::WS_PROJ], stride,
                       (int32_t*)c->ws[c12381_ctx::WS_PREF], d_out, fmt, T);
    HIPCK(c, hipGetLastError());
    return 0;
}
// C12381_F_COMPRESSED_IN: pts are n x 97 bytes (g2_point.hpp:73-77 -> ECP2_fromOctet), decoded in the kernel's prologue
int c12381_g2_mul_batch_flags_dev(c12381_ctx* c, size_t n, const uint8_t* pts, const uint8_t* sc, uint8_t* out, int fmt, unsigned flags) {
    int rc = bind(c); if (rc) return rc;
    if (!pts || !sc || !out || (fmt != 97 && fmt != 192) || (flags & ~(unsigned)(C12381_F_IN_SUBGROUP | C12381_F_COMPRESSED_IN))) return C12381_E_ARG;
    if (n == 0) return 0;
    if ((rc = ensure(c, c12381_ctx::WS_PROJ, (size_t)6 * NL * round_up(n, 64) * 4))) return rc;
    if ((rc = g2_mul_dev_strided(c, n, pts, (flags & C12381_F_COMPRESSED_IN) ? 97 : 192, sc, out, fmt, nullptr, true, (flags & C12381_F_IN_SUBGROUP) != 0))) return rc;
    return g2_finish(c, n, out, fmt);
}
int c12381_g2_mul_batch_dev(c12381_ctx* c, size_t n, const uint8_t* pts, const uint8_t* sc, uint8_t* out, int fmt) {
    return c12381_g2_mul_batch_flags_dev(c, n, pts, sc, out, fmt, 0u);
}
static int g2_mul_dev_strided(c12381_ctx* c, size_t n, const uint8_t* pts, size_t pt_stride, const uint8_t* sc, uint8_t* out, int fmt,
                              const int32_t* skip_if, bool finish, bool in_g2) {
    int rc = bind(c); if (rc) return rc;
    if (!pts || !sc || !out || (fmt != 97 && fmt != 192)) return C12381_E_ARG;
    if (n == 0) return 0;
    const size_t chunk = n < G2_CHUNK ? round_up(n, 64) : G2_CHUNK;
    // C12381_G2_LANES=1 keeps the one-lane-per-point kernel for the batch entry points (A/B measurements); default:
    // two lanes per point (k_g2h.hip), whose per-lane table records are those of G1 (2 x 1408 B per point)
    static const bool two_lanes = [] { const char* e = tuning_env("C12381_G2_LANES"); return !(e && e[0] == '1'); }();
    const bool pairwise = finish && two_lanes;
    if ((rc = ensure(c, c12381_ctx::WS_TAB, (size_t)(pairwise ? 2 * G2_TAB * G1_ENT_DWORDS : G2_TAB_DWORDS) * chunk * 4))) return rc;
    int32_t* proj = finish ? (int32_t*)c->ws[c12381_ctx::WS_PROJ] : nullptr;
    for (size_t off = 0; off < n; off += chunk) {
        const size_t m = n - off < chunk ? n - off : chunk;
        timed tm(c, 2);
        if (pairwise) {
            hipLaunchKernelGGL(g2_mul2_kernel, dim3(grid_for(2 * m)), dim3(BLOCK), 0, c->stream, m, pts + pt_stride * off, pt_stride, sc + 32 * off,
                               (int32_t*)c->ws[c12381_ctx::WS_TAB], c->d_flag, skip_if, proj, round_up(n, 64), off, in_g2 ? 1 : 0);
            HIPCK(c, hipGetLastError());
            continue;
        }
        hipLaunchKernelGGL(g2_mul_kernel, dim3(grid_for(m)), dim3(BLOCK), 0, c->stream, m, pts + pt_stride * off, pt_stride, sc + 32 * off,
                           (int32_t*)c->ws[c12381_ctx::WS_TAB], chunk, out + (size_t)fmt * off, fmt, c->d_flag, skip_if, proj, round_up(n, 64), off, in_g2 ? 1 : 0);
        HIPCK(c, hipGetLastError());
    }
    return 0;
}
int c12381_g2_mul_batch_flags(c12381_ctx* c, size_t n, const uint8_t* pts, const uint8_t* sc, uint8_t* out, int fmt, unsigned flags) {
    int rc = bind(c); if (rc) return rc;
    if (!pts || !sc || !out || (fmt != 97 && fmt != 192) || (flags & ~(unsigned)(C12381_F_IN_SUBGROUP | C12381_F_COMPRESSED_IN))) return C12381_E_ARG;
    if (n == 0) return 0;
    staged s;
    if ((rc = stage_in(c, s, pts, ((flags & C12381_F_COMPRESSED_IN) ? 97 : 192) * n, sc, 32 * n, (size_t)fmt * n))) return rc;
    if ((rc = c12381_g2_mul_batch_flags_dev(c, n, s.in0, s.in1, s.out, fmt, flags))) return rc;
    if ((rc = stage_out(c, s, out, (size_t)fmt * n))) return rc;
    return read_flag(c);
}
int c12381_g2_mul_batch(c12381_ctx* c, size_t n, const uint8_t* pts, const uint8_t* sc, uint8_t* out, int fmt) {
    return c12381_g2_mul_batch_flags(c, n, pts, sc, out, fmt, 0u);
}
// Π q_i^{x_i} in G2 (scalars == NULL: the plain product of the points, g2_point.hpp:225-236).  The reference evaluates it as n
// multiply(point2&, big) calls and a chain of add(point2&, point2&); here: the batched scalar multiplication into the
// projective workspace, then a tree sum (two levels), one affine conversion.  Only the final point is canonical.
int c12381_g2_msm_dev(c12381_ctx* c, size_t n, const uint8_t* pts, const uint8_t* sc, uint8_t* out, int fmt) {
    int rc = bind(c); if (rc) return rc;
    if (!out || (n && !pts) || (fmt != 97 && fmt != 192)) return C12381_E_ARG;
    if (n == 0) { HIPCK(c, hipMemsetAsync(out, 0, fmt, c->stream)); return 0; }
    const size_t stride = round_up(n, 64);
    if ((rc = ensure(c, c12381_ctx::WS_PROJ, (size_t)6 * NL * stride * 4))) return rc;
    if (sc) {
        if ((rc = g2_mul_dev_strided(c, n, pts, 192, sc, out, fmt, nullptr, true))) return rc;
    } else {
        hipLaunchKernelGGL(g2_lift_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, c->stream, n, pts, (int32_t*)c->ws[c12381_ctx::WS_PROJ], stride, c->d_flag);
        HIPCK(c, hipGetLastError());
    }
    const int32_t* cur = (const int32_t*)c->ws[c12381_ctx::WS_PROJ];
    size_t cur_n = n, cur_stride = stride;
    int slot = c12381_ctx::WS_RED0;
    while (cur_n > 1) {
        const size_t m = cur_n > 4096 ? round_up(cur_n / 32, 64) : (cur_n > 64 ? 64 : 1);
        const size_t m_stride = round_up(m, 64);
        if ((rc = ensure(c, slot, (size_t)6 * NL * m_stride * 4))) return rc;
        hipLaunchKernelGGL(g2_reduce_kernel, dim3(grid_for(m)), dim3(BLOCK), 0, c->stream, cur_n, cur, cur_stride, m, (int32_t*)c->ws[slot], m_stride);
        HIPCK(c, hipGetLastError());
        cur = (const int32_t*)c->ws[slot]; cur_n = m; cur_stride = m_stride;
        slot = slot == c12381_ctx::WS_RED0 ? c12381_ctx::WS_RED1 : c12381_ctx::WS_RED0;
    }
    if ((rc = ensure(c, c12381_ctx::WS_PREF, (size_t)2 * NL * 64 * 4))) return rc;
    hipLaunchKernelGGL(g2_finish_kernel, dim3(1), dim3(BLOCK), 0, c->stream, (size_t)1, cur, cur_stride, (int32_t*)c->ws[c12381_ctx::WS_PREF], out, fmt, (size_t)1);
    HIPCK(c, hipGetLastError());
    return 0;
}
int c12381_g2_msm(c12381_ctx* c, size_t n, const uint8_t* pts, const uint8_t* sc, uint8_t* out, int fmt) {
    int rc = bind(c); if (rc) return rc;
    if (!out || (n && !pts) || (fmt != 97 && fmt != 192)) return C12381_E_ARG;
    staged s;
    if ((rc = stage_in(c, s, pts, 192 * n, sc, sc ? 32 * n : 0, (size_t)fmt))) return rc;
    if ((rc = c12381_g2_msm_dev(c, n, s.in0, sc ? s.in1 : nullptr, s.out, fmt))) return rc;
    if ((rc = stage_out(c, s, out, (size_t)fmt))) return rc;
    return read_flag(c);
}
int c12381_g2_add_batch(c12381_ctx* c, size_t n, const uint8_t* a, const uint8_t* b, uint8_t* out, int fmt) {
    int rc = bind(c); if (rc) return rc;
    if (!a || !b || !out || (fmt != 97 && fmt != 192)) return C12381_E_ARG;
    if (n == 0) return 0;
    staged s;
    if ((rc = stage_in(c, s, a, 192 * n, b, 192 * n, (size_t)fmt * n))) return rc;
    hipLaunchKernelGGL(g2_add_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, c->stream, n, s.in0, (size_t)192, s.in1, s.out, fmt, c->d_flag, (const int32_t*)nullptr);
    HIPCK(c, hipGetLastError());
    if ((rc = stage_out(c, s, out, (size_t)fmt * n))) return rc;
    return read_flag(c);
}

// ---------------------------------------------------------------- pairing
// C12381_PAIR_LANES=1 selects the one-lane-per-pairing kernels (kept for A/B measurements); default is 3.
#ifdef C12381_EXPERIMENTS
static int pair_lanes() {
    static const int v = [] { const char* e = tuning_env("C12381_PAIR_LANES"); return (e && e[0] == '1') ? 1 : 3; }();
    return v;
}
#else
static constexpr int pair_lanes() { return 3; }
#endif
static unsigned grid_tri(size_t n) {
    const size_t waves = (n + TRI_PER_WAVE - 1) / TRI_PER_WAVE;
    return (unsigned)((waves * 64 + BLOCK - 1) / BLOCK);
}
// Work-queue variant (k_pair3.hip): used when the batch is more than one machine-filling round of wavefronts, where the
// plain grid would end in a mostly idle round.  C12381_PAIR_QUEUE=0 / 1 forces it off / on (A/B measurements, tests).
constexpr size_t PAIR_QUEUE_WAVES = 2048;                  // resident wavefronts at 2 per SIMD
static int pair_queue_mode() {
    static const int v = [] { const char* e = tuning_env("C12381_PAIR_QUEUE"); return e ? (e[0] == '0' ? 0 : 1) : -1; }();
    return v;
}
// bound of the hand-over spin in the queue kernels (k_pair3.hip queue_wait): 2^20 sleeps of 4096 cycles, about two
// seconds — three orders of magnitude beyond a task.  C12381_PAIR_SPIN_LIMIT overrides it; a negative value makes every
// wait fail (tests of the poison path).
static int pair_spin_limit() {
    static const int v = [] { const char* e = tuning_env("C12381_PAIR_SPIN_LIMIT"); return e ? std::atoi(e) : (1 << 20); }();
    return v;
}
// Diagnostic: C12381_PAIR_STAMPS=<file> makes every task of pair3_queue_kernel record its claim / start / end times (s_memtime)
// into a device buffer that c12381_sync() writes to the file — per-phase durations and hand-over waits (tools/queue_phase_times.py).
static const char* pair_stamps_path() {
    static const char* p = tuning_env("C12381_PAIR_STAMPS");
    return p;
}
static unsigned long long* pair_stamps(c12381_ctx* c, size_t n) {
    if (!pair_stamps_path()) return nullptr;
    const size_t tasks = (n + TRI_PER_WAVE - 1) / TRI_PER_WAVE * 16;      // ten per group (room for up to 14), one more for its whole-group stamp
    if (c->stamps_tasks < tasks) {
        (void)hipStreamSynchronize(c->stream);              // a kernel of this context may still be writing the old buffer
        if (c->stamps) (void)hipFree(c->stamps);
        if (hipMalloc((void**)&c->stamps, tasks * 32 + c12381_ctx::STAMP_WAVES * 96) != hipSuccess) { c->stamps = nullptr; c->stamps_tasks = 0; return nullptr; }
        c->stamps_tasks = tasks;
    }
    (void)hipMemsetAsync(c->stamps, 0, c->stamps_tasks * 32 + c12381_ctx::STAMP_WAVES * 96, c->stream);
    return c->stamps;
}
// the per-wavefront region behind the per-task stamps (null when the diagnostic is off)
static unsigned long long* pair_wave_stats(c12381_ctx* c, size_t n) {
    unsigned long long* s = pair_stamps(c, n);
    return s ? s + c->stamps_tasks * 4 : nullptr;
}
static void pair_stamps_dump(c12381_ctx* c) {
    if (!pair_stamps_path() || !c->stamps) return;
    std::vector<unsigned long long> h(c->stamps_tasks * 4 + c12381_ctx::STAMP_WAVES * 12);
    if (hipMemcpy(h.data(), c->stamps, h.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return;
    if (FILE* f = std::fopen(pair_stamps_path(), "wb")) { std::fwrite(h.data(), 8, h.size(), f); std::fclose(f); }
}
static bool pair_use_queue(size_t n) {
    const int m = pair_queue_mode();
    if (m >= 0) return m == 1;
    return (n + TRI_PER_WAVE - 1) / TRI_PER_WAVE > PAIR_QUEUE_WAVES;
}
// the kernels' rule for how many groups bypass the queue (k_pair3.hip queue_direct_groups), mirrored for the slab size
static size_t queue_direct_groups_host(size_t ngroups, size_t nwaves) {
    if (ngroups <= nwaves) return 0;
    size_t queued = ngroups / 3;
    if (queued < nwaves / 2) queued = nwaves / 2;
    if (queued > 2 * nwaves) queued = 2 * nwaves;
    if (g_queue_groups_host > 0) queued = (size_t)g_queue_groups_host < ngroups ? (size_t)g_queue_groups_host : ngroups;
    return ngroups - queued;
}
// state slab: [flags: one word per group][task counter][whole-group counter][pad to 256 B][one block per QUEUED group] — whole
// groups keep their state in registers and the LDS slot; 2^18 BBS+ verifications: 4096 blocks (344 MB) instead of 12484 (1.0 GB).
// tagged (every kernel but the GT power): blocks of PAIR_QUEUE_STATE_BYTES in 8-byte tagged words, `epoch` = this launch's tag base.  The slab
// then only ever holds tagged words or zeros (zeroed when it is allocated and when the 28-bit epoch wraps), so a word of an earlier launch —
// at whatever offset that launch's group count put it — can never carry the tag of this one.  The GT power keeps the fenced 16-byte rows in a
// slab of its own.
static int pair_queue_setup(c12381_ctx* c, size_t n, uint4*& state, unsigned int*& flags, unsigned int*& counter, unsigned& blocks, unsigned int* epoch = nullptr) {
    const size_t groups = (n + TRI_PER_WAVE - 1) / TRI_PER_WAVE;
    const size_t head = round_up((groups + 2) * 4, 256);          // flags | task counter | whole-group counter
    const size_t waves = groups < PAIR_QUEUE_WAVES ? groups : PAIR_QUEUE_WAVES;
    blocks = (unsigned)((waves * 64 + BLOCK - 1) / BLOCK);
    const size_t nwaves = (size_t)blocks * (BLOCK / 64);
    const size_t nq = groups - queue_direct_groups_host(groups, nwaves);
    const bool tagged = epoch != nullptr;
    const int slot = tagged ? c12381_ctx::WS_PAIR_ST : c12381_ctx::WS_POW_ST;
    const size_t bytes = head + nq * (tagged ? PAIR_QUEUE_STATE_BYTES : (size_t)PAIR_QUEUE_STATE_ROWS * 1024);
    int rc;
    bool fresh = c->ws_bytes[slot] < bytes;
    if ((rc = ensure(c, slot, bytes))) return rc;
    if (tagged) {
        c->queue_epoch = (c->queue_epoch + 1u) & 0x0fffffffu;
        if (c->queue_epoch == 0) { c->queue_epoch = 1; fresh = true; }
        if (fresh) HIPCK(c, hipMemsetAsync(c->ws[slot], 0, c->ws_bytes[slot], c->stream));
        *epoch = c->queue_epoch;
    }
    uint8_t* base = (uint8_t*)c->ws[slot];
    flags = (unsigned int*)base;
    counter = flags + groups;
    state = (uint4*)(base + head);
    HIPCK(c, hipMemsetAsync(base, 0, (groups + 2) * 4, c->stream));
    return 0;
}
static int launch_pair(c12381_ctx* c, size_t n, const uint8_t* g1, const uint8_t* g2, uint8_t* gt) {
#ifdef C12381_EXPERIMENTS
    if (pair_lanes() == 1) { hipLaunchKernelGGL(pair_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, c->stream, n, g1, g2, gt, c->d_flag); HIPCK(c, hipGetLastError()); return 0; }
#endif
    if (pair_use_queue(n)) {
        uint4* st; unsigned int *fl, *ct, ep; unsigned blocks; int rc;
        if ((rc = pair_queue_setup(c, n, st, fl, ct, blocks, &ep))) return rc;
        unsigned long long* const stp = pair_stamps(c, n);
        hipLaunchKernelGGL(pair3_queue_kernel, dim3(blocks), dim3(BLOCK), 0, c->stream, n, g1, g2, gt, c->d_flag, st, fl, ct, pair_spin_limit(), ep, stp, stp ? stp + c->stamps_tasks * 4 : nullptr);
    } else hipLaunchKernelGGL(pair3_kernel, dim3(grid_tri(n)), dim3(BLOCK), 0, c->stream, n, g1, g2, gt, c->d_flag);
    HIPCK(c, hipGetLastError());
    return 0;
}
static int launch_pair_eq(c12381_ctx* c, size_t n, const uint8_t* a1, const uint8_t* a2, const uint8_t* b1, const uint8_t* b2, size_t b2_stride, uint8_t* ok,
                          const int32_t* skip_if = nullptr) {
#ifdef C12381_EXPERIMENTS
    if (pair_lanes() == 1) { hipLaunchKernelGGL(pair_eq_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, c->stream, n, a1, a2, b1, b2, b2_stride, ok, c->d_flag); HIPCK(c, hipGetLastError()); return 0; }
#endif
    if (pair_use_queue(n)) {
        uint4* st; unsigned int *fl, *ct, ep; unsigned blocks; int rc;
        if ((rc = pair_queue_setup(c, n, st, fl, ct, blocks, &ep))) return rc;
        hipLaunchKernelGGL(pair3_eq_queue_kernel, dim3(blocks), dim3(BLOCK), 0, c->stream, n, a1, a2, b1, b2, b2_stride, ok, c->d_flag, st, fl, ct, skip_if, pair_spin_limit(), ep);
    } else hipLaunchKernelGGL(pair3_eq_kernel, dim3(grid_tri(n)), dim3(BLOCK), 0, c->stream, n, a1, a2, b1, b2, b2_stride, ok, c->d_flag, skip_if);
    HIPCK(c, hipGetLastError());
    return 0;
}
int c12381_pair_batch_dev(c12381_ctx* c, size_t n, const uint8_t* g1, const uint8_t* g2, uint8_t* gt) {
    int rc = bind(c); if (rc) return rc;
    if (!g1 || !g2 || !gt) return C12381_E_ARG;
    if (n == 0) return 0;
    if (pair_lanes() != 1 && pair_use_queue(n)) {            // workspace and its reset stay outside the timed bracket
        uint4* st; unsigned int *fl, *ct, ep; unsigned blocks;
        if ((rc = pair_queue_setup(c, n, st, fl, ct, blocks, &ep))) return rc;
        timed tm(c, 3);
        unsigned long long* const stp = pair_stamps(c, n);
        hipLaunchKernelGGL(pair3_queue_kernel, dim3(blocks), dim3(BLOCK), 0, c->stream, n, g1, g2, gt, c->d_flag, st, fl, ct, pair_spin_limit(), ep, stp, stp ? stp + c->stamps_tasks * 4 : nullptr);
        HIPCK(c, hipGetLastError());
        return 0;
    }
    timed tm(c, 3);
    return launch_pair(c, n, g1, g2, gt);
}
// C12381_F_COMPRESSED_IN: g1 = n x 49, g2 = n x 97 bytes.  The pairing kernels read their inputs once per queue task (up to five times),
// so the decoding runs as its own two kernels into a workspace (288 B per pairing, against ~280 ns of arithmetic); a rejected
// encoding becomes an off-curve record there and surfaces exactly like an invalid 96 / 192-byte input: 0xff lane, C12381_E_POINT.
int c12381_pair_batch_flags_dev(c12381_ctx* c, size_t n, const uint8_t* g1, const uint8_t* g2, uint8_t* gt, unsigned flags) {
    int rc = bind(c); if (rc) return rc;
    if (!g1 || !g2 || !gt || (flags & ~(unsigned)C12381_F_COMPRESSED_IN)) return C12381_E_ARG;
    if (n == 0) return 0;
    if (!(flags & C12381_F_COMPRESSED_IN)) return c12381_pair_batch_dev(c, n, g1, g2, gt);
    if ((rc = ensure(c, c12381_ctx::WS_DEC1, 96 * n))) return rc;
    if ((rc = ensure(c, c12381_ctx::WS_DEC2, 192 * n))) return rc;
    uint8_t *d1 = (uint8_t*)c->ws[c12381_ctx::WS_DEC1], *d2 = (uint8_t*)c->ws[c12381_ctx::WS_DEC2];
    hipLaunchKernelGGL(g1_decompress_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, c->stream, n, g1, d1, (uint8_t*)nullptr, 1);
    hipLaunchKernelGGL(g2_decompress_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, c->stream, n, g2, d2, (uint8_t*)nullptr, 1);
    HIPCK(c, hipGetLastError());
    return c12381_pair_batch_dev(c, n, d1, d2, gt);
}
int c12381_pair_batch_flags(c12381_ctx* c, size_t n, const uint8_t* g1, const uint8_t* g2, uint8_t* gt, unsigned flags) {
    int rc = bind(c); if (rc) return rc;
    if (!g1 || !g2 || !gt || (flags & ~(unsigned)C12381_F_COMPRESSED_IN)) return C12381_E_ARG;
    if (n == 0) return 0;
    const bool comp = (flags & C12381_F_COMPRESSED_IN) != 0;
    staged s;
    if ((rc = stage_in(c, s, g1, (comp ? 49 : 96) * n, g2, (comp ? 97 : 192) * n, 576 * n))) return rc;
    if ((rc = c12381_pair_batch_flags_dev(c, n, s.in0, s.in1, s.out, flags))) return rc;
    if ((rc = stage_out(c, s, gt, 576 * n))) return rc;
    return read_flag(c);
}
int c12381_pair_batch(c12381_ctx* c, size_t n, const uint8_t* g1, const uint8_t* g2, uint8_t* gt) { return c12381_pair_batch_flags(c, n, g1, g2, gt, 0u); }
// Product of k pairings per element with shared squarings (pair3_prod_kernel)
int c12381_pair_product_batch_dev(c12381_ctx* c, size_t n, int k, const uint8_t* g1s, const uint8_t* g2s, uint8_t* gt, unsigned flags) {
    int rc = bind(c); if (rc) return rc;
    if (!g1s || !g2s || !gt || k < 1 || k > MAX_PROD || (flags & ~(unsigned)C12381_F_MILLER_ONLY)) return C12381_E_ARG;
    if (n == 0) return 0;
    timed tm(c, 3);
    hipLaunchKernelGGL(pair3_prod_kernel, dim3(grid_tri(n)), dim3(BLOCK), 0, c->stream, n, k, g1s, g2s, gt, c->d_flag, (flags & C12381_F_MILLER_ONLY) ? 1 : 0);
    HIPCK(c, hipGetLastError());
    return 0;
}
int c12381_pair_product_batch(c12381_ctx* c, size_t n, int k, const uint8_t* g1s, const uint8_t* g2s, uint8_t* gt, unsigned flags) {
    int rc = bind(c); if (rc) return rc;
    if (!g1s || !g2s || !gt || k < 1 || k > MAX_PROD || (flags & ~(unsigned)C12381_F_MILLER_ONLY)) return C12381_E_ARG;
    if (n == 0) return 0;
    staged s;
    if ((rc = stage_in(c, s, g1s, 96 * n * (size_t)k, g2s, 192 * n * (size_t)k, 576 * n))) return rc;
    if ((rc = c12381_pair_product_batch_dev(c, n, k, s.in0, s.in1, s.out, flags))) return rc;
    if ((rc = stage_out(c, s, gt, 576 * n))) return rc;
    return read_flag(c);
}
static int lines_table(c12381_ctx* c, int slot, const uint8_t* d_q192, int need_g2);
// gt[i] = e(P_i, Q) with ONE G2 argument for the batch: the 69 line-coefficient triples of Q are computed once (and kept
// until Q changes), every element then runs the table-driven Miller loop.  Same field elements as the running-point loop,
// so the GT bytes equal c12381_pair_batch on n copies of Q for every Q, infinity included.
int c12381_pair_fixed_g2_batch_dev(c12381_ctx* c, size_t n, const uint8_t* g1, const uint8_t* g2_192, uint8_t* gt) {
    int rc = bind(c); if (rc) return rc;
    if (!g1 || !g2_192 || !gt) return C12381_E_ARG;
    if (n == 0) return 0;
    if ((rc = lines_table(c, c12381_ctx::WS_FQ_P, g2_192, 0))) return rc;
    uint4* st; unsigned int *fl, *ct; unsigned blocks;
    unsigned int ep;
    if ((rc = pair_queue_setup(c, n, st, fl, ct, blocks, &ep))) return rc;
    timed tm(c, 3);
    hipLaunchKernelGGL(pair3_fixed_queue_kernel, dim3(blocks), dim3(BLOCK), 0, c->stream, n, g1, (const int32_t*)c->ws[c12381_ctx::WS_FQ_P], gt, c->d_flag,
                       st, fl, ct, pair_spin_limit(), ep);
    HIPCK(c, hipGetLastError());
    return 0;
}
int c12381_pair_fixed_g2_batch(c12381_ctx* c, size_t n, const uint8_t* g1, const uint8_t* g2_192, uint8_t* gt) {
    int rc = bind(c); if (rc) return rc;
    if (!g1 || !g2_192 || !gt) return C12381_E_ARG;
    if (n == 0) return 0;
    staged s;
    if ((rc = stage_in(c, s, g1, 96 * n, g2_192, 192, 576 * n))) return rc;
    if ((rc = c12381_pair_fixed_g2_batch_dev(c, n, s.in0, s.in1, s.out))) return rc;
    if ((rc = stage_out(c, s, gt, 576 * n))) return rc;
    return read_flag(c);
}
int c12381_pair_eq_batch_dev(c12381_ctx* c, size_t n, const uint8_t* a1, const uint8_t* a2, const uint8_t* b1, const uint8_t* b2, uint8_t* ok) {
    int rc = bind(c); if (rc) return rc;
    if (!a1 || !a2 || !b1 || !b2 || !ok) return C12381_E_ARG;
    if (n == 0) return 0;
    timed tm(c, 4);
    return launch_pair_eq(c, n, a1, a2, b1, b2, (size_t)192, ok);
}
int c12381_pair_eq_batch(c12381_ctx* c, size_t n, const uint8_t* a1, const uint8_t* a2, const uint8_t* b1, const uint8_t* b2, uint8_t* ok) {
    int rc = bind(c); if (rc) return rc;
    if (!a1 || !a2 || !b1 || !b2 || !ok) return C12381_E_ARG;
    if (n == 0) return 0;
    int r2;
    if ((r2 = ensure(c, c12381_ctx::WS_IN0, round_up(96 * n, 256)))) return r2;
    if ((r2 = ensure(c, c12381_ctx::WS_IN1, round_up(192 * n, 256)))) return r2;
    if ((r2 = ensure(c, c12381_ctx::WS_RED0, round_up(96 * n, 256)))) return r2;
    if ((r2 = ensure(c, c12381_ctx::WS_RED1, round_up(192 * n, 256)))) return r2;
    if ((r2 = ensure(c, c12381_ctx::WS_OUT, round_up(n, 256)))) return r2;
    uint8_t* d_a1 = (uint8_t*)c->ws[c12381_ctx::WS_IN0]; uint8_t* d_a2 = (uint8_t*)c->ws[c12381_ctx::WS_IN1];
    uint8_t* d_b1 = (uint8_t*)c->ws[c12381_ctx::WS_RED0]; uint8_t* d_b2 = (uint8_t*)c->ws[c12381_ctx::WS_RED1];
    uint8_t* d_ok = (uint8_t*)c->ws[c12381_ctx::WS_OUT];
    HIPCK(c, hipMemcpyAsync(d_a1, a1, 96 * n, hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipMemcpyAsync(d_a2, a2, 192 * n, hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipMemcpyAsync(d_b1, b1, 96 * n, hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipMemcpyAsync(d_b2, b2, 192 * n, hipMemcpyHostToDevice, c->stream));
    if ((rc = c12381_pair_eq_batch_dev(c, n, d_a1, d_a2, d_b1, d_b2, d_ok))) return rc;
    HIPCK(c, hipMemcpyAsync(ok, d_ok, n, hipMemcpyDeviceToHost, c->stream));
    return read_flag(c);
}

// ---------------------------------------------------------------- decode / split pairing / GT
int c12381_g1_decompress_batch_dev(c12381_ctx* c, size_t n, const uint8_t* in49, uint8_t* out96, uint8_t* status) {
    int rc = bind(c); if (rc) return rc;
    if (!in49 || !out96 || !status) return C12381_E_ARG;
    if (n == 0) return 0;
    hipLaunchKernelGGL(g1_decompress_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, c->stream, n, in49, out96, status, 0);
    HIPCK(c, hipGetLastError());
    return 0;
}
int c12381_g2_decompress_batch_dev(c12381_ctx* c, size_t n, const uint8_t* in97, uint8_t* out192, uint8_t* status) {
    int rc = bind(c); if (rc) return rc;
    if (!in97 || !out192 || !status) return C12381_E_ARG;
    if (n == 0) return 0;
    hipLaunchKernelGGL(g2_decompress_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, c->stream, n, in97, out192, status, 0);
    HIPCK(c, hipGetLastError());
    return 0;
}
int c12381_g1_decompress_batch(c12381_ctx* c, size_t n, const uint8_t* in49, uint8_t* out96, uint8_t* status) {
    int rc = bind(c); if (rc) return rc;
    if (!in49 || !out96 || !status) return C12381_E_ARG;
    if (n == 0) return 0;
    staged s;
    if ((rc = stage_in(c, s, in49, 49 * n, nullptr, n, 96 * n))) return rc;
    hipLaunchKernelGGL(g1_decompress_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, c->stream, n, s.in0, s.out, s.in1, 0);
    HIPCK(c, hipGetLastError());
    if ((rc = stage_out(c, s, out96, 96 * n))) return rc;
    HIPCK(c, hipMemcpyAsync(status, s.in1, n, hipMemcpyDeviceToHost, c->stream));
    return read_flag(c);
}
// ---------------------------------------------------------------- hash-to-G1, Zp helpers
static int g1_map_common(c12381_ctx* c, size_t n, const uint8_t* d_in, int mode, uint8_t* d_out, int fmt) {
    const size_t stride = round_up(n, 64);
    int rc;
    if ((rc = ensure(c, c12381_ctx::WS_PROJ, (size_t)3 * NL * stride * 4))) return rc;
    hipLaunchKernelGGL(g1_from_hash_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, c->stream, n, d_in, mode, (int32_t*)c->ws[c12381_ctx::WS_PROJ], stride,
                       c->d_flag);
    HIPCK(c, hipGetLastError());
    return g1_finish(c, n, (const int32_t*)c->ws[c12381_ctx::WS_PROJ], stride, d_out, fmt);
}
int c12381_g1_from_hash_batch_dev(c12381_ctx* c, size_t n, const uint8_t* digests, uint8_t* out, int fmt) {
    int rc = bind(c); if (rc) return rc;
    if (!digests || !out || (fmt != 49 && fmt != 96)) return C12381_E_ARG;
    if (n == 0) return 0;
    return g1_map_common(c, n, digests, 0, out, fmt);
}
int c12381_g1_from_hash_batch(c12381_ctx* c, size_t n, const uint8_t* digests, uint8_t* out, int fmt) {
    int rc = bind(c); if (rc) return rc;
    if (!digests || !out || (fmt != 49 && fmt != 96)) return C12381_E_ARG;
    if (n == 0) return 0;
    staged s;
    if ((rc = stage_in(c, s, digests, 64 * n, nullptr, 0, (size_t)fmt * n))) return rc;
    if ((rc = g1_map_common(c, n, s.in0, 0, s.out, fmt))) return rc;
    if ((rc = stage_out(c, s, out, (size_t)fmt * n))) return rc;
    return read_flag(c);
}
int c12381_g1_map_to_point_batch(c12381_ctx* c, size_t n, const uint8_t* u48, uint8_t* out96) {
    int rc = bind(c); if (rc) return rc;
    if (!u48 || !out96) return C12381_E_ARG;
    if (n == 0) return 0;
    staged s;
    if ((rc = stage_in(c, s, u48, 48 * n, nullptr, 0, 96 * n))) return rc;
    if ((rc = g1_map_common(c, n, s.in0, 1, s.out, 96))) return rc;
    if ((rc = stage_out(c, s, out96, 96 * n))) return rc;
    return read_flag(c);
}
int c12381_g1_clear_cofactor_batch(c12381_ctx* c, size_t n, const uint8_t* in96, uint8_t* out96) {
    int rc = bind(c); if (rc) return rc;
    if (!in96 || !out96) return C12381_E_ARG;
    if (n == 0) return 0;
    staged s;
    if ((rc = stage_in(c, s, in96, 96 * n, nullptr, 0, 96 * n))) return rc;
    if ((rc = g1_map_common(c, n, s.in0, 2, s.out, 96))) return rc;
    if ((rc = stage_out(c, s, out96, 96 * n))) return rc;
    return read_flag(c);
}
// out[i] = 1 / (x[i] + gamma) (gamma may be null), simultaneous inversion in runs of ZP_INV_RUN (k_hash_zp.hip)
static int zp_batch_inverse(c12381_ctx* c, size_t n, const uint8_t* x, const uint8_t* gamma, uint8_t* out) {
    int rc;
    if ((rc = ensure(c, c12381_ctx::WS_PREF, 32 * n))) return rc;
    const size_t T = (n + ZP_INV_RUN - 1) / ZP_INV_RUN;
    hipLaunchKernelGGL(zp_batch_inv_kernel, dim3(grid_for(T)), dim3(BLOCK), 0, c->stream, n, T, x, gamma, out, (uint32_t*)c->ws[c12381_ctx::WS_PREF]);
    HIPCK(c, hipGetLastError());
    return 0;
}
int c12381_zp_op_batch_dev(c12381_ctx* c, int op, size_t n, const uint8_t* a, const uint8_t* b, uint8_t* out) {
    int rc = bind(c); if (rc) return rc;
    if (op < 0 || op > 4 || !a || !out || (op <= 2 && !b)) return C12381_E_ARG;
    if (n == 0) return 0;
    if (op == 4) return zp_batch_inverse(c, n, a, nullptr, out);
    hipLaunchKernelGGL(zp_op_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, c->stream, op, n, a, op <= 2 ? b : nullptr, out);
    HIPCK(c, hipGetLastError());
    return 0;
}
int c12381_zp_op_batch(c12381_ctx* c, int op, size_t n, const uint8_t* a, const uint8_t* b, uint8_t* out) {
    int rc = bind(c); if (rc) return rc;
    if (op < 0 || op > 4 || !a || !out || (op <= 2 && !b)) return C12381_E_ARG;
    if (n == 0) return 0;
    staged s;
    if ((rc = stage_in(c, s, a, 32 * n, op <= 2 ? b : nullptr, op <= 2 ? 32 * n : 0, 32 * n))) return rc;
    if ((rc = c12381_zp_op_batch_dev(c, op, n, s.in0, s.in1, s.out))) return rc;
    if ((rc = stage_out(c, s, out, 32 * n))) return rc;
    return read_flag(c);
}
int c12381_zp_from_hash_batch(c12381_ctx* c, size_t n, const uint8_t* digests, uint8_t* out) {
    int rc = bind(c); if (rc) return rc;
    if (!digests || !out) return C12381_E_ARG;
    if (n == 0) return 0;
    staged s;
    if ((rc = stage_in(c, s, digests, 64 * n, nullptr, 0, 32 * n))) return rc;
    hipLaunchKernelGGL(zp_from_hash_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, c->stream, n, s.in0, s.out);
    HIPCK(c, hipGetLastError());
    if ((rc = stage_out(c, s, out, 32 * n))) return rc;
    return read_flag(c);
}
// strided partial sums, 64 terms per lane and stage, ping-pong between two reduction slots
int c12381_zp_inner_product_dev(c12381_ctx* c, size_t n, const uint8_t* a, const uint8_t* b, uint8_t* out) {
    int rc = bind(c); if (rc) return rc;
    if (!out || (n && !a)) return C12381_E_ARG;
    if (n == 0) { HIPCK(c, hipMemsetAsync(out, 0, 32, c->stream)); return 0; }
    const uint8_t *cur_a = a, *cur_b = b;
    size_t cur_n = n;
    int slot = c12381_ctx::WS_RED0;
    for (;;) {
        const size_t T = (cur_n + 63) / 64;
        uint8_t* dst = out;
        if (T > 1) {
            if ((rc = ensure(c, slot, round_up(32 * T, 256)))) return rc;
            dst = (uint8_t*)c->ws[slot];
        }
        hipLaunchKernelGGL(zp_fold_kernel, dim3(grid_for(T)), dim3(BLOCK), 0, c->stream, cur_n, cur_a, cur_b, T, dst);
        HIPCK(c, hipGetLastError());
        if (T == 1) return 0;
        cur_a = dst; cur_b = nullptr; cur_n = T;
        slot = slot == c12381_ctx::WS_RED0 ? c12381_ctx::WS_RED1 : c12381_ctx::WS_RED0;
    }
}
int c12381_zp_inner_product(c12381_ctx* c, size_t n, const uint8_t* a, const uint8_t* b, uint8_t* out) {
    int rc = bind(c); if (rc) return rc;
    if (!out || (n && !a)) return C12381_E_ARG;
    if (n == 0) { std::memset(out, 0, 32); return 0; }
    staged s;
    if ((rc = stage_in(c, s, a, 32 * n, b, b ? 32 * n : 0, 32))) return rc;
    if ((rc = c12381_zp_inner_product_dev(c, n, s.in0, b ? s.in1 : nullptr, s.out))) return rc;
    if ((rc = stage_out(c, s, out, 32))) return rc;
    return read_flag(c);
}

int c12381_g2_decompress_batch(c12381_ctx* c, size_t n, const uint8_t* in97, uint8_t* out192, uint8_t* status) {
    int rc = bind(c); if (rc) return rc;
    if (!in97 || !out192 || !status) return C12381_E_ARG;
    if (n == 0) return 0;
    staged s;
    if ((rc = stage_in(c, s, in97, 97 * n, nullptr, n, 192 * n))) return rc;
    hipLaunchKernelGGL(g2_decompress_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, c->stream, n, s.in0, s.out, s.in1, 0);
    HIPCK(c, hipGetLastError());
    if ((rc = stage_out(c, s, out192, 192 * n))) return rc;
    HIPCK(c, hipMemcpyAsync(status, s.in1, n, hipMemcpyDeviceToHost, c->stream));
    return read_flag(c);
}
static int launch_miller(c12381_ctx* c, size_t n, const uint8_t* g1, const uint8_t* g2, uint8_t* out) {
#ifdef C12381_EXPERIMENTS
    if (pair_lanes() == 1) { hipLaunchKernelGGL(miller_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, c->stream, n, g1, g2, out, c->d_flag); HIPCK(c, hipGetLastError()); return 0; }
#endif
    if (pair_use_queue(n)) {              // more than one machine round of wavefront tasks: quarter-loop tasks from the work queue
        uint4* st; unsigned int *fl, *ct, ep; unsigned blocks; int rc;
        if ((rc = pair_queue_setup(c, n, st, fl, ct, blocks, &ep))) return rc;
        hipLaunchKernelGGL(miller3_queue_kernel, dim3(blocks), dim3(BLOCK), 0, c->stream, n, g1, g2, out, c->d_flag, st, fl, ct, pair_spin_limit(), ep, pair_wave_stats(c, n));
    } else hipLaunchKernelGGL(miller3_kernel, dim3(grid_tri(n)), dim3(BLOCK), 0, c->stream, n, g1, g2, out, c->d_flag);
    HIPCK(c, hipGetLastError());
    return 0;
}
static int launch_gt_op(c12381_ctx* c, int op, size_t n, const uint8_t* a, const uint8_t* b, uint8_t* out) {
#ifdef C12381_EXPERIMENTS
    if (pair_lanes() == 1) { hipLaunchKernelGGL(gt_op_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, c->stream, op, n, a, b, out); HIPCK(c, hipGetLastError()); return 0; }
#endif
    if (op == 3 && pair_use_queue(n)) {   // final exponentiations alone, more than one machine round: its six steps as queue tasks
        uint4* st; unsigned int *fl, *ct, ep; unsigned blocks; int rc;
        if ((rc = pair_queue_setup(c, n, st, fl, ct, blocks, &ep))) return rc;
        hipLaunchKernelGGL(fexp3_queue_kernel, dim3(blocks), dim3(BLOCK), 0, c->stream, n, a, out, c->d_flag, st, fl, ct, pair_spin_limit(), ep, pair_wave_stats(c, n));
    } else if (op == 2 && pair_use_queue(n)) {
        // the power, more than one machine round: five tasks per queued group (k_pair3.hip gt3_pow_queue_kernel); one table per wavefront of the
        // grid and one per queued group (at most 2048 + 4096 tables of 224 KB)
        uint4* st; unsigned int *fl, *ct; unsigned blocks; int rc;
        if ((rc = pair_queue_setup(c, n, st, fl, ct, blocks))) return rc;
        const size_t groups = (n + TRI_PER_WAVE - 1) / TRI_PER_WAVE, nwaves = (size_t)blocks * (BLOCK / 64);
        const size_t tables = nwaves + (groups - queue_direct_groups_host(groups, nwaves));
        // the rule queues at most 2 x the grid: 6144 tables = 1.4 GB, held until c12381_trim / c12381_destroy; a tuning override beyond that is refused
        if (tables > 3 * PAIR_QUEUE_WAVES) { std::snprintf(c->err, sizeof c->err, "GT power: %zu tables exceed the workspace budget (queued-groups override too large)", tables); return C12381_E_ARG; }
        if ((rc = ensure(c, c12381_ctx::WS_GT_POW, tables * GT_POW_TAB_BYTES_PER_WAVE))) return rc;
        hipLaunchKernelGGL(gt3_pow_queue_kernel, dim3(blocks), dim3(BLOCK), 0, c->stream, n, a, b, out, c->d_flag, (uint4*)c->ws[c12381_ctx::WS_GT_POW], st, fl, ct,
                           pair_spin_limit());
    } else if (op == 2) {
        // the power in one plain launch: at most PAIR_QUEUE_WAVES wavefronts get here (longer batches took the queue above), each with its table
        // of x^0 .. x^15 behind it (224 KB per wavefront).  Only an experiments run with the queue forced off can be longer: it runs the
        // reference's digit sequence without tables.
        const size_t waves = (n + TRI_PER_WAVE - 1) / TRI_PER_WAVE;
        uint4* tab = nullptr;
        if (waves <= PAIR_QUEUE_WAVES) {
            int rc;
            if ((rc = ensure(c, c12381_ctx::WS_GT_POW, waves * GT_POW_TAB_BYTES_PER_WAVE))) return rc;
            tab = (uint4*)c->ws[c12381_ctx::WS_GT_POW];
        }
        hipLaunchKernelGGL(gt3_op_kernel, dim3(grid_tri(n)), dim3(BLOCK), 0, c->stream, op, n, a, b, out, tab);
    } else hipLaunchKernelGGL(gt3_op_kernel, dim3(grid_tri(n)), dim3(BLOCK), 0, c->stream, op, n, a, b, out, (uint4*)nullptr);
    HIPCK(c, hipGetLastError());
    return 0;
}
static int launch_gt_is_unity(c12381_ctx* c, size_t n, const uint8_t* a, uint8_t* out) {
#ifdef C12381_EXPERIMENTS
    if (pair_lanes() == 1) { hipLaunchKernelGGL(gt_is_unity_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, c->stream, n, a, out); HIPCK(c, hipGetLastError()); return 0; }
#endif
    hipLaunchKernelGGL(gt3_is_unity_kernel, dim3(grid_tri(n)), dim3(BLOCK), 0, c->stream, n, a, out);
    HIPCK(c, hipGetLastError());
    return 0;
}
int c12381_miller_batch(c12381_ctx* c, size_t n, const uint8_t* g1, const uint8_t* g2, uint8_t* out576) {
    int rc = bind(c); if (rc) return rc;
    if (!g1 || !g2 || !out576) return C12381_E_ARG;
    if (n == 0) return 0;
    staged s;
    if ((rc = stage_in(c, s, g1, 96 * n, g2, 192 * n, 576 * n))) return rc;
    if ((rc = launch_miller(c, n, s.in0, s.in1, s.out))) return rc;
    if ((rc = stage_out(c, s, out576, 576 * n))) return rc;
    return read_flag(c);
}
int c12381_miller_batch_dev(c12381_ctx* c, size_t n, const uint8_t* g1, const uint8_t* g2, uint8_t* out576) {
    int rc = bind(c); if (rc) return rc;
    if (!g1 || !g2 || !out576) return C12381_E_ARG;
    if (n == 0) return 0;
    timed tm(c, 6);
    return launch_miller(c, n, g1, g2, out576);
}
int c12381_gt_op_batch(c12381_ctx* c, int op, size_t n, const uint8_t* a576, const uint8_t* b, uint8_t* out576) {
    int rc = bind(c); if (rc) return rc;
    if (op < 0 || op > 3 || !a576 || !out576 || ((op == 0 || op == 2) && !b)) return C12381_E_ARG;
    if (n == 0) return 0;
    staged s;
    const size_t bb = op == 0 ? 576 * n : (op == 2 ? 32 * n : 0);
    if ((rc = stage_in(c, s, a576, 576 * n, bb ? b : nullptr, bb, 576 * n))) return rc;
    if ((rc = launch_gt_op(c, op, n, s.in0, s.in1, s.out))) return rc;
    if ((rc = stage_out(c, s, out576, 576 * n))) return rc;
    return read_flag(c);
}
int c12381_gt_op_batch_dev(c12381_ctx* c, int op, size_t n, const uint8_t* a576, const uint8_t* b, uint8_t* out576) {
    int rc = bind(c); if (rc) return rc;
    if (op < 0 || op > 3 || !a576 || !out576 || ((op == 0 || op == 2) && !b)) return C12381_E_ARG;
    if (n == 0) return 0;
    timed tm(c, 7);
    return launch_gt_op(c, op, n, a576, b, out576);
}
int c12381_fexp_batch(c12381_ctx* c, size_t n, const uint8_t* in576, uint8_t* out576) { return c12381_gt_op_batch(c, 3, n, in576, nullptr, out576); }
int c12381_fexp_batch_dev(c12381_ctx* c, size_t n, const uint8_t* in576, uint8_t* out576) { return c12381_gt_op_batch_dev(c, 3, n, in576, nullptr, out576); }
int c12381_gt_is_unity_batch(c12381_ctx* c, size_t n, const uint8_t* a576, uint8_t* out) {
    int rc = bind(c); if (rc) return rc;
    if (!a576 || !out) return C12381_E_ARG;
    if (n == 0) return 0;
    staged s;
    if ((rc = stage_in(c, s, a576, 576 * n, nullptr, 0, n))) return rc;
    if ((rc = launch_gt_is_unity(c, n, s.in0, s.out))) return rc;
    if ((rc = stage_out(c, s, out, n))) return rc;
    return read_flag(c);
}
int c12381_gt_is_unity_batch_dev(c12381_ctx* c, size_t n, const uint8_t* a576, uint8_t* out) {
    int rc = bind(c); if (rc) return rc;
    if (!a576 || !out) return C12381_E_ARG;
    if (n == 0) return 0;
    return launch_gt_is_unity(c, n, a576, out);
}

// Fixed-base tables (fixed_base.hpp): make sure slot `slot` holds the table of the point at `d_base`; everything is
// queued on the stream (the "same base as last time?" comparison runs on the device), nothing waits for the host.
static int fixed_table(c12381_ctx* c, int slot, const uint8_t* d_base, bool is_g2) {
    const size_t entries = (size_t)(is_g2 ? FB_G2_WINDOWS : FB_G1_WINDOWS) * FB_ENTRIES;
    const size_t dwords = FB_HEADER_DWORDS + entries * (size_t)(is_g2 ? FB_G2_DWORDS : FB_G1_DWORDS);
    int rc;
    if (c->ws_bytes[slot] < dwords * 4) {
        if ((rc = ensure(c, slot, dwords * 4))) return rc;
        HIPCK(c, hipMemsetAsync(c->ws[slot], 0, FB_HEADER_DWORDS * 4, c->stream));      // no magic yet: first use is a miss
    }
    int32_t* buf = (int32_t*)c->ws[slot];
    hipLaunchKernelGGL(fixed_cache_check_kernel, dim3(1), dim3(64), 0, c->stream, d_base, is_g2 ? 192 : 96, buf);
    HIPCK(c, hipGetLastError());
    if (is_g2) hipLaunchKernelGGL(g2_fixed_table_kernel, dim3(grid_for(entries)), dim3(BLOCK), 0, c->stream, d_base, buf);
    else hipLaunchKernelGGL(g1_fixed_table_kernel, dim3(grid_for(entries)), dim3(BLOCK), 0, c->stream, d_base, buf);
    HIPCK(c, hipGetLastError());
    return 0;
}
// Coefficient table of a fixed G2 argument of the Miller loop (pairing3.hpp): same header / cache protocol as above.
static int lines_table(c12381_ctx* c, int slot, const uint8_t* d_q192, int need_g2) {
    const size_t dwords = FB_HEADER_DWORDS + (size_t)FQ_TABLE_DWORDS;
    int rc;
    if (c->ws_bytes[slot] < dwords * 4) {
        if ((rc = ensure(c, slot, dwords * 4))) return rc;
        HIPCK(c, hipMemsetAsync(c->ws[slot], 0, FB_HEADER_DWORDS * 4, c->stream));
    }
    int32_t* buf = (int32_t*)c->ws[slot];
    hipLaunchKernelGGL(fixed_cache_check_kernel, dim3(1), dim3(64), 0, c->stream, d_q192, 192, buf);
    HIPCK(c, hipGetLastError());
    static const int raw = [] { const char* e = tuning_env("C12381_FQ_RAW"); return (e && e[0] == '1') ? 4 : 0; }();
    hipLaunchKernelGGL(g2_lines_table_kernel, dim3(1), dim3(BLOCK), 0, c->stream, d_q192, buf, need_g2 | raw);
    HIPCK(c, hipGetLastError());
    return 0;
}
static bool fixed_base_enabled() {
    static const bool on = [] { const char* e = tuning_env("C12381_FIXED_BASE"); return !(e && e[0] == '0'); }();
    return on;
}

// ---------------------------------------------------------------- one base for the whole batch (g^x_i)
int c12381_g1_mul_fixed_batch_dev(c12381_ctx* c, size_t n, const uint8_t* base96, const uint8_t* sc, uint8_t* out, int fmt) {
    int rc = bind(c); if (rc) return rc;
    if (!base96 || !sc || !out || (fmt != 49 && fmt != 96)) return C12381_E_ARG;
    if (n == 0) return 0;
    const size_t stride = round_up(n, 64);
    if ((rc = ensure(c, c12381_ctx::WS_PROJ, (size_t)3 * NL * stride * 4))) return rc;
    const int32_t* skip = nullptr;
    if (fixed_base_enabled()) {
        if ((rc = fixed_table(c, c12381_ctx::WS_FB_G1_0, base96, false))) return rc;
        skip = (const int32_t*)c->ws[c12381_ctx::WS_FB_G1_0];
        hipLaunchKernelGGL(g1_fixed_eval_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, c->stream, n, skip, sc, (int32_t*)c->ws[c12381_ctx::WS_PROJ], stride,
                           (size_t)0);
        HIPCK(c, hipGetLastError());
    }
    if ((rc = g1_mul_to_proj(c, n, base96, sc, stride, 0, 0, skip))) return rc;
    return g1_finish(c, n, (const int32_t*)c->ws[c12381_ctx::WS_PROJ], stride, out, fmt);
}
int c12381_g1_mul_fixed_batch(c12381_ctx* c, size_t n, const uint8_t* base96, const uint8_t* sc, uint8_t* out, int fmt) {
    int rc = bind(c); if (rc) return rc;
    if (!base96 || !sc || !out || (fmt != 49 && fmt != 96)) return C12381_E_ARG;
    if (n == 0) return 0;
    staged s;
    if ((rc = stage_in(c, s, sc, 32 * n, base96, 96, (size_t)fmt * n))) return rc;
    if ((rc = c12381_g1_mul_fixed_batch_dev(c, n, s.in1, s.in0, s.out, fmt))) return rc;
    if ((rc = stage_out(c, s, out, (size_t)fmt * n))) return rc;
    return read_flag(c);
}
int c12381_g2_mul_fixed_batch_dev(c12381_ctx* c, size_t n, const uint8_t* base192, const uint8_t* sc, uint8_t* out, int fmt) {
    int rc = bind(c); if (rc) return rc;
    if (!base192 || !sc || !out || (fmt != 97 && fmt != 192)) return C12381_E_ARG;
    if (n == 0) return 0;
    const int32_t* skip = nullptr;
    const size_t stride = round_up(n, 64);
    if ((rc = ensure(c, c12381_ctx::WS_PROJ, (size_t)6 * NL * stride * 4))) return rc;
    if (fixed_base_enabled()) {
        if ((rc = fixed_table(c, c12381_ctx::WS_FB_G2, base192, true))) return rc;
        skip = (const int32_t*)c->ws[c12381_ctx::WS_FB_G2];
        hipLaunchKernelGGL(g2_fixed_eval_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, c->stream, n, skip, sc, (const uint8_t*)nullptr, out, fmt, c->d_flag,
                           (int32_t*)c->ws[c12381_ctx::WS_PROJ], stride);
        HIPCK(c, hipGetLastError());
    }
    if ((rc = g2_mul_dev_strided(c, n, base192, 0, sc, out, fmt, skip, true))) return rc;      // exactly one of the two kernels fills WS_PROJ
    return g2_finish(c, n, out, fmt);
}
int c12381_g2_mul_fixed_batch(c12381_ctx* c, size_t n, const uint8_t* base192, const uint8_t* sc, uint8_t* out, int fmt) {
    int rc = bind(c); if (rc) return rc;
    if (!base192 || !sc || !out || (fmt != 97 && fmt != 192)) return C12381_E_ARG;
    if (n == 0) return 0;
    staged s;
    if ((rc = stage_in(c, s, sc, 32 * n, base192, 192, (size_t)fmt * n))) return rc;
    if ((rc = c12381_g2_mul_fixed_batch_dev(c, n, s.in1, s.in0, s.out, fmt))) return rc;
    if ((rc = stage_out(c, s, out, (size_t)fmt * n))) return rc;
    return read_flag(c);
}

// B_j = g1 + r_j h0 + sum_i m_ij h_i for a batch of BBS+ signatures (bbs+.cpp:51, :72): (nmsg + 1) columns of n scalar
// multiplications with ONE base each — table-driven for subgroup bases, generic otherwise — summed per lane.  Result:
// projective SoA in WS_RED0 (`red`, stride `rstride`); `stride` is the stride of the column workspace WS_PROJ.
static int bbs_message_points(c12381_ctx* c, size_t n, size_t nmsg, const uint8_t* g1_96, const uint8_t* h0_96, const uint8_t* h_96, const uint8_t* r_32,
                              const uint8_t* m_32, bool fb, int32_t*& red, size_t& rstride, size_t& stride) {
    int rc;
    const size_t cols = nmsg + 1, total = cols * n;
    stride = round_up(total, 64);
    if ((rc = ensure(c, c12381_ctx::WS_PROJ, (size_t)3 * NL * stride * 4))) return rc;
    for (size_t col = 0; col < cols; ++col) {
        const uint8_t* base = col == 0 ? h0_96 : h_96 + 96 * (col - 1);
        const uint8_t* sc = col == 0 ? r_32 : m_32 + 32 * n * (col - 1);
        const int32_t* skip = nullptr;
        if (fb && col < 4) {                                   // table slots for h0 and the first three h_i
            const int slot = c12381_ctx::WS_FB_G1_0 + (int)col;
            if ((rc = fixed_table(c, slot, base, false))) return rc;
            skip = (const int32_t*)c->ws[slot];
            hipLaunchKernelGGL(g1_fixed_eval_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, c->stream, n, skip, sc, (int32_t*)c->ws[c12381_ctx::WS_PROJ],
                               stride, col * n);
            HIPCK(c, hipGetLastError());
        }
        if ((rc = g1_mul_to_proj(c, n, base, sc, stride, 0, col * n, skip))) return rc;
    }
    rstride = round_up(n, 64);
    if ((rc = ensure(c, c12381_ctx::WS_RED0, (size_t)3 * NL * rstride * 4))) return rc;
    red = (int32_t*)c->ws[c12381_ctx::WS_RED0];
    hipLaunchKernelGGL(g1_reduce_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, c->stream, total, (const int32_t*)c->ws[c12381_ctx::WS_PROJ], stride, n, red, rstride);
    HIPCK(c, hipGetLastError());
    hipLaunchKernelGGL(g1_add_const_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, c->stream, n, red, rstride, g1_96, c->d_flag);
    HIPCK(c, hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------- BBS+ batch verification
// ok[j] = [ e(A_j, w + x_j g2) == e(g1 + r_j h0 + sum_i m_{i,j} h_i, g2) ]   — the verification equation of the
// reference's examples/bbs-plus/src/bbs+.cpp:57-73, evaluated as liner_pair.hpp:339-350 does (two Miller loops,
// one final exponentiation).  Message scalars are message-major: m[i*n + j] belongs to signature j.  All
// pointers are DEVICE pointers; the public parameters are single points.
int c12381_bbs_plus_verify_batch_dev(c12381_ctx* c, size_t n, size_t nmsg, const uint8_t* g1_96, const uint8_t* g2_192, const uint8_t* h0_96,
                                     const uint8_t* h_96, const uint8_t* w_192, const uint8_t* A_96, const uint8_t* x_32, const uint8_t* r_32,
                                     const uint8_t* m_32, uint8_t* ok) {
    int rc = bind(c); if (rc) return rc;
    if (!g1_96 || !g2_192 || !h0_96 || !w_192 || !A_96 || !x_32 || !r_32 || !ok || (nmsg && (!h_96 || !m_32))) return C12381_E_ARG;
    if (n == 0) return 0;
    // Q_j = w + x_j g2
    if ((rc = ensure(c, c12381_ctx::WS_BBS_Q, 192 * n))) return rc;
    if ((rc = ensure(c, c12381_ctx::WS_BBS_B, 192 * n))) return rc;
    uint8_t* d_q = (uint8_t*)c->ws[c12381_ctx::WS_BBS_Q];
    uint8_t* d_b = (uint8_t*)c->ws[c12381_ctx::WS_BBS_B];
    // Both G2 arguments of the equation are public points.  When g2 and w are elements of G2 the equation is evaluated
    // as e(A, w) * e(x A - B, g2) == 1 (bilinearity in the G2 argument holds for every point A of the curve, and the
    // cofactor part of the GLV multiple x A pairs to 1), so BOTH Miller loops run against fixed G2 points: their line
    // coefficients come from two 69-entry tables, no G2 arithmetic per signature at all.  Otherwise — the reference
    // checks nothing — the generic path below evaluates e(A, w + x g2) == e(B, g2) exactly as written.  `gate` selects:
    // every kernel of either path reads it and returns at once if it belongs to the other path.
    const bool fb = fixed_base_enabled();
    const bool fq = fb && pair_lanes() != 1;
    const int32_t *gate_fast = nullptr, *gate_generic = nullptr;      // skip_if pointers: skip when [HDR_VALID] != 0
    if (fq) {
        if ((rc = lines_table(c, c12381_ctx::WS_FQ_W, w_192, 1))) return rc;
        if ((rc = lines_table(c, c12381_ctx::WS_FQ_G, g2_192, 1))) return rc;
        if ((rc = ensure(c, c12381_ctx::WS_FQ_GATE, 128 * 4))) return rc;
        int32_t* gate = (int32_t*)c->ws[c12381_ctx::WS_FQ_GATE];
        hipLaunchKernelGGL(gate_and_kernel, dim3(1), dim3(BLOCK), 0, c->stream, gate, (const int32_t*)c->ws[c12381_ctx::WS_FQ_W],
                           (const int32_t*)c->ws[c12381_ctx::WS_FQ_G]);
        HIPCK(c, hipGetLastError());
        gate_generic = gate;          // generic kernels: skip when the fixed-G2 path is valid
        gate_fast = gate + GATE_OTHER;        // kernels that exist only for the fixed-G2 path and take a skip pointer: skip when it is not
    }
    // generic path: Q_j = w + x_j g2 (g2's multiples from its fixed-base table when it is a subgroup point)
    const int32_t* skip_g2 = nullptr;
    if (fb && !fq) {
        if ((rc = fixed_table(c, c12381_ctx::WS_FB_G2, g2_192, true))) return rc;
        skip_g2 = (const int32_t*)c->ws[c12381_ctx::WS_FB_G2];
        hipLaunchKernelGGL(g2_fixed_eval_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, c->stream, n, skip_g2, x_32, w_192, d_q, 192, c->d_flag, (int32_t*)nullptr, (size_t)0);
        HIPCK(c, hipGetLastError());
    }
    if ((rc = g2_mul_dev_strided(c, n, g2_192, 0, x_32, d_b, 192, fq ? gate_generic : skip_g2))) return rc;
    hipLaunchKernelGGL(g2_add_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, c->stream, n, w_192, (size_t)0, d_b, d_q, 192, c->d_flag, fq ? gate_generic : skip_g2);
    HIPCK(c, hipGetLastError());
    int32_t* red; size_t rstride, stride;
    if ((rc = bbs_message_points(c, n, nmsg, g1_96, h0_96, h_96, r_32, m_32, fb, red, rstride, stride))) return rc;
    if (fq) {
        // fixed-G2 path: red <- x A - B  (x A by the generic scalar multiplication: A differs per signature)
        if ((rc = g1_mul_to_proj(c, n, A_96, x_32, stride, 96, 0, gate_fast))) return rc;
        hipLaunchKernelGGL(g1_rsub_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, c->stream, n, red, rstride, (const int32_t*)c->ws[c12381_ctx::WS_PROJ], stride,
                           (size_t)0, gate_generic);
        HIPCK(c, hipGetLastError());
    }
    if ((rc = g1_finish(c, n, red, rstride, d_b, 96))) return rc;
    timed tm(c, 4);
    if (fq) {
        uint4* st; unsigned int *fl, *ct, ep; unsigned blocks;
        if ((rc = pair_queue_setup(c, n, st, fl, ct, blocks, &ep))) return rc;
        hipLaunchKernelGGL(pair3_prod_fixed_queue_kernel, dim3(blocks), dim3(BLOCK), 0, c->stream, n, A_96, d_b,
                           (const int32_t*)c->ws[c12381_ctx::WS_FQ_W] + FB_HEADER_DWORDS, (const int32_t*)c->ws[c12381_ctx::WS_FQ_G] + FB_HEADER_DWORDS, ok,
                           c->d_flag, st, fl, ct, gate_generic, pair_spin_limit(), ep);
        HIPCK(c, hipGetLastError());
    }
    return launch_pair_eq(c, n, A_96, d_q, d_b, g2_192, (size_t)0, ok, gate_generic);
}
int c12381_bbs_plus_verify_batch(c12381_ctx* c, size_t n, size_t nmsg, const uint8_t* g1_96, const uint8_t* g2_192, const uint8_t* h0_96,
                                 const uint8_t* h_96, const uint8_t* w_192, const uint8_t* A_96, const uint8_t* x_32, const uint8_t* r_32,
                                 const uint8_t* m_32, uint8_t* ok) {
    int rc = bind(c); if (rc) return rc;
    if (!g1_96 || !g2_192 || !h0_96 || !w_192 || !A_96 || !x_32 || !r_32 || !ok || (nmsg && (!h_96 || !m_32))) return C12381_E_ARG;
    if (n == 0) return 0;
    // one staging slab: public parameters, then the per-signature arrays
    const size_t o_g1 = 0, o_g2 = 96, o_h0 = 288, o_w = 384, o_h = 576, o_A = round_up(o_h + 96 * nmsg, 256), o_x = o_A + 96 * n,
                 o_r = o_x + 32 * n, o_m = o_r + 32 * n, o_ok = round_up(o_m + 32 * n * nmsg, 256), bytes = o_ok + round_up(n, 256);
    if ((rc = ensure(c, c12381_ctx::WS_BBS_IN, bytes))) return rc;
    uint8_t* d = (uint8_t*)c->ws[c12381_ctx::WS_BBS_IN];
    HIPCK(c, hipMemcpyAsync(d + o_g1, g1_96, 96, hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipMemcpyAsync(d + o_g2, g2_192, 192, hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipMemcpyAsync(d + o_h0, h0_96, 96, hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipMemcpyAsync(d + o_w, w_192, 192, hipMemcpyHostToDevice, c->stream));
    if (nmsg) HIPCK(c, hipMemcpyAsync(d + o_h, h_96, 96 * nmsg, hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipMemcpyAsync(d + o_A, A_96, 96 * n, hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipMemcpyAsync(d + o_x, x_32, 32 * n, hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipMemcpyAsync(d + o_r, r_32, 32 * n, hipMemcpyHostToDevice, c->stream));
    if (nmsg) HIPCK(c, hipMemcpyAsync(d + o_m, m_32, 32 * n * nmsg, hipMemcpyHostToDevice, c->stream));
    if ((rc = c12381_bbs_plus_verify_batch_dev(c, n, nmsg, d + o_g1, d + o_g2, d + o_h0, d + o_h, d + o_w, d + o_A, d + o_x, d + o_r, d + o_m, d + o_ok))) return rc;
    HIPCK(c, hipMemcpyAsync(ok, d + o_ok, n, hipMemcpyDeviceToHost, c->stream));
    return read_flag(c);
}

// ---------------------------------------------------------------- BBS+ verification from the wire formats
// The whole caller pattern of examples/bbs-plus/src/bbs+.cpp:57-73 on ONE stream: decode the public points and the signatures'
// A (g1/g2_decompress_kernel), parse x and r, encode the message bytes (bbs_wire_prep_kernel), then the
// verification pipeline above (f2).  Every message has msg_len bytes (ceil(msg_len / 31) units; more units than h entries is
// the reference's "message is too long": C12381_E_ARG).  ok[j] = 1 / 0, or 0xff where the reference would throw.
int c12381_bbs_plus_verify_wire_batch_dev(c12381_ctx* c, size_t n, size_t nh, size_t msg_len, const uint8_t* g1_g2_h0_195, const uint8_t* h_49,
                                          const uint8_t* pk_97, const uint8_t* sig_145, const uint8_t* msgs, uint8_t* ok) {
    int rc = bind(c); if (rc) return rc;
    const size_t nblk = (msg_len + 30) / 31;
    if (!g1_g2_h0_195 || !pk_97 || !sig_145 || !ok || (nh && !h_49) || (msg_len && !msgs) || nblk > nh) return C12381_E_ARG;
    if (n == 0) return 0;
    const size_t npub1 = 2 + nblk;
    // slab: [pub G1 49s][pub G2 97s][pub G1 96s: g1, h0, h...][pub G2 192s: g2, w][status pub1][status pub2] | per signature: a49, A96, x, r, m, status x2
    const size_t o_p49 = 0, o_p97 = round_up(o_p49 + 49 * npub1, 16), o_p96 = round_up(o_p97 + 2 * 97, 256), o_p192 = o_p96 + 96 * npub1,
                 o_st1 = round_up(o_p192 + 384, 16), o_st2 = o_st1 + round_up(npub1, 16), o_a49 = round_up(o_st2 + 16, 256),
                 o_A = round_up(o_a49 + 49 * n, 256), o_x = o_A + 96 * n, o_r = o_x + 32 * n, o_m = o_r + 32 * n,
                 o_ss = round_up(o_m + 32 * n * nblk, 256), o_sa = o_ss + round_up(n, 256), bytes = o_sa + round_up(n, 256);
    if ((rc = ensure(c, c12381_ctx::WS_BBS_WIRE, bytes))) return rc;
    uint8_t* d = (uint8_t*)c->ws[c12381_ctx::WS_BBS_WIRE];
    // The handful of public points decode on the side stream: two square-root chains of one lane each (0.4 + 0.85 ms of pure latency) beside
    // the parsing and the n square roots of the signatures' A on the context's stream, instead of in front of them.
    if ((rc = ensure_fork_event(c))) return rc;
    HIPCK(c, hipEventRecord(c->ev_chunk[0], c->stream));                    // the caller's inputs are ordered on the context's stream
    HIPCK(c, hipStreamWaitEvent(c->side, c->ev_chunk[0], 0));
    hipLaunchKernelGGL(bbs_wire_pub_kernel, dim3(grid_for(49 * npub1 + 2 * 97)), dim3(BLOCK), 0, c->side, nblk, g1_g2_h0_195, h_49, pk_97, d + o_p49, d + o_p97);
    HIPCK(c, hipGetLastError());
    hipLaunchKernelGGL(g1_decompress_kernel, dim3(grid_for(npub1)), dim3(BLOCK), 0, c->side, npub1, d + o_p49, d + o_p96, d + o_st1, 0);
    hipLaunchKernelGGL(g2_decompress_kernel, dim3(1), dim3(BLOCK), 0, c->side, (size_t)2, d + o_p97, d + o_p192, d + o_st2, 0);
    HIPCK(c, hipGetLastError());
    HIPCK(c, hipEventRecord(c->ev_side, c->side));
    hipLaunchKernelGGL(bbs_wire_prep_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, c->stream, n, msg_len, nblk, sig_145, msgs, d + o_a49, d + o_x, d + o_r, d + o_m, d + o_ss);
    hipLaunchKernelGGL(g1_decompress_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, c->stream, n, d + o_a49, d + o_A, d + o_sa, 0);
    HIPCK(c, hipGetLastError());
    HIPCK(c, hipStreamWaitEvent(c->stream, c->ev_side, 0));
    if ((rc = c12381_bbs_plus_verify_batch_dev(c, n, nblk, d + o_p96, d + o_p192, d + o_p96 + 96, d + o_p96 + 192, d + o_p192 + 192, d + o_A, d + o_x, d + o_r,
                                               d + o_m, ok))) return rc;
    hipLaunchKernelGGL(bbs_wire_finish_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, c->stream, n, npub1, d + o_ss, d + o_sa, d + o_st1, d + o_st2, ok, c->d_flag);
    HIPCK(c, hipGetLastError());
    return 0;
}
int c12381_bbs_plus_verify_wire_batch(c12381_ctx* c, size_t n, size_t nh, size_t msg_len, const uint8_t* g1_g2_h0_195, const uint8_t* h_49,
                                      const uint8_t* pk_97, const uint8_t* sig_145, const uint8_t* msgs, uint8_t* ok) {
    int rc = bind(c); if (rc) return rc;
    const size_t nblk = (msg_len + 30) / 31;
    if (!g1_g2_h0_195 || !pk_97 || !sig_145 || !ok || (nh && !h_49) || (msg_len && !msgs) || nblk > nh) return C12381_E_ARG;
    if (n == 0) return 0;
    const size_t o_pp = 0, o_pk = 256, o_h = 512, o_sig = round_up(o_h + 49 * nh, 256), o_msg = round_up(o_sig + 145 * n, 256),
                 o_ok = round_up(o_msg + msg_len * n, 256), bytes = o_ok + round_up(n, 256);
    if ((rc = ensure(c, c12381_ctx::WS_BBS_WIRE_IN, bytes))) return rc;
    uint8_t* d = (uint8_t*)c->ws[c12381_ctx::WS_BBS_WIRE_IN];
    HIPCK(c, hipMemcpyAsync(d + o_pp, g1_g2_h0_195, 195, hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipMemcpyAsync(d + o_pk, pk_97, 97, hipMemcpyHostToDevice, c->stream));
    if (nh) HIPCK(c, hipMemcpyAsync(d + o_h, h_49, 49 * nh, hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipMemcpyAsync(d + o_sig, sig_145, 145 * n, hipMemcpyHostToDevice, c->stream));
    if (msg_len) HIPCK(c, hipMemcpyAsync(d + o_msg, msgs, msg_len * n, hipMemcpyHostToDevice, c->stream));
    if ((rc = c12381_bbs_plus_verify_wire_batch_dev(c, n, nh, msg_len, d + o_pp, d + o_h, d + o_pk, d + o_sig, d + o_msg, d + o_ok))) return rc;
    HIPCK(c, hipMemcpyAsync(ok, d + o_ok, n, hipMemcpyDeviceToHost, c->stream));
    return read_flag(c);
}

// ---------------------------------------------------------------- BBS+ aggregate verification (optional)
// ONE verdict for the whole batch by a random linear combination: with caller-drawn rho_j,
//   prod_j [ e(A_j, w) e(x_j A_j - B_j, g2) ]^rho_j
//     = e( sum_j rho_j A_j, w ) * e( sum_j (rho_j x_j) A_j - (sum_j rho_j) g1 - (sum_j rho_j r_j) h0 - sum_i (sum_j rho_j m_ij) h_i, g2 )
// — the per-signature point arithmetic collapses into inner products mod r, two bucket products over the A_j and one
// product of two pairings.  all_ok = 1 iff g2, w are elements of G2 and the combined product is 1; every signature the
// per-signature entry accepts contributes a factor 1 (cofactor components of any argument pair to 1 against G2), so
// a batch of valid signatures always yields 1, and a batch containing an invalid one yields 1 with probability at most
// 2^-k over k-bit uniform rho_j.  all_ok = 0 settles nothing: the caller then runs the per-signature entry.
// The reference has no such mode (it verifies one signature at a time, bbs+.cpp:57-73); the booleans of
// c12381_bbs_plus_verify_batch stay the parity surface.
int c12381_bbs_plus_verify_aggregate_dev(c12381_ctx* c, size_t n, size_t nmsg, const uint8_t* g1_96, const uint8_t* g2_192, const uint8_t* h0_96,
                                         const uint8_t* h_96, const uint8_t* w_192, const uint8_t* A_96, const uint8_t* x_32, const uint8_t* r_32,
                                         const uint8_t* m_32, const uint8_t* rho_32, uint8_t* all_ok) {
    int rc = bind(c); if (rc) return rc;
    if (!g1_96 || !g2_192 || !h0_96 || !w_192 || !all_ok || (n && (!A_96 || !x_32 || !r_32 || !rho_32)) || (nmsg && (!h_96 || (n && !m_32)))) return C12381_E_ARG;
    const size_t terms = n + nmsg + 2;
    if (terms > MSM_MAX_TERMS) return C12381_E_ARG;            // split the batch: one bucket product per call
    if (n == 0) { HIPCK(c, hipMemsetAsync(all_ok, 1, 1, c->stream)); return 0; }
    HIPCK(c, hipMemsetAsync(all_ok, 0, 1, c->stream));
    if ((rc = lines_table(c, c12381_ctx::WS_FQ_W, w_192, 1))) return rc;
    if ((rc = lines_table(c, c12381_ctx::WS_FQ_G, g2_192, 1))) return rc;
    if ((rc = ensure(c, c12381_ctx::WS_FQ_GATE, 128 * 4))) return rc;
    int32_t* gate = (int32_t*)c->ws[c12381_ctx::WS_FQ_GATE];
    hipLaunchKernelGGL(gate_and_kernel, dim3(1), dim3(BLOCK), 0, c->stream, gate, (const int32_t*)c->ws[c12381_ctx::WS_FQ_W],
                       (const int32_t*)c->ws[c12381_ctx::WS_FQ_G]);
    HIPCK(c, hipGetLastError());
    const size_t o_p = round_up(96 * terms, 256);
    if ((rc = ensure(c, c12381_ctx::WS_BBS_B, o_p + 256))) return rc;
    if ((rc = ensure(c, c12381_ctx::WS_BBS_Q, 32 * terms))) return rc;
    uint8_t* pts = (uint8_t*)c->ws[c12381_ctx::WS_BBS_B];        // A_1 .. A_n, g1, h0, h_1 .. h_nmsg | P1, P2
    uint8_t* sc = (uint8_t*)c->ws[c12381_ctx::WS_BBS_Q];         // rho_j x_j | -sum rho_j, -sum rho_j r_j, -sum_j rho_j m_ij
    uint8_t *tail = sc + 32 * n, *p1 = pts + o_p, *p2 = p1 + 96;
    hipLaunchKernelGGL(zp_op_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, c->stream, 0, n, rho_32, x_32, sc);
    HIPCK(c, hipGetLastError());
    {   // tail[y] = sum_j rho_j * (1 | r_j | m_{y-2,j}): all nmsg + 2 inner products as columns of the same fold stages
        const size_t cols = nmsg + 2;
        const uint8_t* cur_a = rho_32;
        size_t cur_n = n, col_stride = 0;
        int slot = c12381_ctx::WS_RED0, first = 1;
        for (;;) {
            const size_t T = (cur_n + 63) / 64;
            uint8_t* dst = tail;
            if (T > 1) {
                if ((rc = ensure(c, slot, round_up(32 * T * cols, 256)))) return rc;
                dst = (uint8_t*)c->ws[slot];
            }
            hipLaunchKernelGGL(zp_fold_cols_kernel, dim3(grid_for(T), (unsigned)cols), dim3(BLOCK), 0, c->stream, cur_n, cur_a, col_stride, r_32, m_32, first, T, dst);
            HIPCK(c, hipGetLastError());
            if (T == 1) break;
            cur_a = dst; col_stride = 32 * T; cur_n = T; first = 0;
            slot = slot == c12381_ctx::WS_RED0 ? c12381_ctx::WS_RED1 : c12381_ctx::WS_RED0;
        }
    }
    hipLaunchKernelGGL(zp_op_kernel, dim3(grid_for(nmsg + 2)), dim3(BLOCK), 0, c->stream, 3, nmsg + 2, (const uint8_t*)tail, (const uint8_t*)nullptr, tail);
    HIPCK(c, hipGetLastError());
    HIPCK(c, hipMemcpyAsync(pts, A_96, 96 * n, hipMemcpyDeviceToDevice, c->stream));
    HIPCK(c, hipMemcpyAsync(pts + 96 * n, g1_96, 96, hipMemcpyDeviceToDevice, c->stream));
    HIPCK(c, hipMemcpyAsync(pts + 96 * (n + 1), h0_96, 96, hipMemcpyDeviceToDevice, c->stream));
    if (nmsg) HIPCK(c, hipMemcpyAsync(pts + 96 * (n + 2), h_96, 96 * nmsg, hipMemcpyDeviceToDevice, c->stream));
    if ((rc = c12381_g1_msm_dev(c, n, A_96, rho_32, p1, 96))) return rc;
    if ((rc = c12381_g1_msm_dev(c, terms, pts, sc, p2, 96))) return rc;
    uint4* st; unsigned int *fl, *ct, ep; unsigned blocks;
    if ((rc = pair_queue_setup(c, 1, st, fl, ct, blocks, &ep))) return rc;
    hipLaunchKernelGGL(pair3_prod_fixed_queue_kernel, dim3(blocks), dim3(BLOCK), 0, c->stream, (size_t)1, (const uint8_t*)p1, (const uint8_t*)p2,
                       (const int32_t*)c->ws[c12381_ctx::WS_FQ_W] + FB_HEADER_DWORDS, (const int32_t*)c->ws[c12381_ctx::WS_FQ_G] + FB_HEADER_DWORDS, all_ok,
                       c->d_flag, st, fl, ct, (const int32_t*)gate, pair_spin_limit(), ep);
    HIPCK(c, hipGetLastError());
    return 0;
}
int c12381_bbs_plus_verify_aggregate(c12381_ctx* c, size_t n, size_t nmsg, const uint8_t* g1_96, const uint8_t* g2_192, const uint8_t* h0_96,
                                     const uint8_t* h_96, const uint8_t* w_192, const uint8_t* A_96, const uint8_t* x_32, const uint8_t* r_32,
                                     const uint8_t* m_32, const uint8_t* rho_32, int* all_ok) {
    int rc = bind(c); if (rc) return rc;
    if (!g1_96 || !g2_192 || !h0_96 || !w_192 || !all_ok || (n && (!A_96 || !x_32 || !r_32 || !rho_32)) || (nmsg && (!h_96 || (n && !m_32)))) return C12381_E_ARG;
    *all_ok = 0;
    if (n == 0) { *all_ok = 1; return 0; }
    const size_t o_g1 = 0, o_g2 = 96, o_h0 = 288, o_w = 384, o_h = 576, o_A = round_up(o_h + 96 * nmsg, 256), o_x = o_A + 96 * n,
                 o_r = o_x + 32 * n, o_rho = o_r + 32 * n, o_m = o_rho + 32 * n, o_ok = round_up(o_m + 32 * n * nmsg, 256), bytes = o_ok + 256;
    if ((rc = ensure(c, c12381_ctx::WS_BBS_IN, bytes))) return rc;
    uint8_t* d = (uint8_t*)c->ws[c12381_ctx::WS_BBS_IN];
    HIPCK(c, hipMemcpyAsync(d + o_g1, g1_96, 96, hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipMemcpyAsync(d + o_g2, g2_192, 192, hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipMemcpyAsync(d + o_h0, h0_96, 96, hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipMemcpyAsync(d + o_w, w_192, 192, hipMemcpyHostToDevice, c->stream));
    if (nmsg) HIPCK(c, hipMemcpyAsync(d + o_h, h_96, 96 * nmsg, hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipMemcpyAsync(d + o_A, A_96, 96 * n, hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipMemcpyAsync(d + o_x, x_32, 32 * n, hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipMemcpyAsync(d + o_r, r_32, 32 * n, hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipMemcpyAsync(d + o_rho, rho_32, 32 * n, hipMemcpyHostToDevice, c->stream));
    if (nmsg) HIPCK(c, hipMemcpyAsync(d + o_m, m_32, 32 * n * nmsg, hipMemcpyHostToDevice, c->stream));
    if ((rc = c12381_bbs_plus_verify_aggregate_dev(c, n, nmsg, d + o_g1, d + o_g2, d + o_h0, d + o_h, d + o_w, d + o_A, d + o_x, d + o_r, d + o_m, d + o_rho,
                                                   d + o_ok))) return rc;
    uint8_t verdict = 0;
    HIPCK(c, hipMemcpyAsync(&verdict, d + o_ok, 1, hipMemcpyDeviceToHost, c->stream));
    rc = read_flag(c);                                          // synchronises the stream
    *all_ok = (rc == 0 && verdict == 1) ? 1 : 0;
    return rc;
}

// BBS+ signing for a batch (bbs+.cpp:38-55): A_j = (g1 * h0^r_j * prod_i h_i^m_ij)^(1/(gamma + x_j)).  x_j, r_j are the
// caller's random scalars (the reference draws them inside sign()); inverse(0) = 0 gives the point at infinity, as there.
int c12381_bbs_plus_sign_batch_dev(c12381_ctx* c, size_t n, size_t nmsg, const uint8_t* g1_96, const uint8_t* h0_96, const uint8_t* h_96,
                                   const uint8_t* gamma_32, const uint8_t* x_32, const uint8_t* r_32, const uint8_t* m_32, uint8_t* A_out96) {
    int rc = bind(c); if (rc) return rc;
    if (!g1_96 || !h0_96 || !gamma_32 || !x_32 || !r_32 || !A_out96 || (nmsg && (!h_96 || !m_32))) return C12381_E_ARG;
    if (n == 0) return 0;
    if ((rc = ensure(c, c12381_ctx::WS_BBS_B, 192 * n))) return rc;
    if ((rc = ensure(c, c12381_ctx::WS_BBS_Q, 192 * n))) return rc;
    uint8_t* d_b = (uint8_t*)c->ws[c12381_ctx::WS_BBS_B];         // B_j affine
    uint8_t* d_e = (uint8_t*)c->ws[c12381_ctx::WS_BBS_Q];         // 1 / (gamma + x_j)
    int32_t* red; size_t rstride, stride;
    if ((rc = bbs_message_points(c, n, nmsg, g1_96, h0_96, h_96, r_32, m_32, fixed_base_enabled(), red, rstride, stride))) return rc;
    if ((rc = g1_finish(c, n, red, rstride, d_b, 96))) return rc;
    if ((rc = zp_batch_inverse(c, n, x_32, gamma_32, d_e))) return rc;
    return c12381_g1_mul_batch_dev(c, n, d_b, d_e, A_out96, 96);
}
int c12381_bbs_plus_sign_batch(c12381_ctx* c, size_t n, size_t nmsg, const uint8_t* g1_96, const uint8_t* h0_96, const uint8_t* h_96,
                               const uint8_t* gamma_32, const uint8_t* x_32, const uint8_t* r_32, const uint8_t* m_32, uint8_t* A_out96) {
    int rc = bind(c); if (rc) return rc;
    if (!g1_96 || !h0_96 || !gamma_32 || !x_32 || !r_32 || !A_out96 || (nmsg && (!h_96 || !m_32))) return C12381_E_ARG;
    if (n == 0) return 0;
    const size_t o_g1 = 0, o_h0 = 96, o_gm = 192, o_h = 256, o_x = round_up(o_h + 96 * nmsg, 256), o_r = o_x + 32 * n, o_m = o_r + 32 * n,
                 o_A = round_up(o_m + 32 * n * nmsg, 256), bytes = o_A + 96 * n;
    if ((rc = ensure(c, c12381_ctx::WS_BBS_IN, bytes))) return rc;
    uint8_t* d = (uint8_t*)c->ws[c12381_ctx::WS_BBS_IN];
    HIPCK(c, hipMemcpyAsync(d + o_g1, g1_96, 96, hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipMemcpyAsync(d + o_h0, h0_96, 96, hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipMemcpyAsync(d + o_gm, gamma_32, 32, hipMemcpyHostToDevice, c->stream));
    if (nmsg) HIPCK(c, hipMemcpyAsync(d + o_h, h_96, 96 * nmsg, hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipMemcpyAsync(d + o_x, x_32, 32 * n, hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipMemcpyAsync(d + o_r, r_32, 32 * n, hipMemcpyHostToDevice, c->stream));
    if (nmsg) HIPCK(c, hipMemcpyAsync(d + o_m, m_32, 32 * n * nmsg, hipMemcpyHostToDevice, c->stream));
    if ((rc = c12381_bbs_plus_sign_batch_dev(c, n, nmsg, d + o_g1, d + o_h0, d + o_h, d + o_gm, d + o_x, d + o_r, d + o_m, d + o_A))) return rc;
    HIPCK(c, hipMemcpyAsync(A_out96, d + o_A, 96 * n, hipMemcpyDeviceToHost, c->stream));
    return read_flag(c);
}

}  // extern "C"
