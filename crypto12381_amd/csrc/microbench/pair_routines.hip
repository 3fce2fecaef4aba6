// Routine-level microbenchmark of the three-lane pairing arithmetic (pairing3.hpp): every wavefront calls ONE out-of-line
// routine `iters` times on its LDS slot, exactly as the pairing kernels do, at the kernels' launch bounds.  Output per
// routine: wall time, SIMD cycles per call (at the clock the chip held, measured with s_memtime / s_memrealtime) — to be
// compared with the static instruction mix of the routine (tools/isa_stats.py): issue cycles vs stall cycles.
//
// Build:  hipcc -O3 --offload-arch=gfx950 -std=c++17 -fno-optimize-sibling-calls -mllvm -amdgpu-sched-strategy=max-ilp -mllvm -opt-disable=reassociate pair_routines.hip -o pair_routines
// Run:    ./pair_routines [waves_per_simd=2] [iters=200]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <algorithm>

#include "../kernels_common.hpp"
#include "../pairing3.hpp"

using namespace c12381;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

#ifndef PR_WAVES
#define PR_WAVES 2
#endif

typedef pair_slot fp4_slot;

__device__ __forceinline__ void load_fp(fp& r, const int32_t* p) {
#pragma unroll
    for (int i = 0; i < NL; ++i) r.l[i] = C12381_LIMB(p[i] & (int32_t)LMASK);
}
__device__ __forceinline__ void load_fp4(fp4& r, const int32_t* p) {
    load_fp(r.a.a, p); load_fp(r.a.b, p + NL); load_fp(r.b.a, p + 2 * NL); load_fp(r.b.b, p + 3 * NL);
}
__device__ __forceinline__ void store_fp4(int32_t* p, const fp4& r) {
    const int32_t* w = reinterpret_cast<const int32_t*>(&r);
#pragma unroll
    for (int i = 0; i < 4 * NL; ++i) p[i] = w[i];
}

template <int KIND>
__global__ void __launch_bounds__(BLOCK, PR_WAVES) routine_kernel(int iters, const int32_t* seed, int32_t* sink, uint64_t* stamps) {
    __shared__ fp4_slot slots[BLOCK];
    slot_fair_set(slots[threadIdx.x].v, 0);
    fp4& H = slots[threadIdx.x].v;
    const unsigned lane = threadIdx.x & 63u;
    const unsigned trip = lane / 3u;
    tri t;
    t.role = lane == 63u ? 0 : (int)(lane - 3u * trip);
    t.base = lane == 63u ? 63 : (int)(3u * trip);
    const size_t gid = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    const int32_t* sp = seed + (gid % 4096) * 8 * NL;
    fp4 a; fp2 tc; fp px, py;
    load_fp4(a, sp);
    { fp4 h0; load_fp4(h0, sp + 4 * NL); slot_store(H, h0); slot_psel_store(H, a.a.a); }
    tc = a.b; px = a.a.a; py = a.a.b;
    uint64_t t0 = 0, r0 = 0;
    if (threadIdx.x == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        if constexpr (KIND == 0) f12t_sqr_h(H, t);
        else if constexpr (KIND == 1) miller3_dbl_line(H, tc, px, py, false, t);
        else if constexpr (KIND == 2) f12t_usqr_h(H, (it & 1) == 0, t);
        else if constexpr (KIND == 16) f12t_usqr3_h(H, t);                                       // round 4: scaled form, injected linear terms
        else if constexpr (KIND == 3) f12t_mul_h(H, a, t);
        else if constexpr (KIND == 10) { fp4 w; f12t_mul(w, a, a, t); a = w; }
        else if constexpr (KIND == 4) f12t_mul_line_h(H, tc, a.a, a.b, t);
        else if constexpr (KIND == 5) { fp4 w; fp4_mul_call(w, a, a); a = w; }
        else if constexpr (KIND == 6) { fp4 w; f12t_frob(w, a, t); a = w; }
        else if constexpr (KIND == 7) { fp r; fp_mul(r, px, py); fp_mul(px, r, py); }          // two dependent Fp products, registers only
        else if constexpr (KIND == 8) { fp2 r; fp2_mul(r, tc, a.a); fp2_mul(tc, r, a.a); }      // two dependent Fp2 products
        else if constexpr (KIND == 9) { f12t_sqr_h(H, t); miller3_dbl_line(H, tc, px, py, false, t); }   // one Miller iteration without addition step
        else if constexpr (KIND == 12) miller3_fixed_line_raw(H, seed, it & 63, px, py, false, t);      // line records: any normalised limbs (timing only)
        else if constexpr (KIND == 13) miller3_fixed_line1(H, seed, it & 63, px, py, false, t);
        else if constexpr (KIND == 14) miller3_range2_fixed(H, px, py, false, seed + 16384, py, px, false, seed + 32768, 64, 1, t);   // raw tables (format word 0)
        else if constexpr (KIND == 15) miller3_range2_fixed(H, px, py, false, seed + 65536, py, px, false, seed + 81920, 64, 1, t);   // normalised tables (format word 1)
        else if constexpr (KIND == 11) { miller3_regs R = m3r_pack(tc, H, t.role | (t.base << 2)); R = miller3_iter(R); m3r_tc(tc, R); }   // the same as ONE routine
    }
    if (threadIdx.x == 0) {
        const uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0;
    }
    fp4 o;
    slot_load(o, H);
    fp4_add(o, o, a);
    fp_add(o.a.a, o.a.a, px); fp_add(o.a.b, o.a.b, py); fp_add(o.b.a, o.b.a, tc.a); fp_add(o.b.b, o.b.b, tc.b);
    store_fp4(sink + gid * 4 * NL, o);
}

template <int KIND>
int run(const char* name, int iters, int blocks, const int32_t* d_seed, int32_t* d_sink, uint64_t* d_stamps, int waves_per_simd) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(routine_kernel<KIND>, dim3(blocks), dim3(BLOCK), 0, 0, 4, d_seed, d_sink, d_stamps);     // warm-up
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(routine_kernel<KIND>, dim3(blocks), dim3(BLOCK), 0, 0, iters, d_seed, d_sink, d_stamps);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<uint64_t> st(2 * blocks);
    CK(hipMemcpy(st.data(), d_stamps, st.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> clk, cyc;
    for (int b = 0; b < blocks; ++b) if (st[2 * b + 1]) { clk.push_back((double)st[2 * b] / (double)st[2 * b + 1] * 100e6); cyc.push_back((double)st[2 * b]); }
    std::sort(clk.begin(), clk.end()); std::sort(cyc.begin(), cyc.end());
    const double ghz = clk.empty() ? 0 : clk[clk.size() / 2] / 1e9;
    const double wave_cycles = cyc.empty() ? 0 : cyc[cyc.size() / 2] / iters;          // shader cycles of one wave's lifetime per call
    printf("ROUTINE %-22s waves/SIMD %d  iters %5d  %8.3f ms  clock %.3f GHz  wave-cycles/call %9.0f  SIMD-cycles/call %9.0f\n", name, waves_per_simd, iters, ms, ghz,
           wave_cycles, wave_cycles / waves_per_simd);
    fflush(stdout);
    return 0;
}

int main(int argc, char** argv) {
    const int wps = argc > 1 ? atoi(argv[1]) : PR_WAVES;
    const int iters = argc > 2 ? atoi(argv[2]) : 200;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int blocks = cus * wps;                       // 4 waves per block: wps waves per SIMD
    printf("DEVICE %s CUs %d  launch-bounds waves %d  resident waves/SIMD %d\n", prop.name, cus, PR_WAVES, wps);
    std::vector<int32_t> seed((size_t)4096 * 8 * NL);
    uint64_t s = 0x9e3779b97f4a7c15ull;
    for (auto& v : seed) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (int32_t)(s & LMASK); }
    seed[16384 + FQ_FMT_WORD] = 0; seed[32768 + FQ_FMT_WORD] = 0; seed[65536 + FQ_FMT_WORD] = 1; seed[81920 + FQ_FMT_WORD] = 1;
    int32_t *d_seed, *d_sink; uint64_t* d_stamps;
    CK(hipMalloc(&d_seed, seed.size() * 4));
    CK(hipMalloc(&d_sink, (size_t)blocks * BLOCK * 4 * NL * 4));
    CK(hipMalloc(&d_stamps, (size_t)blocks * 16));
    CK(hipMemcpy(d_seed, seed.data(), seed.size() * 4, hipMemcpyHostToDevice));
    if (run<7>("2x fp_mul (regs)", iters * 20, blocks, d_seed, d_sink, d_stamps, wps)) return 1;
    if (run<8>("2x fp2_mul (regs)", iters * 8, blocks, d_seed, d_sink, d_stamps, wps)) return 1;
    if (run<5>("fp4_mul_call", iters * 2, blocks, d_seed, d_sink, d_stamps, wps)) return 1;
    if (run<0>("f12t_sqr_h", iters, blocks, d_seed, d_sink, d_stamps, wps)) return 1;
    if (run<1>("miller3_dbl_line", iters, blocks, d_seed, d_sink, d_stamps, wps)) return 1;
    if (run<9>("sqr + dbl_line", iters, blocks, d_seed, d_sink, d_stamps, wps)) return 1;
    if (run<11>("miller3_iter", iters, blocks, d_seed, d_sink, d_stamps, wps)) return 1;
    if (run<12>("fixed_line (raw)", iters, blocks, d_seed, d_sink, d_stamps, wps)) return 1;
    if (run<13>("fixed_line (l1 = 1)", iters, blocks, d_seed, d_sink, d_stamps, wps)) return 1;
    if (run<14>("range2_fixed raw", iters / 20, blocks, d_seed, d_sink, d_stamps, wps)) return 1;
    if (run<15>("range2_fixed l1 = 1", iters / 20, blocks, d_seed, d_sink, d_stamps, wps)) return 1;
    if (run<2>("f12t_usqr_h", iters * 2, blocks, d_seed, d_sink, d_stamps, wps)) return 1;
    if (run<16>("f12t_usqr3_h", iters * 2, blocks, d_seed, d_sink, d_stamps, wps)) return 1;
    if (run<3>("f12t_mul_h (LDS)", iters, blocks, d_seed, d_sink, d_stamps, wps)) return 1;
    if (run<10>("f12t_mul (private)", iters, blocks, d_seed, d_sink, d_stamps, wps)) return 1;
    if (run<4>("f12t_mul_line_h", iters, blocks, d_seed, d_sink, d_stamps, wps)) return 1;
    if (run<6>("f12t_frob", iters * 2, blocks, d_seed, d_sink, d_stamps, wps)) return 1;
    return 0;
}
