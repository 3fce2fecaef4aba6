// Calibration microbenchmark for the integer-VALU roofline of the BLS12-381 hot path
// ("the build must ship a microbenchmark ... and use the measured peak").
//
//  part 1: issue rate of the instructions a multi-precision multiply is made of
//          (v_mad_u64_u32, v_mul_lo/hi_u32, 24-bit mads, carry adds, 64-bit shifts, v_fma_f64)
//  part 2: throughput of two candidate Fp Montgomery multipliers as hipcc compiles them:
//          S  = 12 x 32-bit saturated limbs (R = 2^384)
//          U  = 14 x 29-bit unsaturated limbs (R = 2^406), one 64-bit column accumulator
//
// NOTE (round 3): part 1 times 16 instructions per loop iteration with events.  The loop's own scalar instructions and the uneven
// arrival of the workgroups are inside those rates; issue_mix.hip counts cycles inside the kernel with 128 instructions per
// iteration and is what bench.py's peak comes from.  This file stays for the relative rates of the instruction kinds.
// Build:  hipcc -O3 --offload-arch=gfx950 valu_rates.hip -o valu_rates
// Run:    ./valu_rates [out.txt]       (prints one line per measurement)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <algorithm>
#include <vector>
#include <string>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

constexpr int ITERS = 4096;

// ---------------------------------------------------------------- part 1
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

// clk[4 * wave + {0,1}] = s_memtime (shader cycles), [2,3] = s_memrealtime (100 MHz) around the loop: the in-kernel clock
// is d(memtime) / d(memrealtime) x 100 MHz (MI355X_MICROARCH.md, DVFS give-back item 6)
template <int KIND>
__global__ void __launch_bounds__(256) rate_kernel(uint32_t* out, uint32_t seed, unsigned long long* clk, int iters) {
    uint32_t a = seed + threadIdx.x, b = seed * 3 + blockIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();
    uint64_t acc[8];
    uint32_t r[8];
    double d[8];
    for (int i = 0; i < 8; ++i) { acc[i] = a + i; r[i] = b + i; d[i] = (double)(a + i); }
    double da = (double)a * 1e-9, db = (double)b * 1e-9;
    for (int it = 0; it < iters; ++it) {
        if constexpr (KIND == 0) {       // v_mad_u64_u32, 8 independent 64-bit accumulators
#define X(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc");
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 1) { // v_mul_lo_u32
#define X(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 2) { // v_mul_hi_u32
#define X(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 3) { // v_mad_u32_u24
#define X(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 4) { // v_add_co_u32 + v_addc_co_u32 (one carry pair)
#define X(i) asm volatile("v_add_co_u32 %0, vcc, %0, %1\n\tv_addc_co_u32 %0, vcc, %0, %2, vcc" : "+v"(r[i]) : "v"(a), "v"(b) : "vcc");
            REP8(X)
#undef X
        } else if constexpr (KIND == 5) { // v_add_u32
#define X(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 6) { // v_fma_f64
#define X(i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[i]) : "v"(da), "v"(db));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 7) { // v_lshrrev_b64
#define X(i) asm volatile("v_lshrrev_b64 %0, 1, %0" : "+v"(acc[i]));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 8) { // v_alignbit_b32
#define X(i) asm volatile("v_alignbit_b32 %0, %1, %0, 29" : "+v"(r[i]) : "v"(a));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 9) { // v_add3_u32
#define X(i) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 10) { // v_mul_hi_u32_u24
#define X(i) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(r[i]) : "v"(a));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 11) { // v_fma_f32
            float* f = reinterpret_cast<float*>(r);
#define X(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f[i]) : "v"(a), "v"(b));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 12) { // v_mad_u64_u32 single dependent chain (latency)
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
                         "v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
                         "v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
                         "v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
                         "v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
                         "v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
                         "v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
                         "v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0"
                         : "+v"(acc[0]) : "v"(a), "v"(b) : "vcc");
        } else if constexpr (KIND == 13) { // v_mad_u64_u32 with SGPR-pair carry-out other than vcc
#define X(i) asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "s20", "s21");
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 14) { // v_mad_u32_u16 / v_mad_u16? use v_mul_u32_u24
#define X(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(r[i]) : "v"(a));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 16) { // v_mad_i64_i32, the instruction every limb product of fp.hpp compiles to (carry-out in an SGPR pair, as the compiler emits it)
#define X(i) asm volatile("v_mad_i64_i32 %0, s[20:21], %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "s20", "s21");
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 17) { // v_mad_i64_i32, ONE dependent accumulator chain (a column of fp_mul)
            asm volatile("v_mad_i64_i32 %0, s[20:21], %1, %2, %0\n\tv_mad_i64_i32 %0, s[20:21], %1, %2, %0\n\t"
                         "v_mad_i64_i32 %0, s[20:21], %1, %2, %0\n\tv_mad_i64_i32 %0, s[20:21], %1, %2, %0\n\t"
                         "v_mad_i64_i32 %0, s[20:21], %1, %2, %0\n\tv_mad_i64_i32 %0, s[20:21], %1, %2, %0\n\t"
                         "v_mad_i64_i32 %0, s[20:21], %1, %2, %0\n\tv_mad_i64_i32 %0, s[20:21], %1, %2, %0\n\t"
                         "v_mad_i64_i32 %0, s[20:21], %1, %2, %0\n\tv_mad_i64_i32 %0, s[20:21], %1, %2, %0\n\t"
                         "v_mad_i64_i32 %0, s[20:21], %1, %2, %0\n\tv_mad_i64_i32 %0, s[20:21], %1, %2, %0\n\t"
                         "v_mad_i64_i32 %0, s[20:21], %1, %2, %0\n\tv_mad_i64_i32 %0, s[20:21], %1, %2, %0\n\t"
                         "v_mad_i64_i32 %0, s[20:21], %1, %2, %0\n\tv_mad_i64_i32 %0, s[20:21], %1, %2, %0"
                         : "+v"(acc[0]) : "v"(a), "v"(b) : "s20", "s21");
        } else if constexpr (KIND == 18) { // the mix of an fp_mul column scan: 12 v_mad_i64_i32 + 1 v_mul_lo + 1 v_and + 1 v_ashrrev_i64 + 1 v_and
#define X(i) asm volatile("v_mad_i64_i32 %0, s[20:21], %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "s20", "s21");
            REP8(X) X(0) X(1) X(2) X(3)
#undef X
            asm volatile("v_mul_lo_u32 %0, %1, %2\n\tv_and_b32 %0, 0xfffffff, %0" : "+v"(r[0]) : "v"((uint32_t)acc[4]), "v"(a));
            asm volatile("v_ashrrev_i64 %0, 28, %0\n\tv_and_b32 %1, 0xfffffff, %1" : "+v"(acc[5]), "+v"(r[1]));
        } else if constexpr (KIND == 15) { // v_lshl_add_u32
#define X(i) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(r[i]) : "v"(a));
            REP8(X) REP8(X)
#undef X
        }
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; ++i) s += (uint32_t)acc[i] + (uint32_t)(acc[i] >> 32) + r[i] + (uint32_t)d[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime();
    if (clk && (threadIdx.x & 63) == 0) {
        const size_t w = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
        clk[4 * w] = t0; clk[4 * w + 1] = t1; clk[4 * w + 2] = w0; clk[4 * w + 3] = w1;
    }
}

// ---------------------------------------------------------------- part 2: S (12 x 32)
namespace S {
__constant__ uint32_t P[12] = {0xffffaaab, 0xb9feffff, 0xb153ffff, 0x1eabfffe, 0xf6b0f624, 0x6730d2a0,
                               0xf38512bf, 0x64774b84, 0x434bacd7, 0x4b1ba7b6, 0x397fe69a, 0x1a0111ea};
constexpr uint32_t PI[12] = {0xffffaaab, 0xb9feffff, 0xb153ffff, 0x1eabfffe, 0xf6b0f624, 0x6730d2a0,
                             0xf38512bf, 0x64774b84, 0x434bacd7, 0x4b1ba7b6, 0x397fe69a, 0x1a0111ea};
constexpr uint32_t N0 = 0xfffcfffd;

__device__ __forceinline__ void mont_mul(uint32_t* r, const uint32_t* a, const uint32_t* b) {
    uint32_t t[13];
#pragma unroll
    for (int i = 0; i < 13; ++i) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            c += (uint64_t)a[j] * b[i] + t[j];
            t[j] = (uint32_t)c;
            c >>= 32;
        }
        c += t[12];
        t[12] = (uint32_t)c;
        uint32_t t13 = (uint32_t)(c >> 32);
        uint32_t m = t[0] * N0;
        c = ((uint64_t)m * PI[0] + t[0]) >> 32;
#pragma unroll
        for (int j = 1; j < 12; ++j) {
            c += (uint64_t)m * PI[j] + t[j];
            t[j - 1] = (uint32_t)c;
            c >>= 32;
        }
        c += t[12];
        t[11] = (uint32_t)c;
        t[12] = t13 + (uint32_t)(c >> 32);
    }
    // conditional subtract
    uint32_t s[12];
    uint64_t bw = 0;
#pragma unroll
    for (int j = 0; j < 12; ++j) {
        uint64_t d = (uint64_t)t[j] - PI[j] - bw;
        s[j] = (uint32_t)d;
        bw = (d >> 32) & 1;
    }
    bool ge = t[12] != 0 || bw == 0;
#pragma unroll
    for (int j = 0; j < 12; ++j) r[j] = ge ? s[j] : t[j];
}
} // namespace S

// ---------------------------------------------------------------- part 2: U (14 x 29)
namespace U {
constexpr uint32_t PI[14] = {0x1fffaaab, 0xff7ffff, 0x14ffffee, 0x17fffd62, 0xf6241ea, 0x9507b58, 0xafd9cc3,
                             0x109e70a2, 0x1764774b, 0x121a5d66, 0x12c6e9ed, 0x12ffcd34, 0x111ea3, 0xd};
constexpr uint32_t N0 = 0x1ffcfffd;
constexpr uint32_t M29 = (1u << 29) - 1;

// r = a*b/2^406 mod p, limbs of a,b < 2^29 (values < 2^12 p); result limbs < 2^29, value < 2p
__device__ __forceinline__ void mont_mul(uint32_t* r, const uint32_t* a, const uint32_t* b) {
    uint32_t m[14];
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 14; ++k) {
#pragma unroll
        for (int i = 0; i <= k; ++i) acc += (uint64_t)a[i] * b[k - i];
#pragma unroll
        for (int i = 0; i < k; ++i) acc += (uint64_t)m[i] * PI[k - i];
        m[k] = ((uint32_t)acc * N0) & M29;
        acc += (uint64_t)m[k] * PI[0];
        acc >>= 29;
    }
#pragma unroll
    for (int k = 14; k < 28; ++k) {
#pragma unroll
        for (int i = k - 13; i < 14; ++i) acc += (uint64_t)a[i] * b[k - i];
#pragma unroll
        for (int i = k - 13; i < 14; ++i) acc += (uint64_t)m[i] * PI[k - i];
        r[k - 14] = (uint32_t)acc & M29;
        acc >>= 29;
    }
}
} // namespace U

template <int VAR>
__global__ void __launch_bounds__(256) fpmul_kernel(uint32_t* io, int n_limbs, int iters) {
    // io layout: limb-major SoA  io[limb * nthreads + tid]  for x then y then out
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int nt = gridDim.x * blockDim.x;
    constexpr int NL = VAR == 0 ? 12 : 14;
    uint32_t x[NL], y[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) { x[i] = io[i * nt + tid]; y[i] = io[(NL + i) * nt + tid]; }
    for (int it = 0; it < iters; ++it) {
        uint32_t z[NL];
        if constexpr (VAR == 0) S::mont_mul(z, x, y); else U::mont_mul(z, x, y);
#pragma unroll
        for (int i = 0; i < NL; ++i) { x[i] = y[i]; y[i] = z[i]; }   // fibonacci-style chain: keeps both operands live
    }
#pragma unroll
    for (int i = 0; i < NL; ++i) io[(2 * NL + i) * nt + tid] = y[i];
}

static uint64_t lcg_state = 0x9e3779b97f4a7c15ull;
static uint32_t lcg() { lcg_state = lcg_state * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(lcg_state >> 32); }

// waves: 8 per SIMD (blocks = 256 CUs x 8) as in round 1, or 2 per SIMD (the occupancy every kernel of the library runs at)
template <int KIND>
int run_rate(const char* name, int ops_per_iter, FILE* fo, int blocks = 256 * 8, int iters = ITERS * 16) {
    const int threads = 256;
    uint32_t* d;
    unsigned long long* dclk;
    const size_t nw = (size_t)blocks * threads / 64;
    CK(hipMalloc(&d, (size_t)blocks * threads * 4));
    CK(hipMalloc(&dclk, nw * 32));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    rate_kernel<KIND><<<blocks, threads>>>(d, 12345, dclk, iters);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        rate_kernel<KIND><<<blocks, threads>>>(d, 12345 + rep, dclk, iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    std::vector<unsigned long long> hc(nw * 4);
    CK(hipMemcpy(hc.data(), dclk, nw * 32, hipMemcpyDeviceToHost));
    std::vector<double> ghz(nw);
    for (size_t w = 0; w < nw; ++w) ghz[w] = (double)(hc[4 * w + 1] - hc[4 * w]) / (double)(hc[4 * w + 3] - hc[4 * w + 2]) * 0.1;
    std::sort(ghz.begin(), ghz.end());
    const double clock_ghz = ghz[nw / 2];
    double ops = (double)blocks * threads * iters * ops_per_iter;
    double rate = ops / (best * 1e-3);
    // lanes per clock and CU at the nominal 2.4 GHz and at the clock the chip held inside this kernel
    double per_clk_cu = rate / 2.4e9 / 256.0, per_clk_cu_real = rate / (clock_ghz * 1e9) / 256.0;
    printf("RATE %-28s waves/SIMD %d  %8.3f ms  %.4e lane-ops/s  %.2f lane-ops/clk/CU@2.4GHz  in-kernel clock %.3f GHz -> %.2f lane-ops/clk/CU\n",
           name, blocks / 256, best, rate, per_clk_cu, clock_ghz, per_clk_cu_real);
    if (fo) fprintf(fo, "RATE %s waves_per_simd=%d ms=%.6f lane_ops_per_s=%.6e per_clk_cu_at_2.4GHz=%.3f in_kernel_clock_GHz=%.3f per_clk_cu_at_that_clock=%.3f\n",
                    name, blocks / 256, best, rate, per_clk_cu, clock_ghz, per_clk_cu_real);
    CK(hipFree(d)); CK(hipFree(dclk));
    return 0;
}

template <int VAR>
int run_fpmul(const char* name, FILE* fo) {
    constexpr int NL = VAR == 0 ? 12 : 14;
    constexpr int BITS = VAR == 0 ? 32 : 29;
    const int blocks = 256 * 4, threads = 256, nt = blocks * threads, iters = 512;
    std::vector<uint32_t> h((size_t)3 * NL * nt);
    lcg_state = 0x1234567;
    for (int t = 0; t < nt; ++t)
        for (int v = 0; v < 2; ++v) {
            // random 380-bit value (< p), split into limbs of BITS bits
            uint32_t w[12];
            for (int i = 0; i < 12; ++i) w[i] = lcg();
            w[11] &= 0x0fffffff;
            for (int i = 0; i < NL; ++i) {
                int bit = i * BITS;
                uint64_t lo = 0;
                int wi = bit / 32, sh = bit % 32;
                if (wi < 12) lo = w[wi];
                if (wi + 1 < 12) lo |= (uint64_t)w[wi + 1] << 32;
                uint32_t limb = (uint32_t)(lo >> sh);
                if (BITS < 32) limb &= (1u << BITS) - 1;
                h[(size_t)(v * NL + i) * nt + t] = limb;
            }
        }
    uint32_t* d;
    CK(hipMalloc(&d, h.size() * 4));
    CK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    fpmul_kernel<VAR><<<blocks, threads>>>(d, NL, iters);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        fpmul_kernel<VAR><<<blocks, threads>>>(d, NL, iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    std::vector<uint32_t> out(h.size());
    CK(hipMemcpy(out.data(), d, h.size() * 4, hipMemcpyDeviceToHost));
    double rate = (double)nt * iters / (best * 1e-3);
    printf("FPMUL %-10s %8.3f ms  %.4e mont-mul/s\n", name, best, rate);
    if (fo) {
        fprintf(fo, "FPMUL %s %.6f %.6e\n", name, best, rate);
        // dump 4 lanes for offline verification: x, y, result (limbs, little-endian order)
        for (int t = 0; t < 4; ++t) {
            int lane = t * 7919 % nt;
            fprintf(fo, "CHECK %s %d %d %d", name, BITS, NL, iters);
            for (int v = 0; v < 3; ++v) { fprintf(fo, " |"); for (int i = 0; i < NL; ++i) fprintf(fo, " %08x", out[(size_t)(v * NL + i) * nt + lane]); }
            fprintf(fo, "\n");
        }
    }
    CK(hipFree(d));
    return 0;
}

int main(int argc, char** argv) {
    FILE* fo = argc > 1 ? fopen(argv[1], "w") : nullptr;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s CUs %d clock %d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
    if (fo) fprintf(fo, "DEVICE %s %d %d\n", prop.name, prop.multiProcessorCount, prop.clockRate);
    if (run_rate<5>("v_add_u32", 16, fo)) return 1;
    if (run_rate<11>("v_fma_f32", 16, fo)) return 1;
    if (run_rate<0>("v_mad_u64_u32", 16, fo)) return 1;
    if (run_rate<13>("v_mad_u64_u32_sgprcarry", 16, fo)) return 1;
    if (run_rate<12>("v_mad_u64_u32_dependent", 16, fo)) return 1;
    // the instruction the library's limb products are (round 3): 8 and 2 waves per SIMD, independent and dependent chains,
    // and the instruction mix of one fp_mul column
    if (run_rate<16>("v_mad_i64_i32", 16, fo)) return 1;
    if (run_rate<16>("v_mad_i64_i32", 16, fo, 256 * 2)) return 1;
    if (run_rate<16>("v_mad_i64_i32", 16, fo, 256 * 2, ITERS * 64)) return 1;      // the same, 4 x longer: the clock a 20-ms kernel holds
    if (run_rate<16>("v_mad_i64_i32", 16, fo, 256 * 3)) return 1;
    if (run_rate<16>("v_mad_i64_i32", 16, fo, 256 * 4)) return 1;
    if (run_rate<16>("v_mad_i64_i32", 16, fo, 256 * 1)) return 1;
    if (run_rate<17>("v_mad_i64_i32_dependent", 16, fo)) return 1;
    if (run_rate<17>("v_mad_i64_i32_dependent", 16, fo, 256 * 2)) return 1;
    if (run_rate<18>("fp_mul_column_mix(12mad+4)", 16, fo)) return 1;
    if (run_rate<18>("fp_mul_column_mix(12mad+4)", 16, fo, 256 * 2)) return 1;
    if (run_rate<0>("v_mad_u64_u32", 16, fo, 256 * 2)) return 1;
    if (run_rate<5>("v_add_u32", 16, fo, 256 * 2)) return 1;
    if (run_rate<1>("v_mul_lo_u32", 16, fo)) return 1;
    if (run_rate<2>("v_mul_hi_u32", 16, fo)) return 1;
    if (run_rate<3>("v_mad_u32_u24", 16, fo)) return 1;
    if (run_rate<14>("v_mul_u32_u24", 16, fo)) return 1;
    if (run_rate<10>("v_mul_hi_u32_u24", 16, fo)) return 1;
    if (run_rate<4>("v_add_co+v_addc_co", 16, fo)) return 1;
    if (run_rate<9>("v_add3_u32", 16, fo)) return 1;
    if (run_rate<15>("v_lshl_add_u32", 16, fo)) return 1;
    if (run_rate<8>("v_alignbit_b32", 16, fo)) return 1;
    if (run_rate<7>("v_lshrrev_b64", 16, fo)) return 1;
    if (run_rate<6>("v_fma_f64", 16, fo)) return 1;
    if (run_fpmul<0>("S12x32", fo)) return 1;
    if (run_fpmul<1>("U14x29", fo)) return 1;
    if (fo) fclose(fo);
    return 0;
}
