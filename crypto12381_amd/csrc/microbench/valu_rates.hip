// Calibration microbenchmark for the integer-VALU roofline of the BLS12-381 hot path
// ("the build must ship a microbenchmark ... and use the measured peak").
//
//  part 1: issue rate of the instructions a multi-precision multiply is made of
//          (v_mad_u64_u32, v_mul_lo/hi_u32, 24-bit mads, carry adds, 64-bit shifts, v_fma_f64)
//  part 2: throughput of two candidate Fp Montgomery multipliers as hipcc compiles them:
//          S  = 12 x 32-bit saturated limbs (R = 2^384)
//          U  = 14 x 29-bit unsaturated limbs (R = 2^406), one 64-bit column accumulator
//
// Build:  hipcc -O3 --offload-arch=gfx950 valu_rates.hip -o valu_rates
// Run:    ./valu_rates [out.txt]       (prints one line per measurement)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

constexpr int ITERS = 4096;

// ---------------------------------------------------------------- part 1
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int KIND>
__global__ void __launch_bounds__(256) rate_kernel(uint32_t* out, uint32_t seed) {
    uint32_t a = seed + threadIdx.x, b = seed * 3 + blockIdx.x;
    uint64_t acc[8];
    uint32_t r[8];
    double d[8];
    for (int i = 0; i < 8; ++i) { acc[i] = a + i; r[i] = b + i; d[i] = (double)(a + i); }
    double da = (double)a * 1e-9, db = (double)b * 1e-9;
    for (int it = 0; it < ITERS; ++it) {
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
        } else if constexpr (KIND == 15) { // v_lshl_add_u32
#define X(i) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(r[i]) : "v"(a));
            REP8(X) REP8(X)
#undef X
        }
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; ++i) s += (uint32_t)acc[i] + (uint32_t)(acc[i] >> 32) + r[i] + (uint32_t)d[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
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

template <int KIND>
int run_rate(const char* name, int ops_per_iter, FILE* fo) {
    const int blocks = 256 * 8, threads = 256;
    uint32_t* d;
    CK(hipMalloc(&d, (size_t)blocks * threads * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    rate_kernel<KIND><<<blocks, threads>>>(d, 12345);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        rate_kernel<KIND><<<blocks, threads>>>(d, 12345 + rep);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    double ops = (double)blocks * threads * ITERS * ops_per_iter;
    double rate = ops / (best * 1e-3);
    // lanes/clk/CU at 2.4 GHz nominal
    double per_clk_cu = rate / 2.4e9 / 256.0;
    printf("RATE %-28s %8.3f ms  %.4e lane-ops/s  %.2f lane-ops/clk/CU@2.4GHz\n", name, best, rate, per_clk_cu);
    if (fo) fprintf(fo, "RATE %s %.6f %.6e %.3f\n", name, best, rate, per_clk_cu);
    CK(hipFree(d));
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
