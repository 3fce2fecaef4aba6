// What does an instruction cost a wavefront when it is mixed with the 64-bit multiply-adds?  (round 3, docs/lab_notes.md 5c)
//
// valu_rates.hip times streams of ONE instruction with events: 16 instructions per loop iteration, so the loop's own scalar
// instructions (28-32 cycles per iteration) and the uneven arrival of the workgroups are part of its numbers.  This program counts
// shader cycles INSIDE the kernel (s_memtime around a loop of 128 instructions per iteration, per wavefront), records on which SIMD
// each wavefront ran and when (HW_ID, s_memrealtime), and measures instruction groups alone, interleaved one by one with
// multiply-adds, and in blocks after them, at 1 / 2 / 4 / 8 workgroups of 256 per CU.
//
// What it shows on MI355X (profiles/r03_issue_mix.txt):
//   - one wavefront alone issues a vector instruction every 4 cycles (4.25 for v_mad_i64_i32), whatever it is;
//   - a SIMD takes its instructions from the OLDEST wavefront: with two wavefronts of the same multiply-add stream the older one runs
//     at its single-wavefront speed and the younger one gets what is left — 32 multiply-adds in 132 cycles, 4.125 per instruction:
//     the pipe is full with ONE wavefront, a second adds 3 %;
//   - v_add_u32 / v_and_b32 occupy the pipe for 2 of their 4 issue cycles, so two wavefronts of pure adds do overlap (67.5 cycles per
//     16, both); 64-bit shifts and adds, DPP moves and selects with an SGPR-pair condition take 4 like the multiply-add;
//   - MIXED streams do not overlap at all: 16 x (mad, add) costs two wavefronts 259.9 cycles = 2 x 16 x 2 x 4.06 — the younger
//     wavefront's multiply-add does not fit the 2-cycle hole an add of the older one leaves, and it issues in order.  Interleaved or in
//     blocks, one by one or two by one: the same.  A kernel at two wavefronts per SIMD therefore runs at
//         (vector instructions of both wavefronts) x 4.06 cycles + the stalls both wavefronts have at the same time;
//   - ds_bpermute_b32 costs the issuing wavefront 24 cycles each (48 with two wavefronts per SIMD doing it), multiply-adds issued
//     behind it are hidden in that time;
//   - VOP2 v_cndmask_b32 reading vcc: 16 in a row issue at 4 cycles each, every further one of the same run at 20-40 cycles (32 in a row:
//     768 cycles instead of 196); the VOP3 form with an SGPR-pair condition stays at 4.
//
// Build:  hipcc -O3 --offload-arch=gfx950 issue_mix.hip -o issue_mix      Run:  ./issue_mix [out.txt]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <algorithm>
#include <vector>
#include <map>
#include <array>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

#define MAD(i) "v_mad_i64_i32 %" #i ", s[20:21], %16, %17, %" #i "\n\t"
#define ADD(i) "v_add_u32 %" #i ", %16, %" #i "\n\t"
#define AND(i) "v_and_b32 %" #i ", 0xfffffff, %" #i "\n\t"
#define SEL(i) "v_cndmask_b32 %" #i ", %16, %" #i ", vcc\n\t"
#define SH64(i) "v_ashrrev_i64 %" #i ", 1, %" #i "\n\t"
#define LA64(i) "v_lshl_add_u64 %" #i ", %" #i ", 0, %" #i "\n\t"
#define DPP(i) "v_mov_b32_dpp %" #i ", %" #i " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
#define NOP "s_nop 0\n\t"
#define SELS(i) "v_cndmask_b32_e64 %" #i ", %16, %" #i ", s[22:23]\n\t"
#define BPERM(i) "ds_bpermute_b32 %" #i ", %17, %" #i "\n\t"
#define X8(B) B B B B B B B B

// operands 0..7: 64-bit accumulators, 8..15: 32-bit registers, 16, 17: multiplicands
#define OPERANDS : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]), \
                   "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(a), "v"(b) : "s20", "s21", "vcc", "s22", "s23"

#define M8 MAD(0) MAD(1) MAD(2) MAD(3) MAD(4) MAD(5) MAD(6) MAD(7)
#define A8 ADD(8) ADD(9) ADD(10) ADD(11) ADD(12) ADD(13) ADD(14) ADD(15)
#define N8 AND(8) AND(9) AND(10) AND(11) AND(12) AND(13) AND(14) AND(15)
#define S8 SEL(8) SEL(9) SEL(10) SEL(11) SEL(12) SEL(13) SEL(14) SEL(15)
#define T8 SELS(8) SELS(9) SELS(10) SELS(11) SELS(12) SELS(13) SELS(14) SELS(15)
#define P8 BPERM(8) BPERM(9) BPERM(10) BPERM(11) BPERM(12) BPERM(13) BPERM(14) BPERM(15)
#define MT8 MAD(0) SELS(8) MAD(1) SELS(9) MAD(2) SELS(10) MAD(3) SELS(11) MAD(4) SELS(12) MAD(5) SELS(13) MAD(6) SELS(14) MAD(7) SELS(15)
#define D8 DPP(8) DPP(9) DPP(10) DPP(11) DPP(12) DPP(13) DPP(14) DPP(15)
#define MA8 MAD(0) ADD(8) MAD(1) ADD(9) MAD(2) ADD(10) MAD(3) ADD(11) MAD(4) ADD(12) MAD(5) ADD(13) MAD(6) ADD(14) MAD(7) ADD(15)
#define MAA8 MAD(0) ADD(8) ADD(9) MAD(1) ADD(10) ADD(11) MAD(2) ADD(12) ADD(13) MAD(3) ADD(14) ADD(15) MAD(4) ADD(8) ADD(9) MAD(5) ADD(10) ADD(11) MAD(6) ADD(12) ADD(13) MAD(7) ADD(14) ADD(15)
#define MMA8 MAD(0) MAD(1) ADD(8) MAD(2) MAD(3) ADD(9) MAD(4) MAD(5) ADD(10) MAD(6) MAD(7) ADD(11)
#define MN8 MAD(0) AND(8) MAD(1) AND(9) MAD(2) AND(10) MAD(3) AND(11) MAD(4) AND(12) MAD(5) AND(13) MAD(6) AND(14) MAD(7) AND(15)
#define MS8 MAD(0) SEL(8) MAD(1) SEL(9) MAD(2) SEL(10) MAD(3) SEL(11) MAD(4) SEL(12) MAD(5) SEL(13) MAD(6) SEL(14) MAD(7) SEL(15)
#define MD8 MAD(0) DPP(8) MAD(1) DPP(9) MAD(2) DPP(10) MAD(3) DPP(11) MAD(4) DPP(12) MAD(5) DPP(13) MAD(6) DPP(14) MAD(7) DPP(15)
// 64-bit simple instructions work on accumulators 4..7 while the multiply-adds use 0..3
#define M4x2 MAD(0) MAD(1) MAD(2) MAD(3) MAD(0) MAD(1) MAD(2) MAD(3)
#define H4x2 SH64(4) SH64(5) SH64(6) SH64(7) SH64(4) SH64(5) SH64(6) SH64(7)
#define L4x2 LA64(4) LA64(5) LA64(6) LA64(7) LA64(4) LA64(5) LA64(6) LA64(7)
#define MH8 MAD(0) SH64(4) MAD(1) SH64(5) MAD(2) SH64(6) MAD(3) SH64(7) MAD(0) SH64(4) MAD(1) SH64(5) MAD(2) SH64(6) MAD(3) SH64(7)
#define ML8 MAD(0) LA64(4) MAD(1) LA64(5) MAD(2) LA64(6) MAD(3) LA64(7) MAD(0) LA64(4) MAD(1) LA64(5) MAD(2) LA64(6) MAD(3) LA64(7)
#define MNOP8 MAD(0) NOP MAD(1) NOP MAD(2) NOP MAD(3) NOP MAD(4) NOP MAD(5) NOP MAD(6) NOP MAD(7) NOP

template <int KIND>
__global__ void __launch_bounds__(256) mix_kernel(uint32_t* out, uint32_t seed, unsigned long long* clk, int iters) {
    uint32_t a = seed + threadIdx.x, b = seed * 3 + blockIdx.x;
    uint64_t acc[8];
    uint32_t r[8];
    for (int i = 0; i < 8; ++i) { acc[i] = a + i; r[i] = b + i; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if constexpr (KIND == 0) asm volatile(X8(M8 M8) OPERANDS);                    // 16 mads
        else if constexpr (KIND == 1) asm volatile(X8(A8 A8) OPERANDS);               // 16 adds
        else if constexpr (KIND == 2) asm volatile(X8(MA8 MA8) OPERANDS);             // 16 mads + 16 adds, one by one
        else if constexpr (KIND == 3) asm volatile(X8(M8 M8 A8 A8) OPERANDS);         // 16 mads, then 16 adds
        else if constexpr (KIND == 4) asm volatile(X8(MAA8 MAA8) OPERANDS);           // 16 mads + 32 adds, interleaved 1:2
        else if constexpr (KIND == 5) asm volatile(X8(M8 M8 A8 A8 A8 A8) OPERANDS);   // 16 mads, then 32 adds
        else if constexpr (KIND == 6) asm volatile(X8(MMA8 MMA8) OPERANDS);           // 16 mads + 8 adds, interleaved 2:1
        else if constexpr (KIND == 7) asm volatile(X8(M8 M8 A8) OPERANDS);            // 16 mads, then 8 adds
        else if constexpr (KIND == 8) asm volatile(X8(MN8 MN8) OPERANDS);             // 16 mads + 16 masks (32-bit literal), one by one
        else if constexpr (KIND == 9) asm volatile(X8(M8 M8 N8 N8) OPERANDS);
        else if constexpr (KIND == 10) asm volatile(X8(MS8 MS8) OPERANDS);            // selects
        else if constexpr (KIND == 11) asm volatile(X8(M8 M8 S8 S8) OPERANDS);
        else if constexpr (KIND == 12) asm volatile(X8(MD8 MD8) OPERANDS);            // DPP moves
        else if constexpr (KIND == 13) asm volatile(X8(M8 M8 D8 D8) OPERANDS);
        else if constexpr (KIND == 14) asm volatile(X8(MH8 MH8) OPERANDS);            // 64-bit shifts
        else if constexpr (KIND == 15) asm volatile(X8(M4x2 M4x2 H4x2 H4x2) OPERANDS);
        else if constexpr (KIND == 16) asm volatile(X8(ML8 ML8) OPERANDS);            // v_lshl_add_u64
        else if constexpr (KIND == 17) asm volatile(X8(M4x2 M4x2 L4x2 L4x2) OPERANDS);
        else if constexpr (KIND == 18) asm volatile(X8(N8 N8) OPERANDS);              // 16 masks alone
        else if constexpr (KIND == 19) asm volatile(X8(H4x2 H4x2) OPERANDS);          // 16 64-bit shifts alone
        else if constexpr (KIND == 20) asm volatile(X8(L4x2 L4x2) OPERANDS);          // 16 v_lshl_add_u64 alone
        else if constexpr (KIND == 21) asm volatile(X8(D8 D8) OPERANDS);              // 16 DPP moves alone
        else if constexpr (KIND == 22) asm volatile(X8(MNOP8 MNOP8) OPERANDS);        // 16 mads with an s_nop after each
        else if constexpr (KIND == 23) asm volatile(X8(S8 S8) OPERANDS);              // 16 selects (VOP2, vcc) alone
        else if constexpr (KIND == 24) asm volatile(X8(T8 T8) OPERANDS);              // 16 selects (VOP3, SGPR pair) alone
        else if constexpr (KIND == 25) asm volatile(X8(MT8 MT8) OPERANDS);            // mads and SGPR-pair selects one by one
        else if constexpr (KIND == 26) asm volatile(X8(M8 M8 T8 T8) OPERANDS);
        else if constexpr (KIND == 27) asm volatile(X8(P8 P8 "s_waitcnt lgkmcnt(0)\n\t") OPERANDS);   // 16 ds_bpermute + one wait
        else if constexpr (KIND == 28) asm volatile(X8(M8 M8 P8 P8 "s_waitcnt lgkmcnt(0)\n\t") OPERANDS);
        else if constexpr (KIND == 30) asm volatile(X8(M8 M8 S8 S8 S8 S8) OPERANDS);                     // 16 mads, then a run of 32 vcc selects
        else if constexpr (KIND == 31) asm volatile(X8(M8 M8 S8 S8 S8 S8 S8 S8 S8 S8) OPERANDS);         // ... of 64
        else if constexpr (KIND == 32) asm volatile(X8(M8 M8 S8 S8 S8 S8 S8 S8 S8 S8 S8 S8 S8 S8 S8 S8 S8 S8) OPERANDS);   // ... of 128
        else if constexpr (KIND == 33) asm volatile(X8(M8 M8 T8 T8 T8 T8 T8 T8 T8 T8) OPERANDS);         // 16 mads, then 64 SGPR-pair selects
        else if constexpr (KIND == 34) asm volatile(X8(M8 M8 "v_cmp_gt_u32 vcc, %16, %17\n\t" S8 S8 S8 S8 S8 S8 S8 S8) OPERANDS);   // vcc written by a compare in front of the run, as compiled code does
        else if constexpr (KIND == 29) asm volatile(X8(P8 P8 M8 M8 "s_waitcnt lgkmcnt(0)\n\t") OPERANDS);   // the permutes issued first, the wait after the mads
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime();
    uint32_t s = 0;
    for (int i = 0; i < 8; ++i) s += (uint32_t)acc[i] + (uint32_t)(acc[i] >> 32) + r[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        const size_t w = (size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
        clk[6 * w] = t0; clk[6 * w + 1] = t1; clk[6 * w + 2] = w0; clk[6 * w + 3] = w1;
        clk[6 * w + 4] = __builtin_amdgcn_s_getreg((31 << 11) | 4);         // HW_REG_HW_ID: wave, SIMD, CU, SH, SE
        clk[6 * w + 5] = __builtin_amdgcn_s_getreg((31 << 11) | 20);        // HW_REG_XCC_ID
    }
}

template <int KIND>
int run(const char* name, int mads, int others, int waves_per_simd, FILE* fo) {
    const int threads = 256, blocks = 256 * waves_per_simd, iters = 1 << 13;          // x 8 copies of the body per iteration
    uint32_t* d;
    unsigned long long* dclk;
    const size_t nw = (size_t)blocks * threads / 64;
    CK(hipMalloc(&d, (size_t)blocks * threads * 4));
    CK(hipMalloc(&dclk, nw * 48));
    mix_kernel<KIND><<<blocks, threads>>>(d, 12345, dclk, iters);
    CK(hipDeviceSynchronize());
    mix_kernel<KIND><<<blocks, threads>>>(d, 54321, dclk, iters);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> hc(nw * 6);
    CK(hipMemcpy(hc.data(), dclk, nw * 48, hipMemcpyDeviceToHost));
    // per wavefront: shader cycles of the loop (s_memtime counts at the shader clock) — the median over all wavefronts
    std::vector<double> cyc(nw), ghz(nw);
    for (size_t w = 0; w < nw; ++w) {
        cyc[w] = (double)(hc[6 * w + 1] - hc[6 * w]) / iters / 8.0;
        ghz[w] = (double)(hc[6 * w + 1] - hc[6 * w]) / (double)(hc[6 * w + 3] - hc[6 * w + 2]) * 0.1;
    }
    std::sort(cyc.begin(), cyc.end()); std::sort(ghz.begin(), ghz.end());
    const double c = cyc[nw / 2];
    // how many wavefronts really shared a SIMD: per SIMD (XCC, SE, SH, CU, SIMD of HW_ID), the sum of the wavefronts' loop times over
    // the span from the first start to the last end on that SIMD (s_memrealtime, 100 MHz, one counter for the chip)
    std::map<unsigned long long, std::array<double, 3>> simd;       // key -> {sum of durations, first start, last end}
    unsigned long long first = ~0ull, last = 0;
    for (size_t w = 0; w < nw; ++w) {
        const unsigned long long hw = hc[6 * w + 4], key = ((hc[6 * w + 5] & 15) << 32) | (hw & 0xfff0);      // drop the wave slot bits
        const double b = (double)hc[6 * w + 2], e = (double)hc[6 * w + 3];
        auto it = simd.find(key);
        if (it == simd.end()) simd[key] = {e - b, b, e};
        else { it->second[0] += e - b; it->second[1] = std::min(it->second[1], b); it->second[2] = std::max(it->second[2], e); }
        first = std::min(first, hc[6 * w + 2]); last = std::max(last, hc[6 * w + 3]);
    }
    std::vector<double> conc;
    for (auto& kv : simd) conc.push_back(kv.second[0] / (kv.second[2] - kv.second[1]));
    std::sort(conc.begin(), conc.end());
    const double span_ms = (double)(last - first) * 1e-5, wave_ms = c * iters * 8.0 / (ghz[nw / 2] * 1e6);
    char line[512];
    snprintf(line, sizeof line, "MIX %-44s waves/SIMD %d  mads %2d others %2d  wave-cycles per group %7.2f  per mad %6.2f  clock %.3f GHz  SIMDs %4zu  resident per SIMD min %.2f median %.2f max %.2f  wave %.2f ms of %.2f ms",
             name, waves_per_simd, mads, others, c, mads ? c / mads : 0.0, ghz[nw / 2], simd.size(), conc.front(), conc[conc.size() / 2], conc.back(), wave_ms, span_ms);
    printf("%s\n", line);
    if (fo) fprintf(fo, "%s\n", line);
    CK(hipFree(d)); CK(hipFree(dclk));
    return 0;
}

int main(int argc, char** argv) {
    FILE* fo = argc > 1 ? fopen(argv[1], "w") : nullptr;
    for (int w : {2, 1}) {
        if (run<0>("16 mad", 16, 0, w, fo)) return 1;
        if (run<22>("16 x (mad, s_nop)", 16, 0, w, fo)) return 1;
        if (run<1>("16 add", 0, 16, w, fo)) return 1;
        if (run<18>("16 and-literal", 0, 16, w, fo)) return 1;
        if (run<23>("16 cndmask", 0, 16, w, fo)) return 1;
        if (run<19>("16 ashr64", 0, 16, w, fo)) return 1;
        if (run<20>("16 lshl_add_u64", 0, 16, w, fo)) return 1;
        if (run<21>("16 mov_dpp", 0, 16, w, fo)) return 1;
        if (run<2>("16 x (mad, add)", 16, 16, w, fo)) return 1;
        if (run<3>("16 mad then 16 add", 16, 16, w, fo)) return 1;
        if (run<4>("16 x (mad, add, add)", 16, 32, w, fo)) return 1;
        if (run<5>("16 mad then 32 add", 16, 32, w, fo)) return 1;
        if (run<6>("8 x (mad, mad, add)", 16, 8, w, fo)) return 1;
        if (run<7>("16 mad then 8 add", 16, 8, w, fo)) return 1;
        if (run<8>("16 x (mad, and-literal)", 16, 16, w, fo)) return 1;
        if (run<9>("16 mad then 16 and-literal", 16, 16, w, fo)) return 1;
        if (run<10>("16 x (mad, cndmask)", 16, 16, w, fo)) return 1;
        if (run<11>("16 mad then 16 cndmask", 16, 16, w, fo)) return 1;
        if (run<12>("16 x (mad, mov_dpp)", 16, 16, w, fo)) return 1;
        if (run<13>("16 mad then 16 mov_dpp", 16, 16, w, fo)) return 1;
        if (run<14>("16 x (mad, ashr64)", 16, 16, w, fo)) return 1;
        if (run<15>("16 mad then 16 ashr64", 16, 16, w, fo)) return 1;
        if (run<16>("16 x (mad, lshl_add_u64)", 16, 16, w, fo)) return 1;
        if (run<17>("16 mad then 16 lshl_add_u64", 16, 16, w, fo)) return 1;
        if (run<24>("16 cndmask (SGPR pair)", 0, 16, w, fo)) return 1;
        if (run<25>("16 x (mad, cndmask SGPR pair)", 16, 16, w, fo)) return 1;
        if (run<26>("16 mad then 16 cndmask SGPR pair", 16, 16, w, fo)) return 1;
        if (run<27>("16 ds_bpermute + wait", 0, 16, w, fo)) return 1;
        if (run<28>("16 mad, 16 ds_bpermute, wait", 16, 16, w, fo)) return 1;
        if (run<29>("16 ds_bpermute, 16 mad, wait", 16, 16, w, fo)) return 1;
        if (run<30>("16 mad then 32 cndmask vcc", 16, 32, w, fo)) return 1;
        if (run<31>("16 mad then 64 cndmask vcc", 16, 64, w, fo)) return 1;
        if (run<32>("16 mad then 128 cndmask vcc", 16, 128, w, fo)) return 1;
        if (run<33>("16 mad then 64 cndmask SGPR pair", 16, 64, w, fo)) return 1;
        if (run<34>("16 mad, v_cmp, 64 cndmask vcc", 16, 65, w, fo)) return 1;
    }
    if (fo) fclose(fo);
    return 0;
}
