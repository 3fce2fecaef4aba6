// Calibration of the FETCH_SIZE counter for THIS kernel's access pattern: every lane reads whole 176-byte records of its own
// 1408-byte table slab as eleven 16-byte loads (tab_load_g1, g1.hpp), record index data dependent — exactly what g1_mul_kernel
// does 66 times per scalar multiplication.  The bytes requested are known (lanes x lookups x 176), so
//     rocprofv3 --pmc FETCH_SIZE -- ./fetch_calib
// tells how the counter's kilobytes relate to them (MI355X_MICROARCH.md: "16-B/lane loads count half"; profiles/traffic.json
// applied that rule to a pattern it had not been calibrated on).
// Build: hipcc -O3 --offload-arch=gfx950 fetch_calib.hip -o fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

constexpr int ENT_DWORDS = 44, TAB = 8, LOOKUPS = 66;
struct alignas(16) q4 { int32_t v[4]; };

__global__ void __launch_bounds__(256, 2) gather_kernel(size_t n, const int32_t* tab, int32_t* out, uint32_t seed) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int32_t* lane_tab = tab + i * (size_t)(TAB * ENT_DWORDS);
    uint32_t s = seed ^ (uint32_t)i * 2654435761u;
    int32_t acc[4] = {0, 0, 0, 0};
#pragma unroll 1
    for (int k = 0; k < LOOKUPS; ++k) {
        s = s * 1664525u + 1013904223u;
        const int idx = (int)((s >> 13) & 7u) ^ (acc[0] & 0);           // data-dependent index
        const q4* src = reinterpret_cast<const q4*>(lane_tab + idx * ENT_DWORDS);
#pragma unroll
        for (int j = 0; j < ENT_DWORDS / 4; ++j) { const q4 t = src[j]; acc[0] += t.v[0]; acc[1] ^= t.v[1]; acc[2] += t.v[2]; acc[3] ^= t.v[3]; }
    }
    out[i] = acc[0] + acc[1] + acc[2] + acc[3];
}
// the same bytes as a plain streaming read (every lane 16 bytes, consecutive lanes consecutive addresses): the pattern the guide calibrated
__global__ void __launch_bounds__(256, 2) stream_kernel(size_t nq, const q4* src, int32_t* out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    int32_t a = 0;
    for (size_t j = i; j < nq; j += (size_t)gridDim.x * 256) { const q4 t = src[j]; a += t.v[0] ^ t.v[1] ^ t.v[2] ^ t.v[3]; }
    out[i] = a;
}

int main() {
    const size_t n = (size_t)1 << 17;
    const size_t tab_bytes = n * TAB * ENT_DWORDS * 4;
    int32_t *tab, *out;
    CK(hipMalloc(&tab, tab_bytes));
    CK(hipMalloc(&out, n * 4));
    CK(hipMemset(tab, 1, tab_bytes));
    hipLaunchKernelGGL(gather_kernel, dim3((unsigned)(n / 256)), dim3(256), 0, 0, n, tab, out, 12345u);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(stream_kernel, dim3(2048), dim3(256), 0, 0, tab_bytes / 16, reinterpret_cast<const q4*>(tab), out);
    CK(hipDeviceSynchronize());
    printf("gather_kernel: lanes %zu x lookups %d x 176 B = %.1f KB requested (slab %.1f MiB)\n", n, LOOKUPS, (double)n * LOOKUPS * 176 / 1024.0, tab_bytes / 1048576.0);
    printf("stream_kernel: %.1f KB requested\n", tab_bytes / 1024.0);
    return 0;
}
