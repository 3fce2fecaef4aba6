// Bench-only (NOT part of the product library, not declared in include/c12381_hip.h): the clock the chip holds while another
// kernel runs.  ONE lane per XCD samples (s_memtime = shader cycles, s_memrealtime = 100 MHz) every `gap` sleeps of 127 x 64 cycles on a
// non-blocking stream of its own, beside the kernel under study; clock over an interval = d(memtime) / d(memrealtime) x 100 MHz
// (MI355X_MICROARCH.md, DVFS give-back).  bench.py loads lib/libc12381_probe.so for this alone, so that `roofline.issue` prices a
// kernel's instruction count at the clock of THE SAME run (round 3 took the clock from another box's probe: VERDICT r03, weak 5).
// Round 5: the eight XCDs are sampled separately (64 one-wavefront workgroups are launched, the first to arrive on each XCD becomes its
// sampler, the others leave at once): a box's XCDs do not hold the same clock, and one sampler on whichever XCD it landed on made the same
// build read "higher clock, slower kernel" on some boxes.
#include <hip/hip_runtime.h>

constexpr int PROBE_XCC_MAX = 16;

// out: PROBE_XCC_MAX blocks of (2 n + 2) words: [0] = claimed flag (zeroed by the caller), [1] = samples written, then n x (memtime, memrealtime)
__global__ void __launch_bounds__(64, 1) c12381_clock_probe_kernel(unsigned long long* out, int n, int gap) {
    if (threadIdx.x != 0) return;
    const unsigned xcc = __builtin_amdgcn_s_getreg(63508) & 15u;              // XCC_ID
    unsigned long long* o = out + (size_t)xcc * (2 * (size_t)n + 2);
    if (atomicCAS(o, 0ull, 1ull) != 0ull) return;                             // this XCD has its sampler
    for (int i = 0; i < n; ++i) {
        o[2 + 2 * i] = __builtin_amdgcn_s_memtime();
        o[3 + 2 * i] = __builtin_amdgcn_s_memrealtime();
        o[1] = (unsigned long long)(i + 1);
        for (int j = 0; j < gap; ++j) __builtin_amdgcn_s_sleep(127);
    }
}

// out: PROBE_XCC_MAX x (2 n + 2) device words on `device`, zeroed; returns 0 or the HIP error code.  The launch is asynchronous: the caller
// synchronizes the device.
extern "C" int c12381_probe_start(int device, unsigned long long* out, int n, int gap) {
    static hipStream_t streams[16] = {};
    if (device < 0 || device >= 16 || !out || n <= 0 || gap < 0) return -1;
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return (int)e;
    if (!streams[device]) {
        e = hipStreamCreateWithFlags(&streams[device], hipStreamNonBlocking);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(c12381_clock_probe_kernel, dim3(64), dim3(64), 0, streams[device], out, n, gap);
    return (int)hipGetLastError();
}
extern "C" int c12381_probe_xcc_max(void) { return PROBE_XCC_MAX; }
