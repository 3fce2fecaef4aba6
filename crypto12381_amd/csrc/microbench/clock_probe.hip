// Bench-only (NOT part of the product library, not declared in include/c12381_hip.h): the clock the chip holds while another
// kernel runs.  ONE lane samples (s_memtime = shader cycles, s_memrealtime = 100 MHz) every `gap` sleeps of 127 x 64 cycles on a
// non-blocking stream of its own, beside the kernel under study; clock over an interval = d(memtime) / d(memrealtime) x 100 MHz
// (MI355X_MICROARCH.md, DVFS give-back).  bench.py loads lib/libc12381_probe.so for this alone, so that `roofline.issue` prices a
// kernel's instruction count at the clock of THE SAME run (round 3 took the clock from another box's probe: VERDICT r03, weak 5).
#include <hip/hip_runtime.h>

__global__ void __launch_bounds__(64, 1) c12381_clock_probe_kernel(unsigned long long* out, int n, int gap) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    for (int i = 0; i < n; ++i) {
        out[2 * i] = __builtin_amdgcn_s_memtime();
        out[2 * i + 1] = __builtin_amdgcn_s_memrealtime();
        for (int j = 0; j < gap; ++j) __builtin_amdgcn_s_sleep(127);
    }
}

// out: 2 n device words on `device`; returns 0 or the HIP error code.  The launch is asynchronous: the caller synchronizes the device.
extern "C" int c12381_probe_start(int device, unsigned long long* out, int n, int gap) {
    static hipStream_t streams[16] = {};
    if (device < 0 || device >= 16 || !out || n <= 0 || gap < 0) return -1;
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return (int)e;
    if (!streams[device]) {
        e = hipStreamCreateWithFlags(&streams[device], hipStreamNonBlocking);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(c12381_clock_probe_kernel, dim3(1), dim3(64), 0, streams[device], out, n, gap);
    return (int)hipGetLastError();
}
