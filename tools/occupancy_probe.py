import torch, time, sys
sys.path.insert(0, '.')
import tools.libsel  # noqa: E402,F401  (C12381_LIB -> capi.use_library)
from crypto12381_amd import Context
ctx = Context(0)
dev = torch.device('cuda', 0)
s = torch.cuda.Stream(device=dev); ctx.set_stream(s.cuda_stream)
iters = 2000
for n in (1<<14, 1<<15, 1<<16, 1<<17, 1<<18, 1<<19):
    a = torch.randint(0, 200, (n*48,), dtype=torch.uint8, device=dev)
    b = torch.randint(0, 200, (n*48,), dtype=torch.uint8, device=dev)
    o = torch.empty(n*48, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    ctx.fp_mulchain_dev(n, iters, a.data_ptr(), b.data_ptr(), o.data_ptr()); torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx.fp_mulchain_dev(n, iters, a.data_ptr(), b.data_ptr(), o.data_ptr()); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    waves_per_simd = n / 64 / 1024
    print("n=%7d waves/SIMD=%.2f  %.3f ms  %.3e mont-mul/s  cycles/mul/wave@2.1GHz=%.0f" % (n, waves_per_simd, dt*1e3, n*iters/dt, dt*2.1e9/iters/max(1,waves_per_simd)))
