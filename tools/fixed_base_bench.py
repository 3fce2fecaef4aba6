import sys, time, torch
sys.path.insert(0, '.')
import tools.libsel  # noqa: E402,F401  (C12381_LIB -> capi.use_library)
from crypto12381_amd import Context
from tools.prof_driver import G1, sc
c = Context(0); dev = torch.device('cuda', 0)
s = torch.cuda.Stream(device=dev); c.set_stream(s.cuda_stream)
n = 1 << 20
k = torch.frombuffer(bytearray(sc(9, n)), dtype=torch.uint8).to(dev)
b = torch.frombuffer(bytearray(G1), dtype=torch.uint8).to(dev)
o = torch.empty(96 * n, dtype=torch.uint8, device=dev)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    c.g1_mul_fixed_dev(n, b.data_ptr(), k.data_ptr(), o.data_ptr(), 96); c.sync()
    print("fixed-base 2^20: %.2f ms" % ((time.perf_counter() - t0) * 1e3))
