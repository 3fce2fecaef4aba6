import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import tools.libsel  # noqa: E402,F401  (C12381_LIB -> capi.use_library)
from crypto12381_amd import Context
from tools.prof_driver import G1, G2, sc
c = Context(0)
n = 1 << 16
p = c.g1_mul(G1 * 1024, sc(3, 1024), 96) * (n // 1024)
q = c.g2_mul(G2, sc(4, 1)[:32], 192)
dev = torch.device("cuda", 0)
dp = torch.frombuffer(bytearray(p), dtype=torch.uint8).to(dev); dq = torch.frombuffer(bytearray(q), dtype=torch.uint8).to(dev)
out = torch.empty(576 * n, dtype=torch.uint8, device=dev)
for _ in range(2):
    c.pair_fixed_g2_dev(n, dp.data_ptr(), dq.data_ptr(), out.data_ptr()); c.sync()
t0 = time.perf_counter()
for _ in range(3):
    c.pair_fixed_g2_dev(n, dp.data_ptr(), dq.data_ptr(), out.data_ptr())
c.sync()
print("pair_fixed_g2 2^16: %.2f ms" % ((time.perf_counter() - t0) / 3 * 1e3))
ref = c.pair(p[:96 * 64], q * 64)
print("equal to generic pairing:", out[:576 * 64].cpu().numpy().tobytes() == ref)
