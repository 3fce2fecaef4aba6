#!/usr/bin/env python3
"""rocprofv3 driver: one 2^22-term MSM through the C ABI (device buffers via torch)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools.libsel  # noqa: E402,F401  (C12381_LIB -> capi.use_library)
from crypto12381_amd import Context
G1 = bytes.fromhex("17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb"
                   "08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1")
dev = torch.device("cuda", 0)
ctx = Context(0)
n = 1 << 22
rng = np.random.Generator(np.random.PCG64(1))
m = 1 << 16
base = torch.from_numpy(rng.integers(0, 256, size=(m, 32), dtype=np.uint8)).to(dev)
gen = torch.from_numpy(np.frombuffer(G1, dtype=np.uint8).copy()).to(dev).repeat(m).contiguous()
p = torch.empty(m * 96, dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
ctx.g1_mul_dev(m, gen.data_ptr(), base.data_ptr(), p.data_ptr(), 96); ctx.sync()
pts = p.repeat(n // m).contiguous()
sc = torch.from_numpy(rng.integers(0, 256, size=(n, 32), dtype=np.uint8)).to(dev)
out = torch.empty(96, dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
for _ in range(2):
    ctx.g1_msm_dev(n, pts.data_ptr(), sc.data_ptr(), out.data_ptr(), 96)
ctx.sync()
