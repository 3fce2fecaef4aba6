"""Time c12381_g1_msm_dev for several sizes in ONE process (window width from C12381_MSM_C if set):
   for c in 0 10 12 14 16; do C12381_MSM_C=$c python3 tools/msm_sweep.py 14 16 18 20 22; done"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tools.libsel  # noqa: E402,F401  (C12381_LIB -> capi.use_library)
from crypto12381_amd import Context  # noqa: E402
from bench import G1_GEN, make_scalars  # noqa: E402


def main():
    lgs = [int(a) for a in sys.argv[1:]] or [18]
    nmax = 1 << max(lgs)
    dev = torch.device("cuda:0")
    torch.cuda.init()
    ctx = Context(0)
    g1 = torch.from_numpy(np.frombuffer(G1_GEN, dtype=np.uint8).copy()).to(dev).repeat(nmax).contiguous()
    pts = torch.empty(96 * nmax, dtype=torch.uint8, device=dev)
    ctx.g1_mul_dev(nmax, g1.data_ptr(), torch.from_numpy(make_scalars(1, nmax)).to(dev).data_ptr(), pts.data_ptr(), 96)
    ctx.sync()
    del g1
    sc = torch.from_numpy(make_scalars(2, nmax)).to(dev)
    out = torch.empty(96, dtype=torch.uint8, device=dev)
    res = []
    for lg in lgs:
        n = 1 << lg
        ctx.g1_msm_dev(n, pts.data_ptr(), sc.data_ptr(), out.data_ptr(), 96)
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(3):
            ctx.g1_msm_dev(n, pts.data_ptr(), sc.data_ptr(), out.data_ptr(), 96)
        ctx.sync()
        res.append(f"2^{lg}: {(time.perf_counter() - t0) / 3 * 1e3:7.2f} ms")
    print(f"C12381_MSM_C={os.environ.get('C12381_MSM_C', '-'):>2}  " + "   ".join(res), flush=True)


if __name__ == "__main__":
    main()
