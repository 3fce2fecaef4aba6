"""Worst-case scalar distributions through c12381_g1_msm_dev at full size: time, and the result against k * (sum of the points)."""
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
    lg = int(sys.argv[1]) if len(sys.argv) > 1 else 22
    n = 1 << lg
    dev = torch.device("cuda:0")
    torch.cuda.init()
    ctx = Context(0)
    g1 = torch.from_numpy(np.frombuffer(G1_GEN, dtype=np.uint8).copy()).to(dev).repeat(n).contiguous()
    pts = torch.empty(96 * n, dtype=torch.uint8, device=dev)
    ctx.g1_mul_dev(n, g1.data_ptr(), torch.from_numpy(make_scalars(11, n)).to(dev).data_ptr(), pts.data_ptr(), 96)
    ctx.sync()
    del g1
    out = torch.empty(96, dtype=torch.uint8, device=dev)

    def run(name, sc):
        sc = sc.contiguous()
        ctx.g1_msm_dev(n, pts.data_ptr(), sc.data_ptr(), out.data_ptr(), 96); ctx.sync()
        t0 = time.perf_counter()
        ctx.g1_msm_dev(n, pts.data_ptr(), sc.data_ptr(), out.data_ptr(), 96); ctx.sync()
        print(f"{name:<28} {(time.perf_counter() - t0) * 1e3:8.2f} ms", flush=True)
        return bytes(out.cpu().numpy())

    uni = torch.from_numpy(make_scalars(12, n)).to(dev)
    run("uniform", uni)
    one = torch.zeros(n, 32, dtype=torch.uint8, device=dev); one[:, 31] = 1
    total = run("all ones", one)
    k = make_scalars(13, 8)[7]
    eq = torch.from_numpy(np.tile(k, (n, 1))).to(dev)
    got = run("all equal (255-bit)", eq)
    exp = ctx.g1_mul(total, k.tobytes(), 96)
    print("all equal == k * sum:", got == exp)
    small = uni.clone(); small[:, :24] = 0
    run("64-bit scalars", small)
    r128 = uni.clone(); r128[:, :16] = 0
    run("128-bit scalars", r128)
    two = eq.clone(); two[::2] = uni[::2]
    run("half equal, half uniform", two)


if __name__ == "__main__":
    main()
