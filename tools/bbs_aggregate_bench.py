"""Time the aggregate BBS+ verification (c12381_bbs_plus_verify_aggregate_dev) on one GPU, for rocprofv3:
   rocprofv3 --kernel-trace --stats --output-format csv -d out -- python3 tools/bbs_aggregate_bench.py [log2 n]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tools.libsel  # noqa: E402,F401  (C12381_LIB -> capi.use_library)
from crypto12381_amd import Context  # noqa: E402
from bench import G1_GEN, G2_GEN, make_scalars  # noqa: E402


def main():
    lg = int(sys.argv[1]) if len(sys.argv) > 1 else 18
    bits = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    n = 1 << lg
    dev = torch.device("cuda:0")
    torch.cuda.init()
    ctx = Context(0)
    up = lambda a: torch.from_numpy(np.frombuffer(bytes(a), dtype=np.uint8).copy()).to(dev)
    g1 = up(G1_GEN).repeat(n).contiguous()
    pts = torch.empty(96 * n, dtype=torch.uint8, device=dev)
    ctx.g1_mul_dev(n, g1.data_ptr(), torch.from_numpy(make_scalars(1, n)).to(dev).data_ptr(), pts.data_ptr(), 96)
    ctx.sync()
    xs_h, rs_h, mm_h = (make_scalars(s, n) for s in (2, 3, 4))
    xs, rs, mm = (torch.from_numpy(a).to(dev) for a in (xs_h, rs_h, mm_h))
    rho = torch.from_numpy(make_scalars(5, n)).to(dev)
    if bits < 256:
        rho.view(n, 32)[:, : (256 - bits) // 8] = 0
    g2 = up(G2_GEN)
    gamma = make_scalars(6, 8)[7].tobytes()                     # lane 7: a random value (lanes 0..4 are the edge cases)
    w = up(ctx.g2_mul(G2_GEN, gamma, 192))
    # real signatures under the key gamma (public parameters: three points of the batch), so the verdict must be 1
    pub = pts[96:384].cpu().numpy().tobytes()
    A = up(ctx.bbs_plus_sign(pub[:96], pub[96:192], pub[192:288], gamma, xs_h.tobytes(), rs_h.tobytes(), mm_h.tobytes()))
    ok = torch.zeros(16, dtype=torch.uint8, device=dev)
    call = lambda: ctx.bbs_plus_verify_aggregate_dev(n, 1, pts[96:192].data_ptr(), g2.data_ptr(), pts[192:288].data_ptr(), pts[288:384].data_ptr(),
                                                     w.data_ptr(), A.data_ptr(), xs.data_ptr(), rs.data_ptr(), mm.data_ptr(), rho.data_ptr(), ok.data_ptr())
    okb = torch.zeros(n, dtype=torch.uint8, device=dev)
    ctx.bbs_plus_verify_dev(n, 1, pts[96:192].data_ptr(), g2.data_ptr(), pts[192:288].data_ptr(), pts[288:384].data_ptr(), w.data_ptr(), A.data_ptr(),
                            xs.data_ptr(), rs.data_ptr(), mm.data_ptr(), okb.data_ptr())
    ctx.sync()
    print(f"per-signature entry: {int((okb == 1).sum())} of {n} accepted")
    for j in torch.nonzero(okb != 1).view(-1).tolist()[:8]:
        print("   rejected lane", j, "ok =", int(okb[j]), "x =", xs_h[j].tobytes().hex(), "r =", rs_h[j].tobytes().hex(), "m =", mm_h[j].tobytes().hex())
    call(); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(3):
        call()
    ctx.sync()
    el = (time.perf_counter() - t0) / 3
    print(f"aggregate n=2^{lg} rho={bits}-bit: {el * 1e3:.2f} ms per batch, {n / el:.3e} signatures/s, verdict {int(ok[0])} (valid signatures: 1 expected)")
    xs.view(-1)[32 * (n // 2) + 31] ^= 1
    call(); ctx.sync()
    print(f"  one wrong x: verdict {int(ok[0])} (0 expected)")


if __name__ == "__main__":
    main()
