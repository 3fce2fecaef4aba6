#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X-native BLS12-381 backend.

Metric (BASELINE.json): G1 scalar-muls/s per MI355X on a batch of 2^20 random (point, scalar)
pairs — BASELINE.json configs[1] — bit-exact vs the CPU path.  A "step" is one pass of the hot
path (c12381_g1_mul_batch_dev: scalar-mul kernel + inversion/encode kernel) over one batch whose
inputs are already resident in HBM.  With --gpus N every rank runs its own 2^20 batch on its own
GPU (independent units, no data-path collective): weak scaling.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--log2-batch 20]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line.  `roofline` follows the contract (bound hbm: algorithmic bytes / kernel
time against 8 TB/s); because this path is integer-VALU bound the line also carries
`valu_roofline`: algorithmic 32x32 multiply-adds per launch / kernel time against the v_mad_u64_u32
issue rate measured on MI355X by csrc/microbench/valu_rates.hip (profiles/r01_valu_rates.txt).
`cpu_baseline` times the compiled reference (oracle/_ref) — or our C port when it is absent — on the
host cores, rank 0, N=1 only.  Only this leg and the sampled parity check touch oracle/.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

R_ORDER = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
G1_GEN = bytes.fromhex(
    "17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb"
    "08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1")

# algorithmic work per G1 scalar-mul: reference operation counts
MAC32_PER_G1_MUL = 579_456
BYTES_PER_G1_MUL = 224           # 96 in + 32 scalar + 96 out (canonical affine)
HBM_PEAK_GBS = 8000.0
VALU_PEAK_MAC32 = 3.10e13        # measured v_mad_u64_u32 lane-ops/s, profiles/r01_valu_rates.txt


def make_scalars(seed: int, n: int) -> np.ndarray:
    """n x 32 big-endian scalars, uniform 256-bit values (the path reduces mod r), fixed edge lanes."""
    rng = np.random.Generator(np.random.PCG64(seed))
    sc = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    edges = [0, 1, R_ORDER - 1, R_ORDER, (1 << 256) - 1]
    for j, k in enumerate(edges):
        if j < n:
            sc[j] = np.frombuffer(k.to_bytes(32, "big"), dtype=np.uint8)
    return sc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--log2-batch", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        dist.init_process_group(backend="nccl", device_id=dev)

    from crypto12381_amd import Context
    ctx = Context(local_rank)
    stream = torch.cuda.Stream(device=dev)          # the library's kernels run on this torch-owned HIP stream
    ctx.set_stream(stream.cuda_stream)

    n = 1 << args.log2_batch
    # ---- synthetic inputs, resident in HBM before the timed region
    base_sc = torch.from_numpy(make_scalars(1000 + rank, n)).to(dev)
    sc = torch.from_numpy(make_scalars(2000 + rank, n)).to(dev)
    gen = torch.from_numpy(np.frombuffer(G1_GEN, dtype=np.uint8).copy()).to(dev).repeat(n).contiguous()
    pts = torch.empty(n * 96, dtype=torch.uint8, device=dev)
    out = torch.empty(n * 96, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize(dev)
    ctx.g1_mul_dev(n, gen.data_ptr(), base_sc.data_ptr(), pts.data_ptr(), 96)      # P_i = G^{s_i} (untimed)
    ctx.sync()
    # lanes 0 and 1 of base_sc are 0 and 1: P_0 = infinity, P_1 = G — edge inputs stay in the batch
    del gen

    def step():
        ctx.g1_mul_dev(n, pts.data_ptr(), sc.data_ptr(), out.data_ptr(), 96)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    if dist:
        dist.barrier()
    ctx.profile(True)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize(dev)
    if dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    mul_ms, mul_launches = ctx.profile_read(0)
    fin_ms, fin_launches = ctx.profile_read(1)
    ctx.profile(False)
    if ctx.sync() != 0:
        raise SystemExit("bench: invalid input point reported by the kernels")
    if dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- parity (outside the timed region): sampled lanes vs the CPU oracle, all edge lanes included
    from oracle.bindings import Oracle, have_reference
    kind = "reference" if have_reference() else "port"
    orc = Oracle(kind)
    idx = list(range(8)) + [int(x) for x in np.random.Generator(np.random.PCG64(7)).integers(0, n, size=56)]
    pts_h, sc_h, out_h = pts.cpu().numpy().reshape(n, 96), sc.cpu().numpy().reshape(n, 32), out.cpu().numpy().reshape(n, 96)
    exp = orc.g1_mul(pts_h[idx].tobytes(), sc_h[idx].tobytes(), 96, 8)
    parity_ok = exp == out_h[idx].tobytes()
    if not parity_ok:
        raise SystemExit("bench: GPU results differ from the CPU oracle — number withheld")

    result = None
    if rank == 0:
        value = world * n * args.steps / elapsed
        launches_per_step = mul_launches / max(args.steps, 1)
        units_per_launch = n / max(launches_per_step, 1)
        avg_launch_s = (mul_ms / max(mul_launches, 1)) * 1e-3
        hbm_achieved = BYTES_PER_G1_MUL * units_per_launch / avg_launch_s / 1e9
        valu_achieved = MAC32_PER_G1_MUL * units_per_launch / avg_launch_s
        traffic = None
        tr_path = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tr_path):
            try:
                traffic = json.load(open(tr_path)).get("g1_mul_kernel_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        result = {
            "metric": "G1 scalar-muls/s per MI355X (batch 2^%d per GPU), bit-exact vs CPU" % args.log2_batch,
            "value": value, "unit": "scalar-muls/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int64 accumulate over 14x28-bit signed limbs", "data": "synthetic",
            "config": {"workload": "configs[1]: batch of 2^%d random G1 scalar-muls (96-B affine in, 32-B scalar, 96-B affine out) per GPU"
                                   % args.log2_batch, "batch_per_gpu": n, "parallelism": "independent shards x%d" % world},
            "parity": {"checked_lanes": len(idx), "oracle": kind, "bit_exact": parity_ok},
            "roofline": {"bound": "hbm", "achieved": hbm_achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": hbm_achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "g1_mul_kernel", "avg_launch_ms": avg_launch_s * 1e3, "launches": int(mul_launches),
                         "units_per_launch": units_per_launch,
                         "note": "integer-VALU-bound path: see valu_roofline for the binding resource"},
            "valu_roofline": {"bound": "int-valu", "achieved": valu_achieved / 1e9, "peak": VALU_PEAK_MAC32 / 1e9, "unit": "GMAC32/s",
                              "frac": valu_achieved / VALU_PEAK_MAC32,
                              "algorithmic_mac32_per_unit": MAC32_PER_G1_MUL,
                              "finish_kernel_ms_per_step": fin_ms / max(args.steps, 1)},
        }
        # ---- CPU baseline: same workload, bounded sample, host cores of this box (N=1 only)
        if world == 1 and not args.no_cpu_baseline:
            cores = min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 16)   # the GPU box gives one GPU a 16-CPU share
            sample = min(n, 1 << 16)
            sp, ss = pts_h[:sample].tobytes(), sc_h[:sample].tobytes()
            t1 = time.perf_counter()
            cpu_out = orc.g1_mul(sp, ss, 96, cores)
            cpu_s = time.perf_counter() - t1
            t2 = time.perf_counter()
            orc.g1_mul(sp[:96 * 2048], ss[:32 * 2048], 96, 1)
            cpu1_s = time.perf_counter() - t2
            if cpu_out != out_h[:sample].tobytes():
                raise SystemExit("bench: CPU baseline output differs from the GPU output")
            result["cpu_baseline"] = {"value": sample / cpu_s, "unit": "scalar-muls/s", "cores": cores, "kind": kind,
                                      "sample": "first %d lanes of the same batch, %d threads; full compare with GPU output bit-exact"
                                                % (sample, cores),
                                      "single_thread_value": 2048 / cpu1_s}
        print(json.dumps(result), flush=True)
    ctx.close()
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
