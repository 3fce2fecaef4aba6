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
MAC32_PER_PAIRING = 4_275_240    # Miller loop + final exponentiation, reference operation counts
BYTES_PER_PAIRING = 864          # 96 + 192 in, 576 out
G2_GEN = bytes.fromhex(
    "13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e"
    "024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8"
    "0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be"
    "0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801")
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
    ap.add_argument("--log2-pairings", type=int, default=16)
    ap.add_argument("--pairings", type=int, default=0, help="exact pairing batch size (overrides --log2-pairings; experiments only)")
    ap.add_argument("--no-pairing", action="store_true", help="skip the secondary (pairings/s) measurement")
    ap.add_argument("--all-configs", action="store_true",
                    help="also time BASELINE configs[3] (MSM n=2^22 per GPU) and configs[4] (2^18 BBS+ verifications per GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the product path has no CPU fallback")
    ndev = torch.cuda.device_count()
    dev_index = local_rank % max(ndev, 1)          # one rank per GPU on the 8-GPU node; wraps only in single-GPU rehearsals
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    # RCCL ("nccl") over xGMI on the GPU node; C12381_BENCH_BACKEND=gloo lets the N>1 path be rehearsed on one GPU
    backend = os.environ.get("C12381_BENCH_BACKEND", "nccl")
    red_dev = dev if backend == "nccl" else torch.device("cpu")
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)

    from crypto12381_amd import Context
    ctx = Context(dev_index)
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
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- second half of the metric: ate pairings/s on a batch of 2^16 (BASELINE configs[2]), same protocol
    pair = None
    if not args.no_pairing:
        npair = args.pairings if args.pairings > 0 else (1 << args.log2_pairings)
        t_sc = torch.from_numpy(make_scalars(3000 + rank, npair)).to(dev)
        g2gen = torch.from_numpy(np.frombuffer(G2_GEN, dtype=np.uint8).copy()).to(dev).repeat(npair).contiguous()
        q2 = torch.empty(npair * 192, dtype=torch.uint8, device=dev)
        gt = torch.empty(npair * 576, dtype=torch.uint8, device=dev)
        p1 = pts[: npair * 96] if npair <= n else pts.repeat((npair + n - 1) // n)[: npair * 96].contiguous()
        torch.cuda.synchronize(dev)
        ctx.g2_mul_dev(npair, g2gen.data_ptr(), t_sc.data_ptr(), q2.data_ptr(), 192)   # Q_i = G2^{t_i} (untimed)
        ctx.sync()
        del g2gen
        psteps = max(1, args.steps)
        for _ in range(max(1, args.warmup)):
            ctx.pair_dev(npair, p1.data_ptr(), q2.data_ptr(), gt.data_ptr())
        torch.cuda.synchronize(dev)
        if dist:
            dist.barrier()
        ctx.profile(True)
        torch.cuda.synchronize(dev)
        tp0 = time.perf_counter()
        for _ in range(psteps):
            ctx.pair_dev(npair, p1.data_ptr(), q2.data_ptr(), gt.data_ptr())
        torch.cuda.synchronize(dev)
        if dist:
            dist.barrier()
        pel = time.perf_counter() - tp0
        pk_ms, pk_launches = ctx.profile_read(3)
        ctx.profile(False)
        if dist:
            t = torch.tensor([pel], dtype=torch.float64, device=red_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            pel = float(t.item())
        pair = {"npair": npair, "steps": psteps, "elapsed": pel, "kernel_ms": pk_ms, "launches": pk_launches,
                "p1": p1, "q2": q2, "gt": gt}

    # ---- optional: configs[3] (MSM) and configs[4] (BBS+ batch verification), reported as extra objects
    extras = {}
    if args.all_configs:
        def timed_steps(fn, steps):
            fn()
            torch.cuda.synchronize(dev)
            if dist:
                dist.barrier()
            t_a = time.perf_counter()
            for _ in range(steps):
                fn()
            torch.cuda.synchronize(dev)
            if dist:
                dist.barrier()
            el = time.perf_counter() - t_a
            if dist:
                tt = torch.tensor([el], dtype=torch.float64, device=red_dev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                el = float(tt.item())
            return el
        # MSM: 2^22 terms per GPU = 4 x the 2^20 point batch (the combine across GPUs is 96 B per rank)
        nm = 1 << 22
        reps = nm // n
        mp_ = pts.repeat(reps).contiguous() if reps > 1 else pts
        ms_ = torch.from_numpy(make_scalars(4000 + rank, nm)).to(dev)
        mo_ = torch.empty(96, dtype=torch.uint8, device=dev)
        el = timed_steps(lambda: ctx.g1_msm_dev(nm, mp_.data_ptr(), ms_.data_ptr(), mo_.data_ptr(), 96), 2)
        extras["msm"] = {"metric": "G1 MSM terms/s (n = 2^22 per GPU, local part; cross-GPU combine = all-gather of 96 B)",
                         "value": world * nm * 2 / el, "unit": "terms/s", "ms_per_msm": el / 2 * 1e3}
        del mp_, ms_
        # BBS+: 2^18 signatures, 1 message block (as the reference's example message), random (mostly invalid) signatures:
        # the verification cost does not depend on validity; parity of this entry point is covered by tests/test_gpu_bbs.py
        nb = 1 << 18
        xs = torch.from_numpy(make_scalars(5000 + rank, nb)).to(dev)
        rs = torch.from_numpy(make_scalars(5001 + rank, nb)).to(dev)
        mm = torch.from_numpy(make_scalars(5002 + rank, nb)).to(dev)
        okb = torch.empty(nb, dtype=torch.uint8, device=dev)
        pub_g1, pub_h0, pub_h = pts[96:192], pts[192:288], pts[288:384]
        g2d = torch.from_numpy(np.frombuffer(G2_GEN, dtype=np.uint8).copy()).to(dev)
        wd = torch.empty(192, dtype=torch.uint8, device=dev)
        ctx.g2_mul_dev(1, g2d.data_ptr(), base_sc[64:96].data_ptr(), wd.data_ptr(), 192)
        a_pts = pts[: nb * 96]
        el = timed_steps(lambda: ctx.bbs_plus_verify_dev(nb, 1, pub_g1.data_ptr(), g2d.data_ptr(), pub_h0.data_ptr(), pub_h.data_ptr(),
                                                         wd.data_ptr(), a_pts.data_ptr(), xs.data_ptr(), rs.data_ptr(), mm.data_ptr(),
                                                         okb.data_ptr()), 2)
        extras["bbs_plus"] = {"metric": "BBS+ signature verifications/s (2^18 per GPU, 1 message block)", "value": world * nb * 2 / el,
                              "unit": "verifications/s", "ms_per_batch": el / 2 * 1e3}
        ctx.sync()
        # optional aggregate mode (one verdict per batch from a random linear combination; not a reference mode)
        rho = torch.from_numpy(make_scalars(5003 + rank, nb)).to(dev)
        rho.view(nb, 32)[:, :16] = 0                                       # 128-bit coefficients
        ok1 = torch.empty(16, dtype=torch.uint8, device=dev)
        el = timed_steps(lambda: ctx.bbs_plus_verify_aggregate_dev(nb, 1, pub_g1.data_ptr(), g2d.data_ptr(), pub_h0.data_ptr(), pub_h.data_ptr(),
                                                                   wd.data_ptr(), a_pts.data_ptr(), xs.data_ptr(), rs.data_ptr(), mm.data_ptr(),
                                                                   rho.data_ptr(), ok1.data_ptr()), 2)
        extras["bbs_plus_aggregate"] = {"metric": "BBS+ signatures/s covered by ONE aggregate verdict (2^18 per GPU, 1 message block, 128-bit coefficients)",
                                        "value": world * nb * 2 / el, "unit": "signatures/s", "ms_per_batch": el / 2 * 1e3}
        ctx.sync()
        # hash-to-G1 from 2^20 digests; 2^20 scalar-field inversions; inner product of 2^22 pairs
        nh = 1 << 20
        dg = torch.from_numpy(np.frombuffer(make_scalars(6000 + rank, 2 * nh).tobytes(), dtype=np.uint8).copy()).to(dev)
        ho = torch.empty(96 * nh, dtype=torch.uint8, device=dev)
        el = timed_steps(lambda: ctx.g1_from_hash_dev(nh, dg.data_ptr(), ho.data_ptr(), 96), 2)
        extras["hash_to_g1"] = {"metric": "hash-to-G1 points/s (2^20 SHA3-512 digests per GPU -> affine G1)", "value": world * nh * 2 / el,
                                "unit": "points/s", "ms_per_batch": el / 2 * 1e3}
        # g^x_i with ONE base for the batch (the reference's most common call shape): fixed-base tables
        fo = torch.empty(96 * nh, dtype=torch.uint8, device=dev)
        el = timed_steps(lambda: ctx.g1_mul_fixed_dev(nh, pts[96:192].data_ptr(), sc.data_ptr(), fo.data_ptr(), 96), 2)
        extras["g1_fixed_base"] = {"metric": "G1 scalar-muls/s with one base for the batch (2^20 per GPU, table-driven)", "value": world * nh * 2 / el,
                                   "unit": "scalar-muls/s", "ms_per_batch": el / 2 * 1e3}
        za = dg[: 32 * nh]
        zo = torch.empty(32 * nh, dtype=torch.uint8, device=dev)
        el = timed_steps(lambda: ctx.zp_op_dev("inv", nh, za.data_ptr(), None, zo.data_ptr()), 2)
        extras["zp_inverse"] = {"metric": "scalar-field inversions/s (2^20 per GPU)", "value": world * nh * 2 / el, "unit": "inversions/s",
                                "ms_per_batch": el / 2 * 1e3}
        ctx.sync()

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
        # HBM bytes per launch from the committed PMC passes (profiles/traffic.json), scaled to this launch size
        traffic = pair_traffic = None
        tr_path = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tr_path):
            try:
                tj = json.load(open(tr_path))
                traffic = tj["g1_mul_kernel_hbm_bytes_per_launch"] * units_per_launch / tj["units_per_launch"]
                if pair is not None:
                    pair_traffic = tj["pair_kernel"]["hbm_bytes_per_launch"] * pair["npair"] / tj["pair_kernel"]["units_per_launch"]
            except Exception:
                traffic = pair_traffic = None
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
        if pair is not None:
            npair = pair["npair"]
            pidx = list(range(4)) + [int(x) for x in np.random.Generator(np.random.PCG64(9)).integers(0, npair, size=12)]
            p1_h = pair["p1"].cpu().numpy().reshape(npair, 96)
            q2_h = pair["q2"].cpu().numpy().reshape(npair, 192)
            gt_h = pair["gt"].cpu().numpy().reshape(npair, 576)
            if orc.pair(p1_h[pidx].tobytes(), q2_h[pidx].tobytes(), 8) != gt_h[pidx].tobytes():
                raise SystemExit("bench: GPU pairing results differ from the CPU oracle — number withheld")
            avg_s = pair["kernel_ms"] / max(pair["launches"], 1) * 1e-3
            result["pairing"] = {
                "metric": "ate pairings/s per MI355X (batch 2^%d per GPU), bit-exact vs CPU" % args.log2_pairings,
                "value": world * npair * pair["steps"] / pair["elapsed"], "unit": "pairings/s",
                "ms_per_step": pair["elapsed"] / pair["steps"] * 1e3,
                "parity": {"checked_lanes": len(pidx), "oracle": kind, "bit_exact": True},
                "roofline": {"bound": "hbm", "achieved": BYTES_PER_PAIRING * npair / avg_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": BYTES_PER_PAIRING * npair / avg_s / 1e9 / HBM_PEAK_GBS, "traffic": pair_traffic, "kernel": ("pair_kernel" if os.environ.get("C12381_PAIR_LANES", "3") == "1" else
                                        ("pair3_queue_kernel" if (npair + 20) // 21 > 2048 and os.environ.get("C12381_PAIR_QUEUE", "") != "0" else "pair3_kernel")),
                             "avg_launch_ms": avg_s * 1e3},
                "valu_roofline": {"bound": "int-valu", "achieved": MAC32_PER_PAIRING * npair / avg_s / 1e9, "peak": VALU_PEAK_MAC32 / 1e9,
                                  "unit": "GMAC32/s", "frac": MAC32_PER_PAIRING * npair / avg_s / VALU_PEAK_MAC32,
                                  "algorithmic_mac32_per_unit": MAC32_PER_PAIRING},
            }
            if world == 1 and not args.no_cpu_baseline:
                ps = min(npair, 1 << 11)
                t3 = time.perf_counter()
                cpu_gt = orc.pair(p1_h[:ps].tobytes(), q2_h[:ps].tobytes(), cores)
                cpu_ps = time.perf_counter() - t3
                if cpu_gt != gt_h[:ps].tobytes():
                    raise SystemExit("bench: CPU pairing baseline differs from the GPU output")
                result["pairing"]["cpu_baseline"] = {"value": ps / cpu_ps, "unit": "pairings/s", "cores": cores, "kind": kind,
                                                     "sample": "first %d lanes of the same batch, %d threads; bit-exact vs GPU" % (ps, cores)}
        if extras:
            result["extra_configs"] = extras
        print(json.dumps(result), flush=True)
    ctx.close()
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
