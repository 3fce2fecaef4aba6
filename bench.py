#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X-native BLS12-381 backend.

Metric (BASELINE.json): G1 scalar-muls/s per MI355X on a batch of 2^20 random (point, scalar) pairs — BASELINE
configs[1] — bit-exact vs the CPU path.  A "step" is one pass of the hot path (c12381_g1_mul_batch_dev: scalar-mul
kernel + inversion/encode kernel) over one batch whose inputs are already resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Started plainly with --gpus N > 1 (no WORLD_SIZE in the environment) the script becomes the launcher: before anything touches the
GPU it starts N ranks of itself under torch.distributed.run on 127.0.0.1, relays their output and exits with their status.  Under a
launcher WORLD_SIZE must equal --gpus, otherwise the run stops (a silent 1-GPU run of an "N-GPU" command is the failure to avoid).

Rank 0 prints ONE JSON line, kept short enough for a log tail (floats rounded to 6 significant digits, explanations once under
`notes`, the pairing leg — the second half of BASELINE's metric — LAST).  Besides the headline it carries one object per remaining BASELINE config, each with its
own timing, parity against the CPU oracle, `roofline` / `hbm_roofline` and (N = 1) `cpu_baseline`:
    pairing   configs[2]  2^16 ate pairings (the WHOLE batch is compared with the oracle at N = 1)
    g2_mul / miller / fexp   the G2 scalar multiplication, the Miller loop and the final exponentiation alone
    msm       configs[3]  one product of 2^22 terms
    bbs_plus  configs[4]  2^18 BBS+ verifications (real signatures, a known set of corrupted lanes), decoded inputs
    bbs_plus_wire          the same 2^18 verifications END TO END from the serialized forms the reference's verify() parses
                           (145-B signatures, raw messages, 195 + 49 nh + 97 B of public material; bbs+.cpp:57-73)
and `cpu_baselines` for G2 multiplication, Miller loop and final exponentiation alone.
With --gpus N every rank runs the weak-scaled legs on its own shard (independent units, no data-path collective) and
two STRONG-scaled legs exercise what BASELINE describes for configs 4 and 5:
    msm_sharded       one 2^22-term product, terms split N ways, device-resident partial points (96 B) exchanged by ONE
                      all_gather on the process group's backend (nccl = RCCL over xGMI), local N-term sum on every rank
    bbs_plus_sharded  the 2^18 signatures split N ways, no collective on the data path
`roofline` is the BINDING bound of this path — integer VALU: algorithmic 32x32 multiply-adds / dominant-kernel
time against `peak` = the multiply-add issue rate measured on MI355X (csrc/microbench/valu_rates.hip, profiles/r03_valu_rates.txt),
with `peak_theoretical` = 256 CUs x 4 SIMDs x 16 lanes/clock x 2.4 GHz beside it; `traffic` = HBM bytes per launch from the
FETCH_SIZE / WRITE_SIZE counter passes (profiles/traffic.json).  `hbm_roofline` is the HBM view of the same kernel (algorithmic
bytes / kernel time against 8 TB/s): evidence that the path is not memory bound.  Legs `g2_mul`, `miller`, `fexp` time the three
remaining hot-path functions (PAIR_G2mul, PAIR_ate, PAIR_fexp) alone, each against the compiled reference on a sample.
`roofline.issue` prices the kernel's vector-instruction count (SQ_INSTS_VALU pass, profiles/issue.json — a property of the build) at
the clock the chip holds inside that kernel IN THIS RUN (one sampling lane per XCD beside the kernel, lib/libc12381_probe.so).
Only the parity / cpu_baseline legs touch oracle/: the compiled reference when oracle/_ref is present (`parity.pinned` true), else
our C port — then the headline's and the pairing leg's parity objects say `"oracle": "port"` and EVERY parity object `"pinned": false`;
nothing substitutes silently.  `streamed`: the same steps issued alternately from two contexts (what a caller that streams batches does).
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import socket
import subprocess
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
G2_GEN = bytes.fromhex(
    "13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e"
    "024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8"
    "0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be"
    "0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801")

# algorithmic work per unit: the reference's measured operation counts x (144 | 156) MAC32
MAC32_G1_MUL = 579_456
MAC32_G2_MUL = 1_163_808
MAC32_MILLER = 2_177_268
MAC32_FEXP = 2_097_972
MAC32_PAIRING = 4_275_240
# one bucket accumulation = one complete mixed addition: 11 products + 8 reductions (msm.hpp) -> 11*144 + 8*156 MAC32;
# the bucket method as built does 2 GLV halves x 8 windows = 16 of them per term (the actual count, SURVEY.md 8(d) "report actual")
MAC32_MSM_TERM = 16 * (11 * 144 + 8 * 156)
# one verification as the reference evaluates it (bbs+.cpp:57-73): 1 G2 mul + 2 G1 muls + 2 Miller loops + 1 final exponentiation
MAC32_BBS_VERIFY = MAC32_G2_MUL + 2 * MAC32_G1_MUL + 2 * MAC32_MILLER + MAC32_FEXP
# the pipeline as built, counted in the host simulation (tools/count_ops.py, profiles/r03_op_counts.json): one generic G1 multiplication (x A),
# two table-driven ones (r h0, m h1), four point additions, the product of two pairings against FIXED G2 arguments, 144 / 156 MAC32 per
# product / reduction like every other figure here
MAC32_BBS_PIPELINE = 5_427_672
BYTES_G1_MUL = 224               # 96 in + 32 scalar + 96 out
BYTES_PAIRING = 864              # 96 + 192 in, 576 out
BYTES_MSM_TERM = 128             # 96 + 32
BYTES_BBS_VERIFY = 96 + 32 + 32 + 32 + 1
# wire form: + the decode of A per signature (ECP_fromOctet: 502 products + 505 reductions); the public material is
# decoded once per batch.  Bytes: 145-B signature + 12-B message in, 1 B out.
MAC32_G1_DECODE = 502 * 144 + 505 * 156
MAC32_BBS_WIRE_PIPELINE = MAC32_BBS_PIPELINE + MAC32_G1_DECODE
BBS_MSG_LEN = 12
BYTES_BBS_WIRE = 145 + BBS_MSG_LEN + 1
HBM_PEAK_GBS = 8000.0
# The multiply-add rate, measured inside the kernel (csrc/microbench/issue_mix.hip, profiles/r03_issue_mix.txt: shader cycles per wavefront
# from s_memtime, 128 instructions per loop iteration): a SIMD issues one v_mad_i64_i32 — the instruction every limb product compiles
# to — per 4.125 cycles, at ONE wavefront per SIMD (4.25) as well as at two (the occupancy of every 256-register kernel here):
# 62.06 lanes per clock and CU of the documented 64.  `peak` prices that issue rate at the 2.4 GHz maximum clock: 3.81e13 MAC32/s.
# Earlier figures, kept for comparison only: rounds 1-2 used 3.10e13 (v_mad_u64_u32, a 1-ms run timed with events); the first half of
# round 3 3.30e13 (csrc/microbench/valu_rates.hip, 16 multiply-adds per loop iteration timed with events: the loop's own scalar
# instructions and the uneven arrival of the workgroups are in that number — profiles/r03_valu_rates.txt).
VALU_PEAK_MAC32 = 256 * 4 * (64 / 4.125) * 2.4e9
VALU_PEAK_MAC32_R02 = 3.10e13
# What bounds these kernels is the ISSUE of vector instructions, whatever they are (profiles/r03_issue_mix.txt): a SIMD takes its next
# vector instruction from its OLDEST wavefront every 4.06 cycles — multiply-add, mask, shift, select alike — and a second wavefront
# only fills the stalls of the first (mixed streams of two wavefronts take exactly the sum of their single-wavefront times).
# roofline.issue = VALU instructions per launch (SQ_INSTS_VALU, profiles/issue.json) x 4.06 cycles / 1024 SIMDs / the clock the chip
# holds inside that kernel (tools/clock_probe.py), against the measured launch time.
VALU_ISSUE_CYCLES = 4.06
# documented ceiling: a wave64 64-bit multiply-add occupies its SIMD for 4 cycles = 16 lanes per clock and SIMD,
# 256 CUs x 4 SIMDs, 2.4 GHz maximum clock (MI355X_MICROARCH.md)
VALU_PEAK_THEORETICAL_MAC32 = 256 * 4 * 16 * 2.4e9
BYTES_G2_MUL = 416               # 192 in + 32 scalar + 192 out


def make_scalars(seed: int, n: int, edges: bool = True) -> np.ndarray:
    """n x 32 big-endian scalars, uniform 256-bit values (the path reduces mod r), fixed edge lanes in front."""
    rng = np.random.Generator(np.random.PCG64(seed))
    sc = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    if edges:
        for j, k in enumerate([0, 1, R_ORDER - 1, R_ORDER, (1 << 256) - 1]):
            if j < n:
                sc[j] = np.frombuffer(k.to_bytes(32, "big"), dtype=np.uint8)
    return sc


def reduced_scalars(seed: int, n: int) -> np.ndarray:
    sc = make_scalars(seed, n, edges=False)
    sc[:, 0] &= 0x3f                                  # < 2^254 < r
    return sc


def dev_bytes(b, dev):
    return torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)


def sum_of_products_mod_r(a: np.ndarray, b: np.ndarray) -> int:
    """sum a_i * b_i mod r for n x 32 big-endian byte rows, exact (Python integers over 2^16-row blocks)."""
    tot = 0
    ab, bb = a.tobytes(), b.tobytes()
    for i in range(a.shape[0]):
        tot += int.from_bytes(ab[32 * i:32 * i + 32], "big") * int.from_bytes(bb[32 * i:32 * i + 32], "big")
    return tot % R_ORDER


def compact(o):
    """floats to 5 significant digits, recursively (the line has to fit a log tail)"""
    if isinstance(o, float):
        return int(o) if o.is_integer() and abs(o) < 1e15 else float("%.5g" % o)     # counts stay exact
    if isinstance(o, dict):
        return {k: compact(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [compact(v) for v in o]
    return o


def launch_ranks(ngpus: int) -> int:
    """`python bench.py --gpus N` started plainly: run N ranks of this script (one per GPU) under torch.distributed.run and relay
    their output.  Called before anything in this process has touched the GPU; the children are fresh processes."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ngpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("bench.py: --gpus %d without a launcher: starting %d ranks: %s" % (ngpus, ngpus, " ".join(cmd)), file=sys.stderr, flush=True)
    return subprocess.run(cmd).returncode


class ClockProbe:
    """the clock the chip holds while fn() runs back to back: lib/libc12381_probe.so samples (s_memtime, s_memrealtime) from one lane PER XCD on
    a stream of its own (csrc/microbench/clock_probe.hip); per XCD the median over the second half of the run.  during() returns the MEAN over
    the XCDs (a multi-round launch is dispatched to whichever XCD has room: the chip's throughput follows the sum of their clocks) and keeps
    the per-XCD values in .last_xcd"""

    def __init__(self, dev_index, dev):
        path = os.path.join(ROOT, "crypto12381_amd", "lib", "libc12381_probe.so")
        self.lib = ctypes.CDLL(path) if os.path.exists(path) else None
        self.dev_index, self.dev = dev_index, dev
        self.nsamp, self.gap = 2500, 30                       # ~100 us per sample: 0.25 s
        self.last_xcd = None
        if self.lib is not None:
            self.lib.c12381_probe_start.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
            self.nx = self.lib.c12381_probe_xcc_max() if hasattr(self.lib, "c12381_probe_xcc_max") else 1
            self.buf = torch.zeros(self.nx * (2 * self.nsamp + 2), dtype=torch.int64, device=dev)

    def during(self, fn, sync):
        if self.lib is None:
            return None
        self.buf.zero_()
        torch.cuda.synchronize(self.dev)
        t0 = time.perf_counter()
        if self.lib.c12381_probe_start(self.dev_index, ctypes.c_void_p(self.buf.data_ptr()), self.nsamp, self.gap) != 0:
            return None
        reps = 0
        while time.perf_counter() - t0 < 0.25:
            fn()
            reps += 1
            if reps % 4 == 0:
                sync()
        sync()
        el = time.perf_counter() - t0
        torch.cuda.synchronize(self.dev)
        allx = self.buf.cpu().numpy().reshape(self.nx, 2 * self.nsamp + 2)
        per = []
        for row in allx:
            k = int(row[1])
            if row[0] == 0 or k < 16:
                continue
            h = row[2:2 + 2 * k].reshape(k, 2)
            span = h[:, 1] - h[0, 1]                          # 100 MHz ticks since this sampler started
            inrun = h[(span > 0.4 * el * 1e8) & (span < 0.95 * el * 1e8)]
            if len(inrun) < 8:
                continue
            ghz = np.diff(inrun[:, 0]) / np.maximum(np.diff(inrun[:, 1]), 1) * 0.1
            per.append(float(np.median(ghz)))
        if not per:
            return None
        self.last_xcd = sorted(per)
        return float(np.mean(per))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--log2-batch", type=int, default=20)
    ap.add_argument("--log2-pairings", type=int, default=16)
    ap.add_argument("--pairings", type=int, default=0, help="exact pairing batch size (overrides --log2-pairings; experiments only)")
    ap.add_argument("--log2-g2", type=int, default=18, help="batch of the G2 scalar-multiplication leg")
    ap.add_argument("--no-split", action="store_true", help="skip the g2_mul / miller / fexp legs")
    ap.add_argument("--log2-msm", type=int, default=22)
    ap.add_argument("--log2-bbs", type=int, default=18)
    ap.add_argument("--no-pairing", action="store_true", help="skip the pairings/s leg")
    ap.add_argument("--no-msm", action="store_true", help="skip the MSM leg (configs[3])")
    ap.add_argument("--no-bbs", action="store_true", help="skip the BBS+ leg (configs[4])")
    ap.add_argument("--all-configs", action="store_true", help="also time the SURVEY 8(f) extras (hash-to-G1, fixed base, Zp inversion, aggregate BBS+)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-streamed", action="store_true", help="skip the two-context streamed legs")
    ap.add_argument("--sampled-parity", action="store_true", help="compare 64 pairing lanes instead of the whole batch (quick A/B runs)")
    ap.add_argument("--no-clock-probe", action="store_true", help="skip the in-run clock probe (roofline.issue then has no clock)")
    ap.add_argument("--lib", default=None, help="another build of the same C ABI (A/B runs, tools/build_variant.sh); default: the product library")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            # nothing in this process has touched the GPU yet (importing torch does not): become the launcher of N fresh ranks
            sys.exit(launch_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE): refusing to report a %d-GPU run as %d GPUs"
                         % (args.gpus, world, world, args.gpus))
    # RCCL ("nccl") over xGMI on the GPU node; C12381_BENCH_BACKEND=gloo lets the N>1 path be rehearsed on one GPU
    backend = os.environ.get("C12381_BENCH_BACKEND", "nccl")
    # CPU baseline workers: one PROCESS per host core, forked here — before anything in this process touches the GPU — so that no worker
    # ever holds a HIP context.  Processes, not threads: MIRACL's constant-time moves advance a function-level static on every call
    # (oracle/pool.py), which costs the reference's G1 / G2 / final-exponentiation paths half their rate across threads of one process.
    if torch.cuda.device_count() == 0:                        # counting devices does not initialise the GPU; nothing is forked on a box without one
        raise SystemExit("bench.py needs a HIP device: the product path has no CPU fallback")
    from oracle.bindings import Oracle, have_reference
    from oracle.pool import OraclePool, OracleThreads
    kind = "reference" if have_reference() else "port"        # the compiled reference when it travelled with the snapshot, else the C port
    cores = min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 16)   # the GPU box gives one GPU a 16-CPU share
    do_cpu = world == 1 and not args.no_cpu_baseline
    cpu_pool = OraclePool(kind, cores) if do_cpu else None
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the product path has no CPU fallback")
    ndev = torch.cuda.device_count()
    if world > ndev and backend == "nccl":
        raise SystemExit("bench.py: %d ranks but %d visible GPU(s): one rank per GPU (C12381_BENCH_BACKEND=gloo rehearses N ranks on fewer GPUs)"
                         % (world, ndev))
    if args.lib:
        from crypto12381_amd import capi
        capi.use_library(args.lib)
    dev_index = local_rank % max(ndev, 1)          # one rank per GPU on the 8-GPU node; wraps only in single-GPU rehearsals
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    comm_dev = dev if backend == "nccl" else torch.device("cpu")
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)

    from crypto12381_amd import Context
    from crypto12381_amd.distributed import msm_sharded_tensors, shard_bounds
    ctx = Context(dev_index)
    stream = torch.cuda.Stream(device=dev)          # the library's kernels run on this torch-owned HIP stream
    ctx.set_stream(stream.cuda_stream)
    _events = []

    def inputs_ready():
        """Device-side edge from torch's current stream (where torch kernels and copies fill the inputs) to the library's stream: an event
        recorded there, handed to c12381_wait_event — the library cannot see a foreign stream (include/c12381_hip.h "Stream ordering");
        no host-side wait.  (Round 4 used a device-wide synchronise here, after a 2-rank run had read half-written inputs.)"""
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        ctx.wait_event(ev.cuda_event)
        _events.append(ev)

    def outputs_ready():
        """the opposite edge: torch's current stream waits for everything the library has launched so far (c12381_record_event)"""
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))               # creates the handle (lazily created events have none before a record)
        ctx.record_event(ev.cuda_event)
        torch.cuda.current_stream(dev).wait_event(ev)
        _events.append(ev)

    def timed(fn, steps, warmup):
        """W untimed + K timed steps bracketed by barrier + synchronize; MAX over ranks."""
        torch.cuda.synchronize(dev)                         # inputs produced by torch kernels on torch's stream are complete before the library reads them
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize(dev)
        if dist:
            dist.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize(dev)
        if dist:
            dist.barrier()
        el = time.perf_counter() - t0
        if dist:
            t = torch.tensor([el], dtype=torch.float64, device=comm_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    probe = None if args.no_clock_probe else ClockProbe(dev_index, dev)
    clocks = {}                                       # kernel name -> GHz held inside it in THIS run (rank 0 reports them)

    xcd_clocks = {}                                   # kernel name -> (min, max) over the XCDs' samplers

    def probe_clock(name, fn):
        if probe is not None and rank == 0:
            clocks[name] = probe.during(fn, ctx.sync)
            if clocks[name] and probe.last_xcd:
                xcd_clocks[name] = (probe.last_xcd[0], probe.last_xcd[-1], len(probe.last_xcd))

    n = 1 << args.log2_batch
    # ---- synthetic inputs, resident in HBM before the timed region
    base_sc_h = make_scalars(1000 + rank, n)
    base_sc = torch.from_numpy(base_sc_h).to(dev)
    sc_h = make_scalars(2000 + rank, n)
    sc = torch.from_numpy(sc_h).to(dev)
    gen1 = dev_bytes(G1_GEN, dev)
    pts = torch.empty(n * 96, dtype=torch.uint8, device=dev)
    out = torch.empty(n * 96, dtype=torch.uint8, device=dev)
    inputs_ready()
    ctx.g1_mul_fixed_dev(n, gen1.data_ptr(), base_sc.data_ptr(), pts.data_ptr(), 96)      # P_i = G^{s_i} (untimed)
    ctx.sync()
    # lanes 0 and 1 of base_sc are 0 and 1: P_0 = infinity, P_1 = G — edge inputs stay in the batch

    # ================================================================== configs[1]: G1 scalar multiplications (headline)
    ctx.profile(True)
    prof_on = [False]

    def g1_step():
        ctx.g1_mul_dev(n, pts.data_ptr(), sc.data_ptr(), out.data_ptr(), 96)
    # the profile brackets only the timed steps: run the warm-up with profiling off
    ctx.profile(False)
    for _ in range(args.warmup):
        g1_step()
    torch.cuda.synchronize(dev)
    ctx.profile(True)
    elapsed = timed(g1_step, args.steps, 0)
    mul_ms, mul_launches = ctx.profile_read(0)
    fin_ms, fin_launches = ctx.profile_read(1)
    ctx.profile(False)
    if ctx.sync() != 0:
        raise SystemExit("bench: invalid input point reported by the kernels")
    probe_clock("g1_mul_kernel", g1_step)

    # ================================================================== configs[2]: pairings
    pair = None
    if not args.no_pairing:
        npair = args.pairings if args.pairings > 0 else (1 << args.log2_pairings)
        t_sc = torch.from_numpy(reduced_scalars(3000 + rank, npair)).to(dev)
        gen2 = dev_bytes(G2_GEN, dev)
        q2 = torch.empty(npair * 192, dtype=torch.uint8, device=dev)
        gt = torch.empty(npair * 576, dtype=torch.uint8, device=dev)
        p1 = pts[: npair * 96] if npair <= n else pts.repeat((npair + n - 1) // n)[: npair * 96].contiguous()
        inputs_ready()
        ctx.g2_mul_fixed_dev(npair, gen2.data_ptr(), t_sc.data_ptr(), q2.data_ptr(), 192)   # Q_i = G2^{t_i} (untimed)
        if ctx.sync() != 0:
            raise SystemExit("bench: the fixed-base G2 multiplication reported an invalid point")
        for _ in range(max(1, args.warmup)):
            ctx.pair_dev(npair, p1.data_ptr(), q2.data_ptr(), gt.data_ptr())
        torch.cuda.synchronize(dev)
        ctx.profile(True)
        pel = timed(lambda: ctx.pair_dev(npair, p1.data_ptr(), q2.data_ptr(), gt.data_ptr()), max(1, args.steps), 0)
        pk_ms, pk_launches = ctx.profile_read(3)
        ctx.profile(False)
        if ctx.sync() != 0:
            raise SystemExit("bench: invalid input reported by the pairing kernel")
        probe_clock("pair", lambda: ctx.pair_dev(npair, p1.data_ptr(), q2.data_ptr(), gt.data_ptr()))
        pair = {"npair": npair, "steps": max(1, args.steps), "elapsed": pel, "kernel_ms": pk_ms, "launches": pk_launches,
                "p1": p1, "q2": q2, "gt": gt}

    # ================================================================== the remaining hot-path functions alone: G2 mul, Miller loop, final exponentiation
    split = None
    if pair is not None and not args.no_split:
        npair = pair["npair"]
        ng2 = 1 << args.log2_g2
        ssteps = min(max(1, args.steps), 5)
        g2_sc_h = make_scalars(3500 + rank, ng2)
        g2_sc = torch.from_numpy(g2_sc_h).to(dev)
        reps2 = (ng2 + npair - 1) // npair
        g2_in = pair["q2"].repeat(reps2)[: ng2 * 192].contiguous() if reps2 > 1 else pair["q2"][: ng2 * 192]
        g2_out = torch.empty(ng2 * 192, dtype=torch.uint8, device=dev)
        mil = torch.empty(npair * 576, dtype=torch.uint8, device=dev)
        fex = torch.empty(npair * 576, dtype=torch.uint8, device=dev)
        # g2_in may be the output of a torch kernel (repeat / contiguous) on torch's current stream, the library enqueues on ITS stream: without
        # this edge the first launch can read g2_in before it is written — seen as "invalid input" on one of three 2-rank runs sharing a GPU
        inputs_ready()
        ctx.g2_mul_dev(ng2, g2_in.data_ptr(), g2_sc.data_ptr(), g2_out.data_ptr(), 192)
        if ctx.sync() != 0:
            raise SystemExit("bench: invalid input reported by the G2 multiplication kernel")
        ctx.miller_dev(npair, pair["p1"].data_ptr(), pair["q2"].data_ptr(), mil.data_ptr())
        if ctx.sync() != 0:
            raise SystemExit("bench: invalid input reported by the Miller-loop kernel")
        ctx.gt_op_dev("fexp", npair, mil.data_ptr(), None, fex.data_ptr())
        torch.cuda.synchronize(dev)
        ctx.profile(True)
        g2_el = timed(lambda: ctx.g2_mul_dev(ng2, g2_in.data_ptr(), g2_sc.data_ptr(), g2_out.data_ptr(), 192), ssteps, 0)
        mil_el = timed(lambda: ctx.miller_dev(npair, pair["p1"].data_ptr(), pair["q2"].data_ptr(), mil.data_ptr()), ssteps, 0)
        fex_el = timed(lambda: ctx.gt_op_dev("fexp", npair, mil.data_ptr(), None, fex.data_ptr()), ssteps, 0)
        g2k_ms, g2k_launches = ctx.profile_read(2)
        milk_ms, milk_launches = ctx.profile_read(6)
        fexk_ms, fexk_launches = ctx.profile_read(7)
        ctx.profile(False)
        if ctx.sync() != 0:
            raise SystemExit("bench: invalid input reported by the split kernels")
        probe_clock("g2_mul2_kernel", lambda: ctx.g2_mul_dev(ng2, g2_in.data_ptr(), g2_sc.data_ptr(), g2_out.data_ptr(), 192))
        probe_clock("miller", lambda: ctx.miller_dev(npair, pair["p1"].data_ptr(), pair["q2"].data_ptr(), mil.data_ptr()))
        probe_clock("fexp", lambda: ctx.gt_op_dev("fexp", npair, mil.data_ptr(), None, fex.data_ptr()))
        split = {"ng2": ng2, "steps": ssteps, "g2_el": g2_el, "mil_el": mil_el, "fex_el": fex_el, "g2k": (g2k_ms, g2k_launches),
                 "milk": (milk_ms, milk_launches), "fexk": (fexk_ms, fexk_launches), "g2_in": g2_in, "g2_sc_h": g2_sc_h, "g2_out": g2_out,
                 "mil": mil, "fex": fex}

    # ================================================================== streamed batches: the same steps issued alternately from TWO contexts
    # A caller that streams batches owns two contexts (two HIP streams, two sets of workspaces): the head of batch i+1 fills the tail of batch i,
    # in which the older wavefront of every SIMD has left.  Same inputs, two output buffers, both compared with the serial leg's
    # output (itself checked against the CPU reference below).  The contract's legs above stay serial: their kernel times are launch times.
    streamed = None
    if not args.no_streamed:
        ctx2 = Context(dev_index)
        stream2 = torch.cuda.Stream(device=dev)
        ctx2.set_stream(stream2.cuda_stream)

        def alternate(call, outs, steps):
            k = [0]

            def step():
                i = k[0] & 1
                k[0] += 1
                call((ctx, ctx2)[i], outs[i])
            for c, o in ((ctx, outs[0]), (ctx2, outs[1])):       # both contexts warm (workspaces, tables)
                call(c, o)
            el = timed(step, steps, 0)
            if ctx.sync() != 0 or ctx2.sync() != 0:
                raise SystemExit("bench: invalid input reported in the streamed legs")
            return el
        streamed = {}
        o2 = [torch.empty_like(out), torch.empty_like(out)]
        el = alternate(lambda c, o: c.g1_mul_dev(n, pts.data_ptr(), sc.data_ptr(), o.data_ptr(), 96), o2, args.steps)
        streamed["g1"] = (n * args.steps / el, el / args.steps * 1e3, bool(torch.equal(o2[0], out) and torch.equal(o2[1], out)))
        del o2
        if pair is not None:
            npair, p1, q2 = pair["npair"], pair["p1"], pair["q2"]
            st = pair["steps"]
            g2b = [torch.empty_like(pair["gt"]), torch.empty_like(pair["gt"])]
            el = alternate(lambda c, o: c.pair_dev(npair, p1.data_ptr(), q2.data_ptr(), o.data_ptr()), g2b, st)
            streamed["pairing"] = (npair * st / el, el / st * 1e3, bool(torch.equal(g2b[0], pair["gt"]) and torch.equal(g2b[1], pair["gt"])))
            if split is not None:
                el = alternate(lambda c, o: c.miller_dev(npair, p1.data_ptr(), q2.data_ptr(), o.data_ptr()), g2b, st)
                streamed["miller"] = (npair * st / el, el / st * 1e3, bool(torch.equal(g2b[0], split["mil"]) and torch.equal(g2b[1], split["mil"])))
                el = alternate(lambda c, o: c.gt_op_dev("fexp", npair, split["mil"].data_ptr(), None, o.data_ptr()), g2b, st)
                streamed["fexp"] = (npair * st / el, el / st * 1e3, bool(torch.equal(g2b[0], split["fex"]) and torch.equal(g2b[1], split["fex"])))
                g2o = [torch.empty_like(split["g2_out"]), torch.empty_like(split["g2_out"])]
                el = alternate(lambda c, o: c.g2_mul_dev(split["ng2"], split["g2_in"].data_ptr(), g2_sc.data_ptr(), o.data_ptr(), 192), g2o, st)
                streamed["g2_mul"] = (split["ng2"] * st / el, el / st * 1e3, bool(torch.equal(g2o[0], split["g2_out"]) and torch.equal(g2o[1], split["g2_out"])))
                del g2o
            del g2b

    # ================================================================== configs[3]: MSM, n = 2^22 per GPU (weak) and sharded (strong)
    msm = None
    if not args.no_msm:
        nm = 1 << args.log2_msm
        reps = max(1, nm // n)

        def msm_points(first_pts, first_base_h, seed):
            """nm DISTINCT points P_i = G^{b_i} (configs[3] reads 2^22 terms g_i^x_i, not 2^20 points four times): block 0 is the G1 leg's batch
            (edge lanes included), the further blocks of n points come from their own seeds, untimed; returns the points and the b_i"""
            if reps == 1:
                return first_pts[: nm * 96], first_base_h[:nm]
            full = torch.empty(nm * 96, dtype=torch.uint8, device=dev)
            full[: n * 96] = first_pts
            bases = [first_base_h]
            for j in range(1, reps):
                b_h = make_scalars(seed + 7919 * j, n, edges=False)
                b_d = torch.from_numpy(b_h).to(dev)
                inputs_ready()
                ctx.g1_mul_fixed_dev(n, gen1.data_ptr(), b_d.data_ptr(), full[j * n * 96:].data_ptr(), 96)
                ctx.sync()
                bases.append(b_h)
            return full, np.concatenate(bases)[:nm]

        mp_, msm_base_h = msm_points(pts, base_sc_h, 1000 + rank)
        ms_h = make_scalars(4000 + rank, nm)
        ms_ = torch.from_numpy(ms_h).to(dev)
        mo_ = torch.empty(96, dtype=torch.uint8, device=dev)
        msteps = min(max(1, args.steps), 5)
        inputs_ready()                                      # mp_ and ms_ come from torch copies on torch's stream
        ctx.g1_msm_dev(nm, mp_.data_ptr(), ms_.data_ptr(), mo_.data_ptr(), 96)
        torch.cuda.synchronize(dev)
        ctx.profile(True)
        mel = timed(lambda: ctx.g1_msm_dev(nm, mp_.data_ptr(), ms_.data_ptr(), mo_.data_ptr(), 96), msteps, 0)
        bk_ms, bk_launches = ctx.profile_read(5)
        ctx.profile(False)
        probe_clock("msm_bucket_kernel", lambda: ctx.g1_msm_dev(nm, mp_.data_ptr(), ms_.data_ptr(), mo_.data_ptr(), 96))
        msm = {"n": nm, "steps": msteps, "elapsed": mel, "bucket_ms": bk_ms, "bucket_launches": bk_launches, "out": mo_.cpu().numpy().tobytes(),
               "reps": reps, "scalars": ms_h, "bases": msm_base_h}
        if world > 1:
            # strong scaling: ONE product for the whole job — identical terms on every rank (seed without the rank), each rank takes its shard
            gs_h = make_scalars(4100, nm)
            gb_h = make_scalars(1000, n)                               # the points of rank 0's batch: P_i = G^{s_i}
            gpts = pts if rank == 0 else torch.empty(n * 96, dtype=torch.uint8, device=dev)
            if rank != 0:
                gb = torch.from_numpy(gb_h).to(dev)
                ctx.g1_mul_fixed_dev(n, gen1.data_ptr(), gb.data_ptr(), gpts.data_ptr(), 96)
                ctx.sync()
            gfull, _ = msm_points(gpts, gb_h, 1000)                    # the same nm distinct points on every rank
            lo, hi = shard_bounds(nm, rank, world)
            sp_, ss_ = gfull[96 * lo:96 * hi], torch.from_numpy(gs_h[lo:hi]).to(dev)
            inputs_ready()

            def local_t(p, s, fmt):
                o = torch.empty(fmt, dtype=torch.uint8, device=dev)
                with torch.cuda.stream(stream):
                    ctx.g1_msm_dev(p.numel() // 96, p.data_ptr(), s.data_ptr(), o.data_ptr(), fmt)
                return o

            def sum_t(p, fmt):                                         # the combine: plain sum of the N partial points
                o = torch.empty(fmt, dtype=torch.uint8, device=dev)
                with torch.cuda.stream(stream):
                    ctx.g1_sum_dev(p.numel() // 96, p.data_ptr(), o.data_ptr(), fmt)
                return o

            res = [None]

            def sharded_step():
                if backend == "nccl":
                    res[0] = msm_sharded_tensors(local_t, sp_, ss_, 96, stream=stream, combine_t=sum_t)
                else:
                    part = local_t(sp_, ss_, 96)
                    stream.synchronize()
                    g = torch.empty(96 * world, dtype=torch.uint8)
                    dist.all_gather_into_tensor(g, part.cpu())
                    res[0] = sum_t(g.to(dev), 96)
            sel = timed(sharded_step, msteps, 1)
            mine = res[0].to(comm_dev)
            allres = torch.empty(96 * world, dtype=torch.uint8, device=comm_dev)
            dist.all_gather_into_tensor(allres, mine)
            same = all(bytes(allres[96 * r:96 * r + 96].cpu().numpy().tobytes()) == bytes(mine.cpu().numpy().tobytes()) for r in range(world))
            single = None
            if rank == 0:                                               # the N = 1 value of the same product, untimed
                so = torch.empty(96, dtype=torch.uint8, device=dev)
                ctx.g1_msm_dev(nm, gfull.data_ptr(), torch.from_numpy(gs_h).to(dev).data_ptr(), so.data_ptr(), 96)
                ctx.sync()
                single = so.cpu().numpy().tobytes() == mine.cpu().numpy().tobytes()
            msm["sharded"] = {"elapsed": sel, "steps": msteps, "same_on_every_rank": bool(same), "equals_single_gpu": single,
                              "rccl_ranks": dist.get_world_size(), "backend": backend}
            del gfull, sp_, ss_
        del mp_, ms_

    # ================================================================== configs[4]: BBS+ verifications
    bbs = None
    if not args.no_bbs:
        nb = 1 << args.log2_bbs
        # public parameters and key as setup(16) / key_gen make them (bbs+.cpp:7-36, examples/bbs-plus/test.cpp:11-14): g1, h0 and SIXTEEN h_i,
        # multiples of the generators; one-block messages use h_1 only, the wire leg decodes all sixteen (pp.h)
        BBS_NH = 16
        pub = ctx.g1_mul_fixed(G1_GEN, reduced_scalars(5100, 2 + BBS_NH).tobytes(), 96)
        pub_g1, pub_h0, pub_h, pub_h_all = pub[:96], pub[96:192], pub[192:288], pub[192:192 + 96 * BBS_NH]
        g2p = ctx.g2_mul_fixed(G2_GEN, reduced_scalars(5101, 1).tobytes(), 192)
        gamma = reduced_scalars(5102, 1).tobytes()
        w = ctx.g2_mul_fixed(g2p, gamma, 192)
        # strong leg: the job's 2^18 signatures are the same on every rank (seeds without the rank); the weak leg of rank r uses its own
        def encode_msgs(raw):
            """encode_to<Zp> (zp_number.hpp:1011-1037) of one unit: 0x01 || message || zero padding as a 32-byte big-endian scalar"""
            mm = np.zeros((raw.shape[0], 32), dtype=np.uint8)
            mm[:, 0] = 1
            mm[:, 1:1 + raw.shape[1]] = raw
            return mm

        def make_sigs(seed, count):
            xs, rs = reduced_scalars(seed, count), reduced_scalars(seed + 1, count)
            raw = np.random.Generator(np.random.PCG64(seed + 2)).integers(0, 256, size=(count, BBS_MSG_LEN), dtype=np.uint8)   # raw messages
            A = np.frombuffer(ctx.bbs_plus_sign(pub_g1, pub_h0, pub_h, gamma, xs.tobytes(), rs.tobytes(), encode_msgs(raw).tobytes()),
                              dtype=np.uint8).reshape(count, 96).copy()
            badl = np.arange(7, count, 1009)
            raw[badl, BBS_MSG_LEN - 1] ^= 1                            # corrupted message: exactly these lanes must fail
            return A, xs, rs, encode_msgs(raw), badl, raw
        A_h, xs_h, rs_h, mm_h, bad_lanes, raw_h = make_sigs(5200 + 10 * rank, nb)
        dA, dx, dr, dm = (torch.from_numpy(a).to(dev) for a in (A_h, xs_h, rs_h, mm_h))
        dpub = [dev_bytes(b, dev) for b in (pub_g1, g2p, pub_h0, pub_h, w)]
        okb = torch.empty(nb, dtype=torch.uint8, device=dev)
        inputs_ready()

        def bbs_step(count=nb, A=dA, x=dx, r=dr, m=dm, ok=okb):
            ctx.bbs_plus_verify_dev(count, 1, dpub[0].data_ptr(), dpub[1].data_ptr(), dpub[2].data_ptr(), dpub[3].data_ptr(), dpub[4].data_ptr(),
                                    A.data_ptr(), x.data_ptr(), r.data_ptr(), m.data_ptr(), ok.data_ptr())
        bsteps = min(max(1, args.steps), 5)
        bbs_step()
        torch.cuda.synchronize(dev)
        ctx.profile(True)
        bel = timed(bbs_step, bsteps, 0)
        bpk_ms, bpk_launches = ctx.profile_read(4)
        ctx.profile(False)
        ctx.sync()
        ok_h = okb.cpu().numpy()
        exp_ok = np.ones(nb, dtype=np.uint8); exp_ok[bad_lanes] = 0
        if not (ok_h == exp_ok).all():
            raise SystemExit("bench: BBS+ verdicts differ from the construction (valid signatures / corrupted lanes) — number withheld")
        probe_clock("pair3_prod_fixed_queue_kernel", bbs_step)
        bbs = {"n": nb, "steps": bsteps, "elapsed": bel, "pair_ms": bpk_ms, "pair_launches": bpk_launches, "A": A_h, "x": xs_h, "r": rs_h, "m": mm_h, "ok": ok_h,
               "pub": (pub_g1, g2p, pub_h0, pub_h, w)}
        # ---- the same verifications end to end from the wire formats verify() parses (bbs+.cpp:57-73): pp.g1_g2_h0 (49 + 97 + 49 B),
        # pp.h (49 B each), pk (97 B), signatures A || x || r as 49 + 48 + 48 B, raw messages.  Serialized on the device, untimed.
        one32 = torch.zeros(nb, 32, dtype=torch.uint8, device=dev); one32[:, 31] = 1
        A49 = torch.empty(nb * 49, dtype=torch.uint8, device=dev)
        inputs_ready()                                      # one32 was filled by torch kernels
        ctx.g1_mul_flags_dev(nb, dA.data_ptr(), one32.data_ptr(), A49.data_ptr(), 49, 1)           # 1 x A in the compressed form (C12381_F_IN_SUBGROUP)
        sig = torch.zeros(nb, 145, dtype=torch.uint8, device=dev)
        outputs_ready()                                     # torch assembles sig from A49 on its own stream
        sig[:, 0:49] = A49.view(nb, 49); sig[:, 65:97] = dx.view(nb, 32); sig[:, 113:145] = dr.view(nb, 32)
        one1 = (1).to_bytes(32, "big")
        pp195 = ctx.g1_mul(pub_g1, one1, 49) + ctx.g2_mul(g2p, one1, 97) + ctx.g1_mul(pub_h0, one1, 49)
        h49, pk97 = ctx.g1_mul(pub_h_all, one1 * BBS_NH, 49), ctx.g2_mul(w, one1, 97)
        dwire = [dev_bytes(b, dev) for b in (pp195, h49, pk97)]
        draw = torch.from_numpy(raw_h).to(dev)
        okw = torch.empty(nb, dtype=torch.uint8, device=dev)
        del one32, A49
        inputs_ready()                                      # sig was assembled by torch kernels on torch's stream

        def wire_step():
            ctx.bbs_plus_verify_wire_dev(nb, BBS_NH, BBS_MSG_LEN, dwire[0].data_ptr(), dwire[1].data_ptr(), dwire[2].data_ptr(), sig.data_ptr(), draw.data_ptr(),
                                         okw.data_ptr())
        wire_step()
        torch.cuda.synchronize(dev)
        wel = timed(wire_step, bsteps, 0)
        ctx.sync()
        okw_h = okw.cpu().numpy()
        if not (okw_h == exp_ok).all():
            raise SystemExit("bench: wire-format BBS+ verdicts differ from the construction (valid signatures / corrupted lanes) — number withheld")
        bbs["wire"] = {"elapsed": wel, "pp": pp195, "h49": h49, "pk": pk97, "sig": sig.cpu().numpy(), "raw": raw_h, "ok": okw_h}
        del sig, draw, okw
        if world > 1:
            gA, gx, gr, gm, gbad = make_sigs(5200, nb)[:5] if rank != 0 else (A_h, xs_h, rs_h, mm_h, bad_lanes)
            lo, hi = shard_bounds(nb, rank, world)
            sA, sx, sr, sm = (torch.from_numpy(np.ascontiguousarray(a[lo:hi])).to(dev) for a in (gA, gx, gr, gm))
            sok = torch.empty(max(hi - lo, 1), dtype=torch.uint8, device=dev)
            inputs_ready()
            sel = timed(lambda: bbs_step(hi - lo, sA, sx, sr, sm, sok), bsteps, 1)
            ctx.sync()
            cnt = torch.tensor([int(sok[: hi - lo].sum().item())], dtype=torch.int64, device=comm_dev)
            dist.all_reduce(cnt)                                        # bookkeeping only: number of accepted signatures in the job
            bbs["sharded"] = {"elapsed": sel, "steps": bsteps, "accepted": int(cnt.item()), "expected_accepted": int(nb - len(gbad))}
            del sA, sx, sr, sm

    if streamed is not None:
        if bbs is not None:
            ok2 = [torch.empty_like(okb), torch.empty_like(okb)]
            el = alternate(lambda c, o: c.bbs_plus_verify_dev(nb, 1, dpub[0].data_ptr(), dpub[1].data_ptr(), dpub[2].data_ptr(), dpub[3].data_ptr(),
                                                              dpub[4].data_ptr(), dA.data_ptr(), dx.data_ptr(), dr.data_ptr(), dm.data_ptr(), o.data_ptr()),
                           ok2, bbs["steps"])
            streamed["bbs_plus"] = (nb * bbs["steps"] / el, el / bbs["steps"] * 1e3, bool(torch.equal(ok2[0], okb) and torch.equal(ok2[1], okb)))
        ctx2.close()
        bad = [k for k, v in streamed.items() if not v[2]]
        if bad:
            raise SystemExit("bench: streamed outputs differ from the serial legs' (%s)" % ", ".join(bad))

    # ================================================================== optional extras
    extras = {}
    if args.all_configs:
        nh = 1 << 20
        dg = torch.from_numpy(np.frombuffer(make_scalars(6000 + rank, 2 * nh).tobytes(), dtype=np.uint8).copy()).to(dev)
        ho = torch.empty(96 * nh, dtype=torch.uint8, device=dev)
        el = timed(lambda: ctx.g1_from_hash_dev(nh, dg.data_ptr(), ho.data_ptr(), 96), 2, 1)
        extras["hash_to_g1"] = {"metric": "hash-to-G1 points/s (2^20 SHA3-512 digests per GPU -> affine G1)", "value": world * nh * 2 / el,
                                "unit": "points/s", "ms_per_batch": el / 2 * 1e3}
        fo = torch.empty(96 * nh, dtype=torch.uint8, device=dev)
        el = timed(lambda: ctx.g1_mul_fixed_dev(nh, pts[96:192].data_ptr(), sc.data_ptr(), fo.data_ptr(), 96), 2, 1)
        extras["g1_fixed_base"] = {"metric": "G1 scalar-muls/s with one base for the batch (2^20 per GPU, table-driven)", "value": world * nh * 2 / el,
                                   "unit": "scalar-muls/s", "ms_per_batch": el / 2 * 1e3}
        za = dg[: 32 * nh]
        zo = torch.empty(32 * nh, dtype=torch.uint8, device=dev)
        el = timed(lambda: ctx.zp_op_dev("inv", nh, za.data_ptr(), None, zo.data_ptr()), 2, 1)
        extras["zp_inverse"] = {"metric": "scalar-field inversions/s (2^20 per GPU)", "value": world * nh * 2 / el, "unit": "inversions/s",
                                "ms_per_batch": el / 2 * 1e3}
        if bbs is not None:
            rho = torch.from_numpy(make_scalars(5003 + rank, bbs["n"])).to(dev)
            rho.view(bbs["n"], 32)[:, :16] = 0                          # 128-bit coefficients
            ok1 = torch.empty(16, dtype=torch.uint8, device=dev)
            el = timed(lambda: ctx.bbs_plus_verify_aggregate_dev(bbs["n"], 1, dpub[0].data_ptr(), dpub[1].data_ptr(), dpub[2].data_ptr(), dpub[3].data_ptr(),
                                                                 dpub[4].data_ptr(), dA.data_ptr(), dx.data_ptr(), dr.data_ptr(), dm.data_ptr(),
                                                                 rho.data_ptr(), ok1.data_ptr()), 2, 1)
            extras["bbs_plus_aggregate"] = {"metric": "BBS+ signatures/s covered by ONE aggregate verdict (optional mode, not in the reference)",
                                            "value": world * bbs["n"] * 2 / el, "unit": "signatures/s", "ms_per_batch": el / 2 * 1e3}
        ctx.sync()

    # ================================================================== parity and CPU baselines (outside every timed region)
    # kind: the compiled reference (oracle/_ref) or the C port — said on every parity object, never silent
    pinned = kind == "reference"
    orc = Oracle(kind)
    cpuN = cpu_pool if cpu_pool is not None else OracleThreads(orc, cores)      # `cores` workers: processes with the baseline, threads without
    if rank == 0 and not pinned:
        print("bench.py: oracle/_ref/libc12381_ref.so is absent: parity is checked against the C port (parity.pinned = false)", file=sys.stderr, flush=True)

    def cpu_time(fn):
        """CPU legs: a sample shorter than 2 s rides the host's CPU-quota burst and then its throttle — such a sample is run three times and
        the MEDIAN time counts (a longer one once); returns (output, seconds)"""
        ts, o = [], None
        while True:
            t = time.perf_counter()
            o = fn()
            ts.append(time.perf_counter() - t)
            if ts[0] >= 2.0 or len(ts) >= 3 or not do_cpu:         # parity-only runs (--no-cpu-baseline, N > 1): once
                return o, float(np.median(ts))
    pts_h = pts.cpu().numpy().reshape(n, 96)
    out_h = out.cpu().numpy().reshape(n, 96)
    # G1: with the CPU baseline the first 2^17 lanes are compared in full, otherwise 64 sampled lanes incl. every edge lane
    idx = list(range(8)) + [int(x) for x in np.random.Generator(np.random.PCG64(7)).integers(0, n, size=56)]
    exp = orc.g1_mul(pts_h[idx].tobytes(), sc_h[idx].tobytes(), 96, min(cores, 8))
    if exp != out_h[idx].tobytes():
        raise SystemExit("bench: GPU results differ from the CPU oracle — number withheld")
    g1_checked = len(idx)

    def par(head=False, **kw):
        """parity object of a leg; the oracle's kind is named on the headline's (and in notes.cpu_baseline), `pinned` on every one"""
        d = {"oracle": kind, "pinned": pinned, "bit_exact": True} if head else {"pinned": pinned, "bit_exact": True}
        d.update(kw)
        return d

    result = None
    if rank == 0:
        value = world * n * args.steps / elapsed
        launches_per_step = mul_launches / max(args.steps, 1)
        units_per_launch = n / max(launches_per_step, 1)
        avg_launch_s = (mul_ms / max(mul_launches, 1)) * 1e-3
        traffic = pair_traffic = msm_traffic = bbs_traffic = g2_traffic = mil_traffic = fex_traffic = None
        tr_path = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tr_path):
            try:
                tj = json.load(open(tr_path))
                traffic = tj["g1_mul_kernel_hbm_bytes_per_launch"] * units_per_launch / tj["units_per_launch"]
                if pair is not None:
                    pair_traffic = tj["pair_kernel"]["hbm_bytes_per_launch"] * pair["npair"] / tj["pair_kernel"]["units_per_launch"]
                if split is not None and "g2_mul2_kernel" in tj:
                    g2_traffic = tj["g2_mul2_kernel"]["hbm_bytes_per_launch"] / tj["g2_mul2_kernel"]["units_per_launch"]     # per point; scaled per launch below
                if split is not None and pair is not None and "miller3_queue_kernel" in tj and "fexp3_queue_kernel" in tj:   # per launch of the pairing leg's batch
                    mil_traffic = tj["miller3_queue_kernel"]["hbm_bytes_per_launch"] * pair["npair"] / tj["miller3_queue_kernel"]["units_per_launch"]
                    fex_traffic = tj["fexp3_queue_kernel"]["hbm_bytes_per_launch"] * pair["npair"] / tj["fexp3_queue_kernel"]["units_per_launch"]
                if msm is not None and "msm_bucket_kernel" in tj:
                    msm_traffic = tj["msm_bucket_kernel"]["hbm_bytes_per_launch"] * msm["n"] / tj["msm_bucket_kernel"]["units_per_launch"]
                if bbs is not None and "pair3_prod_fixed_queue_kernel" in tj:
                    bbs_traffic = tj["pair3_prod_fixed_queue_kernel"]["hbm_bytes_per_launch"] * bbs["n"] / tj["pair3_prod_fixed_queue_kernel"]["units_per_launch"]
            except Exception:
                traffic = pair_traffic = msm_traffic = bbs_traffic = g2_traffic = mil_traffic = fex_traffic = None

        issue_json = {}
        try:
            issue_json = json.load(open(os.path.join(ROOT, "profiles", "issue.json")))
        except Exception:
            issue_json = {}

        def valu(mac_per_unit, units, secs, kernel, tr=None, nbytes=None, clock_key=None, head=False, issue_secs=None, **more):
            """the binding roofline: algorithmic multiply-adds per launch / average launch time against the measured multiply-add rate;
            `hbm` beside it = algorithmic bytes / the same time against 8 TB/s (evidence that the path is not memory bound)"""
            a = mac_per_unit * units / secs
            # the legs' objects leave out what the headline's states once: bound int-valu, peak, unit GMAC32/s
            d = {"bound": "int-valu", "kernel": kernel, "achieved": a / 1e9, "peak": VALU_PEAK_MAC32 / 1e9, "unit": "GMAC32/s"} if head else {"kernel": kernel, "achieved": a / 1e9}
            d.update({"frac": a / VALU_PEAK_MAC32, "traffic": tr, "avg_launch_ms": secs * 1e3, "mac32_per_unit": mac_per_unit})
            if nbytes is not None:
                g = nbytes * units / secs / 1e9
                d["hbm_GBs"] = g                                   # algorithmic bytes / the same time; HBM peak 8000 GB/s
            ik = issue_json.get("kernels", {}).get(kernel)
            clk = clocks.get(clock_key or kernel)
            if ik and clk:
                insts = ik["valu_insts_per_launch"] * units / ik["units_per_launch"]
                bound_s = insts * VALU_ISSUE_CYCLES / 1024 / (clk * 1e9)
                # issue_secs: the launch time of the kernel the instructions were counted in, where `secs` is a longer time base (BBS+: the pipeline)
                xc = xcd_clocks.get(clock_key or kernel)
                d["issue"] = {"valu_insts": insts, "clock_GHz_in_run": clk, "issue_ms": bound_s * 1e3, "issue_over_launch": bound_s / (issue_secs or secs),
                              # the REFERENCE's work per issued lane-instruction — not the share of multiply-adds in the
                              # stream (that is 0.69-0.85, profiles/r05_isa_classes_*.txt)
                              "ref_mac32_per_inst": mac_per_unit * units / 64 / insts}
                if head and xc and kernel == "g1_mul_kernel":     # the headline kernel: the spread of the XCDs' clocks (the legs' lines stay short)
                    d["issue"]["xcd_clock_min_max"] = [xc[0], xc[1]]
            elif ik:
                d["issue"] = {"valu_insts": ik["valu_insts_per_launch"] * units / ik["units_per_launch"], "clock_GHz_in_run": None}
            d.update(more)
            return d
        result = {
            "metric": "G1 scalar-muls/s per MI355X (batch 2^%d per GPU), bit-exact vs CPU" % args.log2_batch,
            "value": value, "unit": "scalar-muls/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "rccl_ranks": (dist.get_world_size() if dist else 1),
            "vs_baseline": None, "dtype": "int64 accumulate over 14x28-bit signed limbs", "data": "synthetic",
            "config": {"workload": "configs[1]: 2^%d random G1 scalar-muls (96-B affine in, 32-B scalar, 96-B affine out) per GPU" % args.log2_batch,
                       "batch_per_gpu": n, "parallelism": "independent shards x%d" % world},
            "roofline": valu(MAC32_G1_MUL, units_per_launch, avg_launch_s, "g1_mul_kernel", traffic, BYTES_G1_MUL, head=True, launches=int(mul_launches),
                             units_per_launch=units_per_launch, finish_kernel_ms_per_step=fin_ms / max(args.steps, 1),
                             peak_theoretical=VALU_PEAK_THEORETICAL_MAC32 / 1e9),
        }
        if do_cpu:
            sample = min(n, 1 << 17)
            sp, ss = pts_h[:sample].tobytes(), sc_h[:sample].tobytes()
            cpu_out, cpu_s = cpu_time(lambda: cpuN.g1_mul(sp, ss, 96))
            _, cpu1_s = cpu_time(lambda: orc.g1_mul(sp[:96 * 4096], ss[:32 * 4096], 96, 1))
            nt = min(sample, 1 << 15)                              # BASELINE.md 3(b) as written: std::thread shards in ONE process
            _, cput_s = cpu_time(lambda: orc.g1_mul(sp[:96 * nt], ss[:32 * nt], 96, cores))
            if cpu_out != out_h[:sample].tobytes():
                raise SystemExit("bench: CPU baseline output differs from the GPU output")
            g1_checked = sample + len(idx)
            result["cpu_baseline"] = {"value": sample / cpu_s, "unit": "scalar-muls/s", "cores": cores, "workers": "processes", "kind": kind,
                                      "sample": "first %d lanes; all equal" % sample,
                                      "one_thread": 4096 / cpu1_s, "eff_cores": (sample / cpu_s) / (4096 / cpu1_s), "threads_value": nt / cput_s}
        result["parity"] = par(head=True, checked_lanes=g1_checked)

        def cpu_b(v, unit, sample, one=None, full_keys=False):
            """one = the same routine's rate on ONE thread (its own short sample): eff_cores = what `cores` threads deliver on this box"""
            d = {"value": v, "sample": sample}      # unit = the leg's unit; cores and kind: the headline's cpu_baseline and notes.cpu_baseline
            if full_keys:
                d.update(cores=cores, kind=kind)
            if one:
                d.update(one_thread=one, eff_cores=v / one)
            return d

        # ---------------------------------------------------------------- MSM
        if msm is not None:
            nm = msm["n"]
            # full-size parity: every P_i = G^{s_i}, so the product is G^(sum s_i k_i) — one oracle multiplication
            # lanes 0..4 of the first block's b_i are edge values (0, 1, r-1, r, 2^256-1): the exponent sum works mod r for all of them
            e = sum_of_products_mod_r(msm["bases"], msm["scalars"])
            if orc.g1_mul(G1_GEN, e.to_bytes(32, "big"), 96, 1) != msm["out"]:
                raise SystemExit("bench: MSM result differs from G^(sum s_i k_i) (CPU oracle) — number withheld")
            per = msm["elapsed"] / msm["steps"]
            bk_s = msm["bucket_ms"] / max(msm["bucket_launches"], 1) * 1e-3
            result["msm"] = {
                "metric": "G1 multi-scalar product terms/s (one product of 2^%d terms per GPU)" % args.log2_msm,
                "value": world * nm / per, "unit": "terms/s", "steps": msm["steps"], "ms_per_step": per * 1e3,
                "workload": "configs[3]: %d distinct points" % nm,
                "parity": par(check="result == G^(sum s_i k_i) over ALL terms"),
                # dominant kernel alone, then the whole product (sorts, preparation, reductions included) against the same peak
                "roofline": valu(MAC32_MSM_TERM, nm, bk_s, "msm_bucket_kernel", msm_traffic, BYTES_MSM_TERM),
                "roofline_whole_step": {"frac": MAC32_MSM_TERM * nm / per / VALU_PEAK_MAC32, "achieved": MAC32_MSM_TERM * nm / per / 1e9},
            }
            if do_cpu:
                sm = min(nm, 1 << 15)
                cpu_m, cm_s = cpu_time(lambda: cpuN.g1_msm(pts_h[:sm].tobytes(), msm["scalars"][:sm].tobytes(), 96))
                _, cm1_s = cpu_time(lambda: orc.g1_msm(pts_h[:4096].tobytes(), msm["scalars"][:4096].tobytes(), 96, 1))
                if cpu_m != ctx.g1_msm(pts_h[:sm].tobytes(), msm["scalars"][:sm].tobytes(), 96):
                    raise SystemExit("bench: CPU MSM baseline differs from the GPU product of the same sample")
                result["msm"]["cpu_baseline"] = cpu_b(sm / cm_s, "terms/s", "first 2^15 terms (ECP_muln); equal", 4096 / cm1_s)
            if "sharded" in msm:
                sh = msm["sharded"]
                if not sh["same_on_every_rank"] or sh["equals_single_gpu"] is False:
                    raise SystemExit("bench: sharded MSM results disagree (ranks or single-GPU value) — number withheld")
                result["msm_sharded"] = {"metric": "G1 multi-scalar product terms/s, ONE product of 2^%d terms over %d GPUs" % (args.log2_msm, world),
                                         "value": nm * sh["steps"] / sh["elapsed"], "unit": "terms/s", "scaling": "strong", "steps": sh["steps"],
                                         "ms_per_step": sh["elapsed"] / sh["steps"] * 1e3, "rccl_ranks": sh["rccl_ranks"], "backend": sh["backend"],
                                         "exchange": "all_gather of 96 B per rank + local sum",
                                         "same_on_every_rank": True, "equals_single_gpu": sh["equals_single_gpu"]}

        # ---------------------------------------------------------------- pairing parity first (the split legs compare with its output)
        if pair is not None:
            npair = pair["npair"]
            p1_h = pair["p1"].cpu().numpy().reshape(npair, 96)
            q2_h = pair["q2"].cpu().numpy().reshape(npair, 192)
            gt_h = pair["gt"].cpu().numpy().reshape(npair, 576)
            full = do_cpu and not args.sampled_parity
            pair_one = None
            if full:
                cpu_gt, cpu_ps = cpu_time(lambda: cpuN.pair(p1_h.tobytes(), q2_h.tobytes()))
                cpu_gt = np.frombuffer(cpu_gt, dtype=np.uint8).reshape(npair, 576)
                k1 = min(npair, 512)
                pair_one = k1 / cpu_time(lambda: orc.pair(p1_h[:k1].tobytes(), q2_h[:k1].tobytes(), 1))[1]
                badp = np.nonzero((cpu_gt != gt_h).any(axis=1))[0]
                if len(badp):
                    raise SystemExit("bench: %d of %d GPU pairing results differ from the CPU oracle (first lanes %s) — number withheld"
                                     % (len(badp), npair, badp[:8].tolist()))
                checked, ps = npair, npair
            else:
                pidx = list(range(4)) + [int(x) for x in np.random.Generator(np.random.PCG64(9)).integers(0, npair, size=60)]
                t3 = time.perf_counter()
                cpu_gt = orc.pair(p1_h[pidx].tobytes(), q2_h[pidx].tobytes(), min(cores, 8))
                cpu_ps = time.perf_counter() - t3
                if cpu_gt != gt_h[pidx].tobytes():
                    raise SystemExit("bench: GPU pairing results differ from the CPU oracle — number withheld")
                checked, ps = len(pidx), len(pidx)

        # ---------------------------------------------------------------- G2 multiplication, Miller loop, final exponentiation alone
        if split is not None:
            npair, ng2, st = pair["npair"], split["ng2"], split["steps"]
            # 2^15 lanes with the CPU baseline (seconds of CPU work per leg, not a burst)
            ns = min(npair, (1 << 15) if do_cpu else (1 << 12))
            g2o_h = split["g2_out"].cpu().numpy().reshape(ng2, 192)
            g2i_h = split["g2_in"].cpu().numpy().reshape(ng2, 192)
            mil_h = split["mil"].cpu().numpy().reshape(npair, 576)
            fex_h = split["fex"].cpu().numpy().reshape(npair, 576)
            # lanes 0..4 carry the edge scalars 0, 1, r - 1, r, 2^256 - 1
            q_cpu, g2_s = cpu_time(lambda: cpuN.g2_mul(g2i_h[:ns].tobytes(), split["g2_sc_h"][:ns].tobytes(), 192))
            m_cpu, mil_s = cpu_time(lambda: cpuN.miller_t(p1_h[:ns].tobytes(), q2_h[:ns].tobytes()))
            f_cpu, fx_s = cpu_time(lambda: cpuN.fexp_t(m_cpu))
            g2_one = mil_one = fx_one = None
            if do_cpu:                                             # one thread, own short samples
                g2_one = 1024 / cpu_time(lambda: orc.g2_mul(g2i_h[:1024].tobytes(), split["g2_sc_h"][:1024].tobytes(), 192, 1))[1]
                mil_one = 1024 / cpu_time(lambda: orc.miller_t(p1_h[:1024].tobytes(), q2_h[:1024].tobytes(), 1))[1]
                fx_one = 512 / cpu_time(lambda: orc.fexp_t(m_cpu[:576 * 512], 1))[1]
            if q_cpu != g2o_h[:ns].tobytes():
                raise SystemExit("bench: GPU G2 multiplications differ from the CPU oracle — number withheld")
            if m_cpu != mil_h[:ns].tobytes():
                raise SystemExit("bench: GPU Miller values differ from the CPU oracle — number withheld")
            if f_cpu != fex_h[:ns].tobytes() or not (fex_h == gt_h).all():
                raise SystemExit("bench: GPU final exponentiations differ from the CPU oracle / the pairing outputs — number withheld")
            queued = (npair + 20) // 21 > 2048

            def leg(metric, unit, units, el, kprof, mac, nbytes, kernel, workload, parity, cpu_s, tr=None, clock_key=None, one=None):
                k_s = kprof[0] / max(kprof[1], 1) * 1e-3
                lps = max(kprof[1] / max(st, 1), 1)
                d = {"metric": metric, "value": world * units * st / el, "unit": unit, "steps": st, "ms_per_step": el / st * 1e3,
                     "workload": workload, "parity": parity,
                     "roofline": valu(mac, units / lps, k_s, kernel, tr, nbytes, clock_key, units_per_launch=units / lps)}
                if do_cpu:
                    d["cpu_baseline"] = cpu_b(ns / cpu_s, unit, "first %d lanes" % ns, one)
                return d
            result["g2_mul"] = leg("G2 scalar-muls/s per MI355X (batch 2^%d per GPU), bit-exact vs CPU" % args.log2_g2, "scalar-muls/s", ng2, split["g2_el"],
                                   split["g2k"], MAC32_G2_MUL, BYTES_G2_MUL, "g2_mul2_kernel",
                                   "PAIR_G2mul, 192-B affine in/out, edge scalars in lanes 0..4",
                                   par(checked_lanes=ns, of=ng2), g2_s,
                                   None if g2_traffic is None else g2_traffic * ng2 / max(split["g2k"][1] / max(st, 1), 1), one=g2_one)
            result["miller"] = leg("Miller loops/s per MI355X (batch 2^%d per GPU), the reference's field element" % args.log2_pairings, "Miller loops/s", npair,
                                   split["mil_el"], split["milk"], MAC32_MILLER, 96 + 192 + 576, "miller3_queue_kernel" if queued else "miller3_kernel",
                                   "PAIR_ate on the pairing leg's inputs", par(checked_lanes=ns, of=npair), mil_s, mil_traffic if queued else None, "miller", mil_one)
            result["fexp"] = leg("final exponentiations/s per MI355X (batch 2^%d per GPU)" % args.log2_pairings, "final exponentiations/s", npair,
                                 split["fex_el"], split["fexk"], MAC32_FEXP, 2 * 576, "fexp3_queue_kernel" if queued else "gt3_op_kernel",
                                 "PAIR_fexp on those Miller values",
                                 par(checked_lanes=ns, of=npair, check="+ every lane equals the pairing leg's output"), fx_s, fex_traffic if queued else None, "fexp", fx_one)

        # ---------------------------------------------------------------- BBS+ (decoded inputs, then the wire formats end to end)
        if bbs is not None:
            nb = bbs["n"]
            per = bbs["elapsed"] / bbs["steps"]
            result["bbs_plus"] = {
                "metric": "BBS+ signature verifications/s (2^%d per GPU, 1 message block), decoded inputs" % args.log2_bbs,
                "value": world * nb / per, "unit": "verifications/s", "steps": bbs["steps"], "ms_per_step": per * 1e3,
                "workload": "configs[4]: setup(16), 1-block messages, real signatures, every 1009th message corrupted",
                "parity": {"check": "all verdicts equal the construction", "bit_exact": True},
                # the pipeline's OWN operation sequence (tools/count_ops.py -> profiles/r03_op_counts.json) over the wall time of the whole pipeline
                "roofline": valu(MAC32_BBS_PIPELINE, nb, per, "pair3_prod_fixed_queue_kernel", bbs_traffic, BYTES_BBS_VERIFY,
                                 issue_secs=bbs["pair_ms"] / max(bbs["pair_launches"], 1) * 1e-3,
                                 pair_kernel_avg_ms=bbs["pair_ms"] / max(bbs["pair_launches"], 1), time_base="pipeline; issue: pairing kernel"),
                "reference_sequence_gmac32_per_s": MAC32_BBS_VERIFY * nb / per / 1e9,
            }
            wr = bbs["wire"]
            wper = wr["elapsed"] / bbs["steps"]
            result["bbs_plus_wire"] = {
                "metric": "BBS+ verifications/s from wire formats (2^%d per GPU: 145-B signatures, %d-B messages)" % (args.log2_bbs, BBS_MSG_LEN),
                "value": world * nb / wper, "unit": "verifications/s", "steps": bbs["steps"], "ms_per_step": wper * 1e3,
                "workload": "configs[4] from verify()'s own input forms (bbs+.cpp:57-73)",
                "parity": {"check": "all verdicts equal the construction", "bit_exact": True},
                "roofline": {"kernel": "whole pipeline", "achieved": MAC32_BBS_WIRE_PIPELINE * nb / wper / 1e9,
                             "frac": MAC32_BBS_WIRE_PIPELINE * nb / wper / VALU_PEAK_MAC32, "traffic": None,
                             "mac32_per_unit": MAC32_BBS_WIRE_PIPELINE, "hbm_GBs": BYTES_BBS_WIRE * nb / wper / 1e9},
            }
            if do_cpu:
                sb = 1 << 11
                pg1, pg2, ph0, ph, pw = bbs["pub"]
                sl = np.r_[0:sb - 64, nb - 64:nb]                       # includes corrupted lanes (7, 1016, ...)
                cpu_ok, cb_s = cpu_time(lambda: cpuN.bbs_plus_verify(pg1, pg2, ph0, ph, pw, bbs["A"][sl].tobytes(), bbs["x"][sl].tobytes(), bbs["r"][sl].tobytes(),
                                                                     bbs["m"][sl].tobytes()))
                s1 = sl[:256]
                cb_one = len(s1) / cpu_time(lambda: orc.bbs_plus_verify(pg1, pg2, ph0, ph, pw, bbs["A"][s1].tobytes(), bbs["x"][s1].tobytes(),
                                                                        bbs["r"][s1].tobytes(), bbs["m"][s1].tobytes(), 1))[1]
                if cpu_ok != bbs["ok"][sl].tobytes():
                    raise SystemExit("bench: CPU BBS+ verdicts differ from the GPU verdicts")
                result["bbs_plus"]["parity"].update(par(oracle_lanes=int(len(sl))))
                result["bbs_plus"]["cpu_baseline"] = cpu_b(len(sl) / cb_s, "verifications/s", "%d signatures incl. corrupted; verdicts equal" % len(sl), cb_one)
                cpu_okw, cw_s = cpu_time(lambda: cpuN.bbs_plus_verify_wire(wr["pp"], wr["h49"], wr["pk"], wr["sig"][sl].tobytes(), wr["raw"][sl].tobytes(), BBS_MSG_LEN))
                cw_one = len(s1) / cpu_time(lambda: orc.bbs_plus_verify_wire(wr["pp"], wr["h49"], wr["pk"], wr["sig"][s1].tobytes(), wr["raw"][s1].tobytes(), BBS_MSG_LEN, 1))[1]
                if cpu_okw != wr["ok"][sl].tobytes():
                    raise SystemExit("bench: CPU wire-format BBS+ verdicts differ from the GPU verdicts")
                result["bbs_plus_wire"]["parity"].update(par(oracle_lanes=int(len(sl))))
                result["bbs_plus_wire"]["cpu_baseline"] = cpu_b(len(sl) / cw_s, "verifications/s", "the same %d from their bytes; verdicts equal" % len(sl), cw_one)
            if "sharded" in bbs:
                sh = bbs["sharded"]
                if sh["accepted"] != sh["expected_accepted"]:
                    raise SystemExit("bench: sharded BBS+ leg accepted %d signatures, expected %d" % (sh["accepted"], sh["expected_accepted"]))
                result["bbs_plus_sharded"] = {"metric": "BBS+ verifications/s, 2^%d signatures split over %d GPUs" % (args.log2_bbs, world),
                                              "value": nb * sh["steps"] / sh["elapsed"], "unit": "verifications/s", "scaling": "strong", "steps": sh["steps"],
                                              "ms_per_step": sh["elapsed"] / sh["steps"] * 1e3, "accepted": sh["accepted"],
                                              "exchange": "none"}
        if extras:
            result["extra_configs"] = extras
        if streamed:
            result["streamed"] = {"how": "two contexts alternate (INTEGRATION.md); outputs equal the serial legs'",
                                  "ms_per_step": {k: v[1] for k, v in streamed.items()},
                                  "g1_per_s": world * streamed["g1"][0]}
            if "pairing" in streamed:
                result["streamed"]["pairings_per_s"] = world * streamed["pairing"][0]
        result["notes"] = {
            "roofline": "int-valu binds: SURVEY 8(d) MAC32 / avg launch time (HIP events) vs the v_mad_i64_i32 issue rate measured "
                        "in-kernel (62.06 lanes/clk/CU x 256 CUs x 2.4 GHz); hbm_GBs = algorithmic bytes / same time (peak 8000); "
                        "traffic = FETCH_SIZE x2 + WRITE_SIZE passes (profiles/traffic.json)",
            "issue": "issue_ms = SQ_INSTS_VALU per launch (profiles/issue.json) x 4.06 cycles / 1024 SIMDs / the clock held inside the kernel in THIS "
                     "run (one sampler per XCD, mean; xcd_clock_min_max); ref_mac32_per_inst = reference MAC32 per issued lane-instruction",
            "bbs_plus": "MAC32 = the pipeline's own op sequence (wire: + decode of A); reference_sequence_gmac32_per_s: the reference's over this time",
            "cpu_baseline": "%s, %d worker processes (not threads: oracle/pool.py), same inputs; under 2 s: median of 3; eff_cores = value / one_thread (own sample)"
                            % ("oracle/_ref = the reference's sources compiled here" if pinned else "C port (oracle/_ref absent)", cores),
        }
        # the pairing leg goes LAST: the second half of BASELINE's metric survives any truncation of the line's head
        if pair is not None:
            avg_s = pair["kernel_ms"] / max(pair["launches"], 1) * 1e-3
            kname = "pair3_queue_kernel" if (npair + 20) // 21 > 2048 else "pair3_kernel"
            result["pairing"] = {
                "metric": "ate pairings/s per MI355X (batch 2^%d per GPU), bit-exact vs CPU" % args.log2_pairings,
                "value": world * npair * pair["steps"] / pair["elapsed"], "unit": "pairings/s", "steps": pair["steps"],
                "ms_per_step": pair["elapsed"] / pair["steps"] * 1e3,
                "workload": "configs[2]: 2^%d pairings e(P_i, Q_i) -> 576-B GT each, per GPU" % args.log2_pairings,
                "parity": par(head=True, checked_lanes=checked, of=npair),
                "roofline": valu(MAC32_PAIRING, npair, avg_s, kname, pair_traffic, BYTES_PAIRING, "pair", head=True),
            }
            if do_cpu:
                result["pairing"]["cpu_baseline"] = cpu_b(ps / cpu_ps, "pairings/s", ("all %d pairings; every lane equal" % ps) if full
                                                          else "%d sampled lanes" % ps, pair_one, full_keys=True)
        line = json.dumps(compact(result), separators=(",", ":"))
        print(line, flush=True)
    if cpu_pool is not None:
        cpu_pool.close()
    if dist:
        dist.barrier()
    ctx.close()
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
