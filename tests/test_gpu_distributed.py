"""GPU test of the sharded MSM: two ranks share the one GPU of the test box (each with its own context),
exchange the partial points over gloo and must both obtain the oracle's result."""
import os
import socket

import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from util import golden, scalars

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, pts, sc, expect, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from crypto12381_amd import Context
        from crypto12381_amd.distributed import msm_sharded, shard_bytes
        ctx = Context(0)
        res = msm_sharded(lambda p, s, fmt: ctx.g1_msm(p, s, fmt),
                          shard_bytes(pts, 96, rank, world), shard_bytes(sc, 32, rank, world), 49)
        ctx.close()
        q.put((rank, res == expect))
    finally:
        dist.destroy_process_group()


def test_msm_sharded_two_ranks_on_gpu(oracle_port):
    from crypto12381_amd import Context
    n = 3001
    g1 = bytes.fromhex(golden("g1")["generator"])
    c = Context(0)
    pts = c.g1_mul(g1 * n, scalars(601, n), 96)
    sc = scalars(602, n)
    single = c.g1_msm(pts, sc, 49)
    c.close()
    expect = oracle_port.g1_msm(pts, sc, 49, 16)
    assert single == expect
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, pts, sc, expect, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    got = dict(q.get(timeout=5) for _ in range(2))
    assert got == {0: True, 1: True}


def test_msm_multi_one_process_several_contexts(oracle_port):
    """c12381_g1_msm_multi: one host process, one context per GPU; on the one-GPU test box the three contexts share
    device 0 (separate streams and workspaces), which exercises the same splitting, threading and combine."""
    from crypto12381_amd import Context
    from crypto12381_amd.capi import g1_msm_multi
    n = 9001                                       # 3 shards, bucket path in each (>= 2^12 would need 12288; mixed paths below)
    g1 = bytes.fromhex(golden("g1")["generator"])
    ctxs = [Context(0) for _ in range(3)]
    pts = ctxs[0].g1_mul(g1 * n, scalars(611, n), 96)
    sc = scalars(612, n)
    expect = oracle_port.g1_msm(pts, sc, 49, 16)
    assert g1_msm_multi(ctxs, pts, sc, 49) == expect
    assert g1_msm_multi(ctxs[:1], pts, sc, 49) == expect
    m = 3 * 5000                                   # every shard above the bucket threshold
    pts2 = (pts * 2)[:96 * m]
    sc2 = scalars(613, m)
    assert g1_msm_multi(ctxs, pts2, sc2, 96) == ctxs[1].g1_msm(pts2, sc2, 96)
    assert g1_msm_multi(ctxs, pts[:96 * 2], sc[:64], 49) == oracle_port.g1_msm(pts[:96 * 2], sc[:64], 49, 1)   # a shard with 0 terms
    assert g1_msm_multi(ctxs, b"", b"", 49) == bytes(49)
    for c in ctxs:
        c.close()


def test_msm_in_parts(oracle_port):
    """Products above the per-pass limit are cut into parts (limit lowered through the environment in a child process)."""
    import subprocess
    import sys
    code = (
        "import sys; sys.path.insert(0, 'tests'); sys.path.insert(0, '.')\n"
        "import tools.libsel\n"
        "from util import golden, scalars\n"
        "from crypto12381_amd import Context\n"
        "g1 = bytes.fromhex(golden('g1')['generator']); n = 11000\n"
        "c = Context(0); pts = c.g1_mul(g1 * n, scalars(621, n), 96); sc = scalars(622, n)\n"
        "print(c.g1_msm(pts, sc, 49).hex())\n")
    env = dict(os.environ)
    ref = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stdout.strip().splitlines()[-1]
    # the per-pass limit (2^26 terms) can be lowered only in the experiments build — same sources, same multi-part code
    env["C12381_MSM_MAX_TERMS"] = "4500"
    env["C12381_LIB"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "crypto12381_amd", "lib", "libc12381_hip_exp.so")
    parts = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stdout.strip().splitlines()[-1]
    assert parts == ref
    from crypto12381_amd import Context
    g1 = bytes.fromhex(golden("g1")["generator"])
    c = Context(0)
    n = 11000
    pts, sc = c.g1_mul(g1 * n, scalars(621, n), 96), scalars(622, n)
    c.close()
    assert ref == oracle_port.g1_msm(pts, sc, 49, 16).hex()


RCCL_CODE = r"""
import os, sys
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import tools.libsel  # C12381_LIB -> capi.use_library
import torch
import torch.distributed as dist
from util import golden, scalars
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29544')
torch.cuda.set_device(0)
dev = torch.device('cuda', 0)
dist.init_process_group(backend='nccl', rank=0, world_size=1, device_id=dev)
from crypto12381_amd import Context
from crypto12381_amd.distributed import msm_sharded_tensors
ctx = Context(0)
stream = torch.cuda.Stream(device=dev)
ctx.set_stream(stream.cuda_stream)
n = 5000
g1 = bytes.fromhex(golden('g1')['generator'])
pts = ctx.g1_mul(g1 * n, scalars(631, n), 96)
sc = scalars(632, n)
tp = torch.frombuffer(bytearray(pts), dtype=torch.uint8).to(dev)
ts = torch.frombuffer(bytearray(sc), dtype=torch.uint8).to(dev)
def local_t(p, s, fmt):
    o = torch.empty(fmt, dtype=torch.uint8, device=dev)
    ctx.g1_msm_dev(p.numel() // 96, p.data_ptr(), s.data_ptr(), o.data_ptr(), fmt)
    return o
# the caller stays on torch's default stream; the function orders the context's stream against it (stream=...)
tp2 = tp.clone() ^ 0                       # produced on the default stream just before the call
res = msm_sharded_tensors(local_t, tp2, ts, 49, stream=stream)
res = res.clone()                          # consumed on the default stream right after
ctx.sync()
print('RESULT', bytes(res.cpu().numpy().tobytes()).hex())
print('EXPECT', ctx.g1_msm(pts, sc, 49).hex())
dist.destroy_process_group()
"""


def test_rccl_exchange_on_device_tensors_single_rank():
    """The device-resident sharded MSM through the nccl (= RCCL) backend: one rank on the one GPU of the test box — the partial
    point goes through all_gather_into_tensor on a device tensor and the context's stream exactly as on the 8-GPU node."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, "-c", RCCL_CODE], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = dict(l.split(" ", 1) for l in r.stdout.splitlines() if l.startswith(("RESULT", "EXPECT")))
    assert lines["RESULT"] == lines["EXPECT"]


def test_bench_two_ranks_on_one_gpu_dry_run():
    """`bench.py --gpus 2` as the driver launches it (torch.distributed.run, one rank per process), rehearsed on the ONE GPU of the test
    box with the gloo backend and small batches: the strong-scaled legs exist, the sharded MSM equals the single-GPU value on every rank,
    both ranks took part, and the rank-local preparation (exponent sum over all terms in Python, signature generation) stays small."""
    import json
    import subprocess
    import sys
    import time
    env = dict(os.environ)
    env["C12381_BENCH_BACKEND"] = "gloo"
    env.pop("C12381_LIB", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29547",
           "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "1", "--log2-batch", "14", "--log2-pairings", "10", "--log2-g2", "12", "--log2-msm", "14",
           "--log2-bbs", "12", "--no-cpu-baseline", "--sampled-parity"]
    t0 = time.perf_counter()
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    wall = time.perf_counter() - t0
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["scaling"] == "weak"
    assert d["msm_sharded"]["equals_single_gpu"] is True and d["msm_sharded"]["same_on_every_rank"] is True and d["msm_sharded"]["rccl_ranks"] == 2
    assert d["bbs_plus_sharded"]["accepted"] > 0
    for leg in ("pairing", "g2_mul", "miller", "fexp", "msm", "bbs_plus"):
        assert leg in d and d[leg]["value"] > 0, leg
    assert wall < 600, "the two-rank dry run took %.0f s" % wall


def test_bench_gpus_flag_starts_the_ranks_itself():
    """Exactly `python3 bench.py --gpus 2 ...` — no torchrun wrapper, the shape of the driver's N = 1 command: the script must become the
    launcher of two ranks (gloo on the ONE test GPU), not run one rank and print n_gpus 1 (missing 1); with a launcher whose
    world size contradicts --gpus it must stop instead of reporting the wrong GPU count."""
    import json
    import subprocess
    import sys
    env = dict(os.environ)
    env["C12381_BENCH_BACKEND"] = "gloo"
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    small = ["--steps", "1", "--warmup", "1", "--log2-batch", "14", "--log2-pairings", "10", "--log2-g2", "12", "--log2-msm", "14", "--log2-bbs", "12",
             "--no-cpu-baseline", "--sampled-parity", "--no-clock-probe"]
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2"] + small, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 prints ONE JSON line, got %d" % len(lines)
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2
    assert d["msm_sharded"]["equals_single_gpu"] is True and d["msm_sharded"]["rccl_ranks"] == 2
    assert d["bbs_plus_sharded"]["accepted"] > 0
    assert list(d)[-1] == "pairing" and d["pairing"]["value"] > 0          # the second half of BASELINE's metric closes the line
    assert d["bbs_plus_wire"]["value"] > 0
    assert len(lines[0]) < 8000, "the JSON line outgrew the driver's log tail: %d characters" % len(lines[0])
    # a launcher that started ONE rank for --gpus 2: refuse
    env1 = dict(env); env1.update({"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2"] + small, cwd=ROOT, env=env1, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "refusing" in (r.stdout + r.stderr)


def test_bench_single_gpu_line_with_cpu_baseline_and_streamed_legs():
    """`python3 bench.py` at N = 1 as the driver runs it (small sizes here): the CPU baseline comes from the process pool that is forked
    before the GPU is touched (oracle/pool.py), every leg carries its own baseline with a one-thread rate, the streamed legs are present and
    equal to the serial ones (bench.py withholds the line otherwise), `pairing` closes the line and the line fits the driver's log tail."""
    import json
    import subprocess
    import sys
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    small = ["--steps", "2", "--warmup", "1", "--log2-batch", "14", "--log2-pairings", "12", "--log2-g2", "12", "--log2-msm", "14", "--log2-bbs", "12"]
    r = subprocess.run([sys.executable, "bench.py"] + small, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["value"] > 0
    cb = d["cpu_baseline"]
    assert cb["workers"] == "processes" and cb["kind"] in ("reference", "port") and cb["cores"] >= 1
    assert cb["value"] > 0 and cb["one_thread"] > 0 and cb["threads_value"] > 0
    assert d["parity"]["bit_exact"] is True and d["parity"]["oracle"] == cb["kind"]
    assert d["roofline"]["frac"] > 0 and d["roofline"]["bound"] == "int-valu"
    for leg in ("msm", "g2_mul", "miller", "fexp", "bbs_plus", "bbs_plus_wire", "pairing"):
        assert d[leg]["value"] > 0 and d[leg]["cpu_baseline"]["value"] > 0 and d[leg]["cpu_baseline"]["one_thread"] > 0, leg
        assert d[leg]["parity"]["bit_exact"] is True and d[leg]["parity"]["pinned"] == d["parity"]["pinned"], leg
    assert set(d["streamed"]["ms_per_step"]) == {"g1", "pairing", "miller", "fexp", "g2_mul", "bbs_plus"}
    assert d["streamed"]["g1_per_s"] > 0 and d["streamed"]["pairings_per_s"] > 0
    assert list(d)[-1] == "pairing"
    assert len(lines[0]) < 8000, "the JSON line outgrew the driver's log tail: %d characters" % len(lines[0])
