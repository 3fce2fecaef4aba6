"""CPU test of the N>1 path: world_size-2 gloo run of the MSM sharding + all-gather + combine logic in
crypto12381_amd/distributed.py.  The arithmetic is injected from the oracle (no GPU here); on the GPU box the
same function is driven by Context.g1_msm (tests/test_gpu_distributed.py)."""
import os
import socket

import torch.distributed as dist
import torch.multiprocessing as mp

from util import cat, golden, scalars


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
        from crypto12381_amd.distributed import msm_sharded, shard_bytes
        from oracle.bindings import Oracle
        orc = Oracle("port")
        res = msm_sharded(lambda p, s, fmt: orc.g1_msm(p, s, fmt, 1),
                          shard_bytes(pts, 96, rank, world), shard_bytes(sc, 32, rank, world), 49)
        # tensor form (the bench's strong-scaled MSM leg uses it with device tensors and the nccl backend)
        import torch
        from crypto12381_amd.distributed import msm_sharded_tensors

        def local_t(p, s, fmt):
            return torch.frombuffer(bytearray(orc.g1_msm(p.numpy().tobytes(), s.numpy().tobytes(), fmt, 1)), dtype=torch.uint8)
        tp = torch.frombuffer(bytearray(shard_bytes(pts, 96, rank, world)), dtype=torch.uint8)
        ts = torch.frombuffer(bytearray(shard_bytes(sc, 32, rank, world)), dtype=torch.uint8)
        res_t = msm_sharded_tensors(local_t, tp, ts, 49).numpy().tobytes()
        # with an injected combine (on the GPU box c12381_g1_sum_dev; here the oracle's unit-scalar product plays the sum)
        one = (1).to_bytes(32, "big")
        res_c = msm_sharded(lambda p, s, fmt: orc.g1_msm(p, s, fmt, 1), shard_bytes(pts, 96, rank, world), shard_bytes(sc, 32, rank, world), 49,
                            combine=lambda p, fmt: orc.g1_msm(p, one * (len(p) // 96), fmt, 1))
        res_ct = msm_sharded_tensors(local_t, tp, ts, 49, combine_t=lambda p, fmt: local_t(
            p, torch.frombuffer(bytearray(one * (p.numel() // 96)), dtype=torch.uint8), fmt)).numpy().tobytes()
        q.put((rank, res == expect and res_t == expect and res_c == expect and res_ct == expect))
    finally:
        dist.destroy_process_group()


def test_msm_sharded_two_ranks(oracle_port):
    g = golden("g1")
    pts, sc = cat(g["points"]), cat(g["scalars"])
    n = 21                                   # odd size: ragged shards (10 + 11)
    pts, sc = pts[:96 * n], sc[:32 * n]
    expect = oracle_port.g1_msm(pts, sc, 49, 2)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, pts, sc, expect, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got = dict(q.get(timeout=5) for _ in range(2))
    assert got == {0: True, 1: True}


def test_shard_bounds_cover_everything():
    from crypto12381_amd.distributed import shard_bounds
    for n in (0, 1, 7, 1 << 20, (1 << 22) + 3):
        for world in (1, 2, 4, 8):
            cuts = [shard_bounds(n, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
            assert max(hi - lo for lo, hi in cuts) - min(hi - lo for lo, hi in cuts) <= 1
