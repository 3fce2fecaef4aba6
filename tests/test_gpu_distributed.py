"""GPU test of the sharded MSM: two ranks share the one GPU of the test box (each with its own context),
exchange the partial points over gloo and must both obtain the oracle's result."""
import os
import socket

import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from util import golden, scalars

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
