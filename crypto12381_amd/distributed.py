"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over
xGMI on the GPU box, "gloo" in the CPU tests).

* Independent units (G1/G2 scalar-mul batches, pairings, BBS+ verifies): contiguous shards, NO collective.
* MSM  Π g_i^{x_i}: shard by terms, local MSM per rank, then ONE exchange — an all-gather of the partial
  points (96 B each) and a local sum on every rank.  The combine operator is elliptic-curve addition, which
  no RCCL reduce op implements, so "all-reduce" is realised as all-gather + local N-term sum.

The arithmetic itself is injected (`local_msm`): on the GPU box it is `Context.g1_msm`; the CPU tests inject
the oracle to exercise exactly this sharding/exchange logic without a GPU.
"""
from __future__ import annotations

from typing import Callable

import torch
import torch.distributed as dist


def shard_bounds(n: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous shard [lo, hi) of n units for `rank` (sizes differ by at most one)."""
    return n * rank // world, n * (rank + 1) // world


def shard_bytes(buf: bytes, record: int, rank: int, world: int) -> bytes:
    lo, hi = shard_bounds(len(buf) // record, rank, world)
    return buf[record * lo:record * hi]


def msm_sharded(local_msm: Callable[[bytes, bytes, int], bytes], pts_shard: bytes, scalars_shard: bytes,
                out_fmt: int = 49, group=None, device: torch.device | str = "cpu", combine: Callable[[bytes, int], bytes] | None = None) -> bytes:
    """Every rank passes ITS shard of (points, scalars); every rank returns the full product.

    local_msm(points96, scalars32, fmt) -> fmt bytes  (fmt 96 = affine, all-zero = infinity).
    combine(points96, fmt) -> the plain sum of the N partial points (on the GPU box Context.g1_sum: a lift and a tree sum, tens of
    microseconds); without it the sum is a product with unit scalars through local_msm (the CPU tests, where only an MSM is injected).
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    partial = local_msm(pts_shard, scalars_shard, 96) if len(pts_shard) else bytes(96)
    if world == 1:
        gathered = partial
    else:
        mine = torch.frombuffer(bytearray(partial), dtype=torch.uint8).to(device)
        parts = [torch.empty(96, dtype=torch.uint8, device=device) for _ in range(world)]
        dist.all_gather(parts, mine, group=group)                 # 96 B per rank: latency-bound, one exchange
        gathered = b"".join(bytes(p.cpu().numpy().tobytes()) for p in parts)
    if combine is not None:
        return combine(gathered, out_fmt)
    one = (1).to_bytes(32, "big")
    return local_msm(gathered, one * (len(gathered) // 96), out_fmt)


def msm_sharded_tensors(local_msm_t: Callable[[torch.Tensor, torch.Tensor, int], torch.Tensor], pts_shard: torch.Tensor,
                        scalars_shard: torch.Tensor, out_fmt: int = 49, group=None, stream=None,
                        combine_t: Callable[[torch.Tensor, int], torch.Tensor] | None = None) -> torch.Tensor:
    """Device-resident form of msm_sharded: the partial point never visits the host.

    local_msm_t(points96 uint8 tensor, scalars32 uint8 tensor, fmt) -> uint8 tensor of fmt bytes on the same device
    (on the GPU box: c12381_g1_msm_dev on the context's stream; the CPU tests inject the oracle on CPU tensors).
    The exchange is ONE all_gather_into_tensor of 96 B per rank on the tensors' own device — RCCL over xGMI when the
    process group is "nccl" —, the combine is the local sum of the N partial points — combine_t(points96 tensor, fmt) when given
    (c12381_g1_sum_dev: lift + tree sum), else their product with unit scalars through local_msm_t — so every rank ends with
    the same bytes.

    Stream ordering (device tensors): `stream` is the torch.cuda.Stream the library context enqueues on
    (Context.set_stream(stream.cuda_stream)).  The function makes that stream wait for the caller's current stream (the
    inputs), runs the local products, the allocation of the gather buffer, the collective and the fill of the unit scalars
    under `torch.cuda.stream(stream)` — so all of them are ordered with the library's kernels — and makes the caller's stream
    wait for the result.  Without `stream` the caller vouches that local_msm_t enqueues on torch's current stream."""
    dev = pts_shard.device
    if stream is not None and dev.type == "cuda":
        caller = torch.cuda.current_stream(dev)
        stream.wait_stream(caller)
        with torch.cuda.stream(stream):
            out = _msm_sharded_tensors(local_msm_t, pts_shard, scalars_shard, out_fmt, group, combine_t)
        caller.wait_stream(stream)
        return out
    return _msm_sharded_tensors(local_msm_t, pts_shard, scalars_shard, out_fmt, group, combine_t)


def _msm_sharded_tensors(local_msm_t, pts_shard, scalars_shard, out_fmt, group, combine_t):
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    dev = pts_shard.device
    if pts_shard.numel():
        partial = local_msm_t(pts_shard, scalars_shard, 96)
    else:
        partial = torch.zeros(96, dtype=torch.uint8, device=dev)          # empty shard: the point at infinity
    if not dist.is_initialized():
        gathered = partial
    else:                                                     # also with one rank: the exchange is the same call at every size
        gathered = torch.empty(96 * world, dtype=torch.uint8, device=dev)
        dist.all_gather_into_tensor(gathered, partial.contiguous(), group=group)
    if combine_t is not None:
        return combine_t(gathered, out_fmt)
    ones = torch.zeros(world, 32, dtype=torch.uint8, device=dev)
    ones[:, 31] = 1
    return local_msm_t(gathered, ones.reshape(-1), out_fmt)
