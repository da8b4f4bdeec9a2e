"""Data-parallel sharding of the E-step over the GPUs of one node.

The N rows shard contiguously, one process per GPU; every N-dependent output of the step is a sum over rows
(reference src/models/tsvgp.py:278-281 and :95), so the only exchange is ONE all-reduce (sum) per step of the
packed accumulator  [acc2 (P*M*M) | acc1 (P*M) | sum ve | #non-positive var | rows]  in fp64 --
RCCL over xGMI with backend "nccl", gloo on CPU for tests.  The M x M prelude/epilogue runs replicated.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def shard_bounds(n_rows: int, world: int = None, r: int = None):
    """Contiguous row block [lo, hi) of rank r: the first (n_rows % world) ranks get one extra row."""
    world = world_size() if world is None else world
    r = rank() if r is None else r
    base, extra = divmod(int(n_rows), world)
    lo = r * base + min(r, extra)
    return lo, lo + base + (1 if r < extra else 0)


def shard_rows(*arrays, world: int = None, r: int = None):
    """Slices every array (same leading dimension) to this rank's contiguous row block."""
    lo, hi = shard_bounds(arrays[0].shape[0], world, r)
    out = tuple(a[lo:hi] for a in arrays)
    return out if len(out) > 1 else out[0]


def pack_stats(stats, with_sites: bool) -> torch.Tensor:
    parts = []
    if with_sites:
        parts += [stats.acc2.reshape(-1), stats.acc1.reshape(-1)]
    dev = stats.ve_sum.device
    parts += [stats.ve_sum.reshape(1).to(torch.float64), stats.nonpos.reshape(1).to(torch.float64),
              torch.full((1,), float(stats.n_rows), dtype=torch.float64, device=dev)]  # a fill kernel, not a host copy
    return torch.cat([p.to(torch.float64) for p in parts])


def unpack_stats(packed: torch.Tensor, P: int, M: int, with_sites: bool):
    o = 0
    acc2 = acc1 = None
    if with_sites:
        acc2 = packed[o:o + P * M * M].reshape(P, M, M)
        o += P * M * M
        acc1 = packed[o:o + P * M].reshape(P, M)
        o += P * M
    return acc2, acc1, packed[o], packed[o + 1], packed[o + 2]


def all_reduce_sum(packed: torch.Tensor) -> torch.Tensor:
    """In-place sum over ranks (no-op for a single process)."""
    if world_size() > 1:
        dist.all_reduce(packed, op=dist.ReduceOp.SUM)
    return packed


def broadcast_from_rank0(t: torch.Tensor) -> torch.Tensor:
    """In-place broadcast of rank 0's values (no-op for a single process).  Used for host-side DECISIONS derived from
    replicated data (the projection route from cond(K_uu)): every rank must take the same branch even if a library
    routine were to round differently from one device to the next."""
    if world_size() > 1:
        dist.broadcast(t, src=0)
    return t
