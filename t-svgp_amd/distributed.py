"""Data-parallel sharding of the E-step over the GPUs of one node.

The N rows shard contiguously, one process per GPU; every N-dependent output of the step is a sum over rows
(reference src/models/tsvgp.py:278-281 and :95), so the only exchange is ONE all-reduce (sum) per step of the
packed accumulator  [lower triangles of acc2 (P*M*(M+1)/2) | acc1 (P*M) | sum ve | #non-positive var | rows]  in fp64 --
RCCL over xGMI with backend "nccl", gloo on CPU for tests.  The M x M prelude/epilogue runs replicated.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


# Issue the collectives whenever a process group exists, also when it has ONE rank (default: a single rank passes its
# tensors through).  For tests: a world_size-1 "nccl" group on the one GPU of a test box drives the real RCCL calls -- group
# initialisation with a device id, stream ordering against the kernels around them, capture of the step as two graphs
# around the all-reduce -- that otherwise only an 8-GPU node would reach.  Also settable as TSVGP_FORCE_COLLECTIVES=1.
FORCE_COLLECTIVES = os.environ.get("TSVGP_FORCE_COLLECTIVES", "0") == "1"


# Timing of the collectives (bench.py's multi-rank lines): set TIMING to a list and every collective below appends
# (kind, bytes, start event, stop event, host seconds) -- HIP events on the stream the collective is ordered against (the
# current stream: a synchronous torch.distributed call makes it wait for the backend's own stream), from a pool filled by
# ``reserve_timing`` so that no event is created inside a timed region.  None (default): no events, no overhead.
TIMING = None
_EVENT_POOL = []


def reserve_timing(n_collectives: int):
    while len(_EVENT_POOL) < 2 * n_collectives:
        _EVENT_POOL.append(torch.cuda.Event(enable_timing=True))


class _timed:
    def __init__(self, kind: str, t: torch.Tensor):
        self.on = TIMING is not None and t.is_cuda
        self.kind, self.nbytes = kind, t.numel() * t.element_size()

    def __enter__(self):
        if self.on:
            import time

            self.e0 = _EVENT_POOL.pop() if _EVENT_POOL else torch.cuda.Event(enable_timing=True)
            self.e1 = _EVENT_POOL.pop() if _EVENT_POOL else torch.cuda.Event(enable_timing=True)
            self.e0.record()
            self.t0 = time.perf_counter()

    def __exit__(self, *exc):
        if self.on:
            import time

            host = time.perf_counter() - self.t0
            self.e1.record()
            TIMING.append((self.kind, self.nbytes, self.e0, self.e1, host))


def timing_summary(entries, steps: int):
    """{kind: {count_per_step, bytes, ms: {mean, min, max}, host_ms_mean}} of TIMING entries (after a synchronize)."""
    out = {}
    for kind, nbytes, e0, e1, host in entries:
        d = out.setdefault(kind, {"n": 0, "bytes": nbytes, "ms": [], "host": 0.0})
        d["n"] += 1
        d["ms"].append(e0.elapsed_time(e1))
        d["host"] += host * 1e3
    return {k: {"count_per_step": d["n"] / max(steps, 1), "payload_bytes": d["bytes"],
                "ms": {"mean": sum(d["ms"]) / len(d["ms"]), "min": min(d["ms"]), "max": max(d["ms"])},
                "host_ms_mean": d["host"] / d["n"]} for k, d in out.items()}


def collectives_on() -> bool:
    """True when all_reduce_sum / broadcast_from_rank0 go to the backend: more than one rank, or a forced single rank."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or FORCE_COLLECTIVES


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


def _tri_indices(M: int, device):
    idx = torch.tril_indices(M, M, device=device)
    return idx[0], idx[1]


def pack_stats(stats, with_sites: bool, eng=None) -> torch.Tensor:
    """[lower triangles of acc2 (P * M (M + 1) / 2) | acc1 (P * M) | sum ve | #non-positive var | rows] in fp64: acc2 is
    symmetric, so only one triangle travels (4.2 MB instead of 8.4 MB at M = 1024, P = 1).  ``eng``: an engine with
    ``sym_pack`` (the HIP pack kernel); without it the triangle is gathered by index."""
    dev = stats.ve_sum.device
    tail = [stats.ve_sum.reshape(1).to(torch.float64), stats.nonpos.reshape(1).to(torch.float64),
            torch.full((1,), float(stats.n_rows), dtype=torch.float64, device=dev)]  # a fill kernel, not a host copy
    if not with_sites:
        return torch.cat(tail)
    P, M = stats.acc2.shape[0], stats.acc2.shape[-1]
    tri = M * (M + 1) // 2
    out = torch.empty(P * tri + P * M + 3, dtype=torch.float64, device=dev)
    if eng is not None and hasattr(eng, "sym_pack") and stats.acc2.is_cuda:
        eng.sym_pack(stats.acc2.to(torch.float64), out)
    else:
        i, j = _tri_indices(M, dev)
        out[:P * tri] = stats.acc2.to(torch.float64)[:, i, j].reshape(-1)
    out[P * tri:P * tri + P * M] = stats.acc1.to(torch.float64).reshape(-1)
    out[P * tri + P * M:] = torch.cat(tail)
    return out


def unpack_stats(packed: torch.Tensor, P: int, M: int, with_sites: bool, eng=None):
    o = 0
    acc2 = acc1 = None
    if with_sites:
        tri = M * (M + 1) // 2
        if eng is not None and hasattr(eng, "sym_unpack") and packed.is_cuda:
            acc2 = eng.sym_unpack(packed, P, M)
        else:
            i, j = _tri_indices(M, packed.device)
            v = packed[:P * tri].reshape(P, tri)
            acc2 = torch.empty((P, M, M), dtype=torch.float64, device=packed.device)
            acc2[:, i, j] = v
            acc2[:, j, i] = v
        o += P * tri
        acc1 = packed[o:o + P * M].reshape(P, M)
        o += P * M
    return acc2, acc1, packed[o], packed[o + 1], packed[o + 2]


def packed_size(P: int, M: int, with_sites: bool) -> int:
    return (P * (M * (M + 1) // 2) + P * M if with_sites else 0) + 3


def reduce_stats(stats, P: int, M: int, with_sites: bool, reduce: bool, eng=None, extra=None):
    """Sum of the per-shard statistics over the ranks: (acc2 [P, M, M], acc1 [P, M], sum ve, #non-positive var, rows, extra).
    One all-reduce of the packed buffer (plus ``extra``, a 1-D fp64 tensor that rides along) when ``reduce``; a single
    process passes its tensors through without packing."""
    if not reduce:
        rows = torch.full((), float(stats.n_rows), dtype=torch.float64, device=stats.ve_sum.device)
        return stats.acc2, stats.acc1, stats.ve_sum, stats.nonpos, rows, extra
    packed = pack_stats(stats, with_sites, eng)
    if extra is not None:
        packed = torch.cat([packed, extra.to(torch.float64).reshape(-1)])
    all_reduce_sum(packed)
    n = packed_size(P, M, with_sites)
    acc2, acc1, ve_sum, nonpos, rows = unpack_stats(packed[:n], P, M, with_sites, eng)
    return acc2, acc1, ve_sum, nonpos, rows, (packed[n:] if extra is not None else None)


def all_reduce_sum(packed: torch.Tensor) -> torch.Tensor:
    """In-place sum over ranks (no-op for a single process)."""
    if collectives_on():
        with _timed("all_reduce_sum", packed):
            dist.all_reduce(packed, op=dist.ReduceOp.SUM)
    return packed


def broadcast_from_rank0(t: torch.Tensor) -> torch.Tensor:
    """In-place broadcast of rank 0's values (no-op for a single process).  Used for host-side DECISIONS derived from
    replicated data (the projection route from cond(K_uu)): every rank must take the same branch even if a library
    routine were to round differently from one device to the next."""
    if collectives_on():
        dist.broadcast(t, src=0)
    return t


def all_reduce_max(t: torch.Tensor) -> torch.Tensor:
    """In-place maximum over ranks (status words of a latent-split step: every rank raises or retries together)."""
    if collectives_on():
        with _timed("all_reduce_max", t):
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t


def all_gather_flat(send: torch.Tensor) -> torch.Tensor:
    """[world * n] <- every rank's [n] (same n on every rank), in rank order; a single process gets its own back."""
    if not collectives_on():
        return send
    out = torch.empty(dist.get_world_size() * send.numel(), dtype=send.dtype, device=send.device)
    with _timed("all_gather", out):
        dist.all_gather_into_tensor(out, send.contiguous())
    return out


def reduce_scatter_sum(packed: torch.Tensor, world: int) -> torch.Tensor:
    """This rank's [n / world] slice of the sum over ranks of ``packed`` [n] (slices in rank order).  RCCL: one
    reduce_scatter; gloo (CPU tests, rehearsals) has none: an all-reduce and a slice, the same numbers."""
    if not collectives_on():
        return packed
    n = packed.numel() // world
    if dist.get_backend() == "nccl":
        out = torch.empty(n, dtype=packed.dtype, device=packed.device)
        with _timed("reduce_scatter_sum", packed):
            dist.reduce_scatter_tensor(out, packed.contiguous(), op=dist.ReduceOp.SUM)
        return out
    with _timed("reduce_scatter_sum(all_reduce)", packed):
        dist.all_reduce(packed, op=dist.ReduceOp.SUM)
    r = dist.get_rank()
    return packed[r * n:(r + 1) * n]
