"""
Multi-GPU layer of the hot path: one process per GPU, ``torch.distributed`` (backend ``"nccl"`` = RCCL over xGMI on
ROCm; ``"gloo"`` in the CPU tests).

What shards: the **batch dimension** - a set of fully independent filters (``filters/base.py:93-119``: every
reduction / scan / search runs over the particle axis only).  SMC^2 puts its theta-particles there
(``inference/sequential/base.py:31-34``), so theta-columns are block-sharded across ranks and each rank runs the
single-GPU fused loop on its ``(N, B / world)`` slice with its slice of the parameters.  The only exchange the path
needs is the **all-gather of the per-filter log-likelihood increments** (``B`` floats = a few KiB: latency-bound, not
link-bound) so that every rank can update the theta-weights / theta-ESS identically (``sequential/state.py:35-44``,
``smc2.py:59-62``).  A single filter (B = 1) does *not* shard - that would need a cross-GPU scan - replicas only.
"""
from typing import Optional, Tuple

import torch
import torch.distributed as dist


def world() -> Tuple[int, int]:
    """(rank, world_size); (0, 1) when no process group is initialised."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_bounds(total: int, rank: Optional[int] = None, world_size: Optional[int] = None) -> Tuple[int, int]:
    """Block sharding of ``total`` batch columns: ``[start, stop)`` of this rank; the first ``total % world`` ranks
    hold one extra column, so any ``total >= world`` works."""
    r, w = world()
    rank = r if rank is None else rank
    world_size = w if world_size is None else world_size
    base, extra = divmod(total, world_size)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def shard_columns(t: torch.Tensor, total: int, dim: int = -1) -> torch.Tensor:
    """This rank's slice of a tensor that carries the batch dimension ``total`` on ``dim`` (a per-theta parameter
    ``(B,)``, per-series observations ``(T, B)``, a uniform tape ``(T, B)`` ...)."""
    lo, hi = shard_bounds(total)
    return t.narrow(dim, lo, hi - lo)


def all_gather_columns(local: torch.Tensor, total: int) -> torch.Tensor:
    """All-gather along the last dim of per-column values (log-likelihoods ``(B_local,)`` or ``(T, B_local)``) into the
    full ``(..., total)`` tensor, in global column order, on every rank.  Uneven shards are padded to the largest one."""
    rank, w = world()
    if w == 1:
        return local
    sizes = [shard_bounds(total, r, w)[1] - shard_bounds(total, r, w)[0] for r in range(w)]
    width = max(sizes)
    lead = local.shape[:-1]
    buf = local.new_zeros(lead + (width,))
    buf[..., : local.shape[-1]] = local
    out = [local.new_empty(lead + (width,)) for _ in range(w)]
    dist.all_gather(out, buf.contiguous())
    return torch.cat([out[r][..., : sizes[r]] for r in range(w)], dim=-1)


def theta_ess(log_weights: torch.Tensor) -> torch.Tensor:
    """ESS of the (gathered) theta-weights - identical on every rank since every rank holds all ``B`` values."""
    w = torch.softmax(log_weights - log_weights.max(), dim=0)
    return 1.0 / (w * w).sum()
