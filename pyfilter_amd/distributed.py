"""
Multi-GPU layer of the hot path: one process per GPU, ``torch.distributed`` (backend ``"nccl"`` = RCCL over xGMI on
ROCm; ``"gloo"`` in the CPU tests).

What shards: the **batch dimension** - a set of fully independent filters (``filters/base.py:93-119``: every
reduction / scan / search runs over the particle axis only).  SMC^2 puts its theta-particles there
(``inference/sequential/base.py:31-34``), so theta-columns are block-sharded across ranks and each rank runs the
single-GPU fused loop on its ``(N, B / world)`` slice with its slice of the parameters.  A single filter (B = 1) does
*not* shard - that would need a cross-GPU scan - replicas only.

Exchanges (``Shard``), all along the filter (batch) dimension:

* every observation: **all-gather of the per-filter log-likelihood increments** - ``B`` floats, a few KiB,
  latency-bound - so that every rank updates the theta-weights and takes the ESS decision identically
  (``sequential/state.py:35-44``, ``smc2.py:59-62``);
* on a rejuvenation (``kernels/mh.py:52-108``): all-gather of the stacked theta ``(B, P)`` (the MVN proposal is built
  from all of them, identically everywhere), the identical systematic theta-resample on every rank, and the
  **redistribution of the surviving filters' states** to the ranks that own their new positions
  (``Shard.route`` once per resampling, ``Route.take`` per buffer, driven by ``inference/smc2.py:_take_filters``: one
  ``all_to_all_single`` of the DISTINCT columns that change owner - <= N (4 D + 12) bytes per moved column - and a local
  gather; nothing a rank already holds travels).
"""
from typing import List, Optional, Tuple

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
    return Shard(total).all_gather(local, dim=-1)


def theta_ess(log_weights: torch.Tensor) -> torch.Tensor:
    """ESS of the (gathered) theta-weights - identical on every rank since every rank holds all ``B`` values."""
    w = torch.softmax(log_weights - log_weights.max(), dim=0)
    return 1.0 / (w * w).sum()


SOLO = "solo"  # Shard(total, SOLO): this process alone holds every filter, whatever process group exists

# Testing switch: with a process group of ONE rank every exchange below is the identity and ``Shard`` / ``Route`` skip the
# collective.  ``force_collectives(True)`` makes a one-rank group issue them anyway - ``all_gather_into_tensor`` /
# ``all_gather`` / ``all_reduce`` / ``all_to_all_single`` with this rank as the only peer - so that a one-GPU box runs the
# very RCCL calls of an N-GPU job (tests/test_distributed_gpu.py::test_rccl_world_of_one_*).  Never set by the library.
_FORCE_COLLECTIVES = False


def force_collectives(on: bool = True) -> None:
    global _FORCE_COLLECTIVES
    _FORCE_COLLECTIVES = bool(on)


class Shard:
    """This rank's block of ``total`` filters and the collectives over the filter dimension.  With one process (no
    process group) every method is the identity / a local operation, so single-GPU code runs through the same calls."""

    def __init__(self, total: int, group=None):
        self.total = int(total)
        self.group = group
        if group is SOLO:
            self.group, self.rank, self.world = None, 0, 1
        elif group is not None and dist.is_available() and dist.is_initialized():
            self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)  # a sub-group: ITS ranks, not the world's
            if self.rank < 0:  # (torch's answer for a non-member: spans[-1] would silently hand out the LAST rank's block)
                raise ValueError("Shard: this process is not a member of the process group it was given")
        else:
            self.rank, self.world = world()
        self.spans: List[Tuple[int, int]] = [shard_bounds(self.total, r, self.world) for r in range(self.world)]
        self.lo, self.hi = self.spans[self.rank]
        # does an exchange go through torch.distributed?  (more than one rank - or a forced one-rank group, see above)
        self.collective = self.world > 1 or (_FORCE_COLLECTIVES and group is not SOLO and dist.is_available() and dist.is_initialized())

    @property
    def local(self) -> int:
        return self.hi - self.lo

    def slice(self, t: torch.Tensor, dim: int = 0) -> torch.Tensor:
        """This rank's block of a globally indexed tensor."""
        return t.narrow(dim, self.lo, self.local)

    def all_gather(self, local: torch.Tensor, dim: int = 0) -> torch.Tensor:
        """Concatenation of every rank's block along ``dim``, in global order, identical on every rank."""
        if not self.collective:
            return local
        dim = dim % local.dim()
        moved = local.movedim(dim, 0).contiguous()
        sizes = [hi - lo for lo, hi in self.spans]
        width = max(sizes)
        if min(sizes) == width and dist.get_backend(self.group) == "nccl":
            # even blocks on RCCL: one collective straight into the concatenated tensor (no padding, no second copy - this
            # is the path that moves whole filter states, ~100 MB per rejuvenation at 1024 x 8192)
            out = moved.new_empty((self.world * width,) + tuple(moved.shape[1:]))
            dist.all_gather_into_tensor(out, moved, group=self.group)
            return out.movedim(0, dim)
        buf = moved.new_zeros((width,) + tuple(moved.shape[1:]))
        buf[: moved.shape[0]] = moved
        out = [torch.empty_like(buf) for _ in range(self.world)]
        dist.all_gather(out, buf, group=self.group)
        return torch.cat([out[r][: sizes[r]] for r in range(self.world)], dim=0).movedim(0, dim)

    def all_mean(self, local_sum: torch.Tensor, local_count: int) -> torch.Tensor:
        """Mean over all filters of a quantity summed locally (e.g. an acceptance rate)."""
        if not self.collective:
            return local_sum / max(1, local_count)
        v = torch.stack([local_sum.to(torch.float64).reshape(()), torch.tensor(float(local_count), device=local_sum.device, dtype=torch.float64)])
        dist.all_reduce(v, group=self.group)
        return (v[0] / v[1]).to(local_sum.dtype)

    def all_max(self, local: torch.Tensor) -> torch.Tensor:
        """Element-wise maximum over the ranks."""
        if not self.collective:
            return local
        out = local.clone()
        dist.all_reduce(out, op=dist.ReduceOp.MAX, group=self.group)
        return out

    def route(self, global_index: torch.Tensor, full_index: Optional[torch.Tensor] = None) -> "Route":
        """The exchange plan of a global gather along the filter dimension (see ``Route``); ``global_index`` = the global
        ancestors of THIS rank's positions.  One small all-gather of the index vectors + a host copy: build it once per
        resampling and move every buffer through it.  ``full_index``: the ancestors of ALL positions when the caller holds
        them anyway (SMC^2's rejuvenation resamples the theta-particles identically on every rank) - no all-gather then."""
        return Route(self, global_index, full_index)

    def take(self, local: torch.Tensor, global_index: torch.Tensor, dim: int = 0) -> torch.Tensor:
        """``concat(all blocks)[global_index]`` along ``dim`` - this rank's new block after a global gather (resampling of
        theta-particles: ``global_index`` = the ancestors of this rank's positions)."""
        return self.route(global_index).take(local, dim)


class Route:
    """Who sends which columns to whom when the filters are gathered by a global index vector (the resampling of the
    theta-particles in an SMC^2 rejuvenation, ``kernels/mh.py:53-57``).  Every rank learns every rank's wanted ancestors
    (all-gather of B integers), from which both sides of every pair derive the same list: the DISTINCT columns rank r wants
    from rank q, ascending.  ``take`` then moves a buffer with ONE ``all_to_all_single`` (RCCL over xGMI: point-to-point
    sends, only the columns that actually change owner - a column wanted several times travels once, a rank's own columns
    never leave it) and a local gather that puts the received columns in place.  SURVEY.md section 8(e): <= N (4 D + 12)
    bytes per MOVED column, against world x that for an all-gather of everything."""

    def __init__(self, shard: "Shard", global_index: torch.Tensor, full_index: Optional[torch.Tensor] = None):
        self.shard = shard
        self.device = global_index.device
        mine = global_index.to(torch.int64).reshape(-1)
        if not shard.collective:
            self.local_index = mine
            return
        if full_index is not None and full_index.numel() == shard.total:
            full = full_index.to(torch.int64).reshape(-1).cpu()  # every rank already holds every rank's wants
        else:
            full = shard.all_gather(mine).cpu()  # (total,) in global position order: rank r's wants = full[lo_r:hi_r]
        me = shard.rank
        lo, hi = shard.spans[me]
        wants = full[lo:hi]
        send_cols, self.send_splits = [], []
        for r, (rlo, rhi) in enumerate(shard.spans):  # what rank r wants from my block
            if r == me:
                self.send_splits.append(0)
                continue
            need = full[rlo:rhi]
            u = torch.unique(need[(need >= lo) & (need < hi)]) - lo
            send_cols.append(u)
            self.send_splits.append(int(u.numel()))
        self.send_index = (torch.cat(send_cols) if send_cols else torch.empty(0, dtype=torch.int64)).to(self.device)
        # where each of my wanted columns ends up: my own block first (no transfer), then the received runs by owner
        pos = torch.empty_like(wants)
        own = (wants >= lo) & (wants < hi)
        pos[own] = wants[own] - lo
        offset = hi - lo
        self.recv_splits = []
        for q, (qlo, qhi) in enumerate(shard.spans):
            if q == me:
                self.recv_splits.append(0)
                continue
            sel = (wants >= qlo) & (wants < qhi)
            u = torch.unique(wants[sel])
            pos[sel] = offset + torch.searchsorted(u, wants[sel])
            self.recv_splits.append(int(u.numel()))
            offset += int(u.numel())
        self.local_index = pos.to(self.device)
        self.moved = sum(self.recv_splits)  # columns this rank receives over the fabric

    def take(self, local: torch.Tensor, dim: int = 0) -> torch.Tensor:
        dim = dim % local.dim()
        if not self.shard.collective:
            return local.index_select(dim, self.local_index.to(local.device))
        moved = local.movedim(dim, 0)
        rest = tuple(moved.shape[1:])
        send = moved.index_select(0, self.send_index.to(local.device)).contiguous()
        recv = moved.new_empty((sum(self.recv_splits),) + rest)
        dist.all_to_all_single(recv, send, self.recv_splits, self.send_splits, group=self.shard.group)
        pool = torch.cat([moved, recv], dim=0) if recv.shape[0] else moved
        return pool.index_select(0, self.local_index.to(local.device)).movedim(0, dim)
