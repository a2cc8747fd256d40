"""world_size-2 ``gloo`` test of the multi-GPU layer (runs on CPU): theta-columns block-sharded across ranks, each rank
filters its shard, the per-filter log-likelihoods are all-gathered - and the result equals the unsharded run.

The per-shard filter here is the oracle (CPU); on the GPU box the same sharding wraps the fused HIP loop (bench.py
``--workload smc2_shard``)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import cpu_ref
from oracle.cases import CASE_BY_NAME, build_spec
from tests.helpers import load_golden


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import models as M
        from pyfilter_amd import distributed as D

        case = CASE_BY_NAME["ou_apf_boot_theta"]  # B = 3 theta-columns -> uneven shards (2 + 1)
        g = load_golden(case["name"], "f64")
        spec = build_spec(case, torch.float64)
        b = case["B"]
        lo, hi = D.shard_bounds(b)
        # shard the per-theta parameters, the particles and the draw tapes along the batch dim
        hp = tuple(D.shard_columns(p, b) if isinstance(p, torch.Tensor) and p.dim() == 1 and p.shape[0] == b else p
                   for p in spec.hidden_params)
        shard = M.ModelSpec(spec.hidden, hp, spec.dim, spec.dt, spec.init, spec.obs, spec.obs_params, spec.obs_dim)
        res = cpu_ref.batch_filter(
            shard, case["filter"], case["proposal"], g["y"], g["x0"][:, lo:hi], g["z_tape"].double()[:, :, lo:hi],
            g["u_tape"].double()[:, lo:hi], ess_threshold=case["ess_threshold"],
        )
        ll_all = D.all_gather_columns(res["loglikelihood"], b)
        means_all = D.all_gather_columns(res["filter_means"][..., 0], b)
        ess = D.theta_ess(ll_all)
        if rank == 0:
            torch.save({"ll": ll_all, "means": means_all, "ess": ess, "bounds": (lo, hi)}, out)
        # every rank must hold the same gathered values
        chk = ll_all.clone()
        dist.broadcast(chk, src=0)
        assert torch.equal(chk, ll_all)
    finally:
        dist.destroy_process_group()


def test_theta_sharding_world2_matches_unsharded(tmp_path):
    out = str(tmp_path / "r0.pt")
    port = _free_port()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out)
    g = load_golden("ou_apf_boot_theta", "f64")
    torch.testing.assert_close(got["ll"], g["loglikelihood"], rtol=1e-12, atol=1e-12)
    torch.testing.assert_close(got["means"], g["filter_means"][..., 0], rtol=1e-12, atol=1e-12)
    w = torch.softmax(g["loglikelihood"], 0)
    torch.testing.assert_close(got["ess"], 1.0 / (w * w).sum())
    assert got["bounds"] == (0, 2)


def test_shard_bounds_cover_everything():
    from pyfilter_amd.distributed import shard_bounds

    for total in (1, 2, 7, 8, 1024, 1025):
        for world in (1, 2, 3, 8):
            if total < world:
                continue
            spans = [shard_bounds(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _route_worker(rank, world, port, total):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pyfilter_amd.distributed import Shard

        sh = Shard(total)
        g = torch.Generator().manual_seed(5)  # the same global data / index vectors on every rank
        full2 = torch.randn(total, 6, generator=g, dtype=torch.float64)       # (B, ...): filters on dim 0
        full3 = torch.randn(3, total, 5, generator=g, dtype=torch.float64)    # (D, B, N): filters on dim 1
        fulli = torch.arange(total, dtype=torch.int32) * 7
        cases = [torch.randint(0, total, (total,), generator=g) for _ in range(4)]
        cases += [torch.arange(total), torch.zeros(total, dtype=torch.int64), torch.full((total,), total - 1)]
        for idx in cases:
            mine = sh.slice(idx)
            route = sh.route(mine)
            torch.testing.assert_close(route.take(sh.slice(full2)), full2[mine], rtol=0, atol=0)
            torch.testing.assert_close(route.take(sh.slice(full3, dim=1), dim=1), full3[:, mine], rtol=0, atol=0)
            assert torch.equal(route.take(sh.slice(fulli)), fulli[mine])
            torch.testing.assert_close(sh.take(sh.slice(full2), mine), full2[mine], rtol=0, atol=0)
            # only DISTINCT columns owned by somebody else cross the fabric
            foreign = mine[(mine < sh.lo) | (mine >= sh.hi)]
            assert route.moved == foreign.unique().numel()
        # a sub-group is its own world: rank / world / spans follow the group, not the default process group
        groups = [dist.new_group([r]) for r in range(world)]
        solo = Shard(total, groups[rank])
        assert (solo.rank, solo.world, solo.lo, solo.hi) == (0, 1, 0, total)
        torch.testing.assert_close(solo.all_max(full2[rank]), full2[rank])
        both = Shard(total, dist.new_group(list(range(world))))
        assert (both.rank, both.world) == (rank, world)
        m = both.all_max(torch.tensor([float(rank), -float(rank)], dtype=torch.float64))
        assert m.tolist() == [float(world - 1), 0.0]
    finally:
        dist.destroy_process_group()


def test_route_moves_only_the_columns_that_change_owner():
    """``Shard.route`` / ``Route.take`` (one ``all_to_all_single`` of the distinct moved columns + a local gather) against
    plain indexing of the concatenated blocks: uneven shards, duplicates, all-from-one-rank, identity; 2 and 3 ranks."""
    for world, total in ((2, 7), (3, 11)):
        mp.spawn(_route_worker, args=(world, _free_port(), total), nprocs=world, join=True)
