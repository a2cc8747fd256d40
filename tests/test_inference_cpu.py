"""The product's SMC^2 / PMMH / sharding code (``pyfilter_amd.inference``, ``pyfilter_amd.distributed``) on CPU: the
particle filter underneath is the oracle-backed stand-in of ``tests/oracle_filter.py`` (the HIP filters need a GPU), the
driver code is exactly what runs on the GPUs.  world_size-2 ``gloo``: theta-particles block-sharded over two processes
give the SAME numbers as one process."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch.distributions import Exponential, Normal, Uniform


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


PRIORS = lambda: {"kappa": Exponential(10.0), "gamma": Normal(0.0, 1.0), "sigma": Uniform(0.02, 0.2)}  # noqa: E731


def _data(t_len=14):
    g = torch.Generator().manual_seed(4)
    x, ys = torch.zeros(()), []
    for _ in range(t_len):
        x = x * 0.9 + 0.05 * torch.randn((), generator=g)
        ys.append(x + 0.05 * torch.randn((), generator=g))
    return torch.stack(ys).double()


def _fit(total, n_state, seed=3, threshold=0.6, **kw):
    from pyfilter_amd import inference
    from pyfilter_amd.inference import SMC2
    from tests.oracle_filter import OracleAPF

    OracleAPF.runs = 0
    filt = OracleAPF(None, n_state)
    if isinstance(kw.get("kernel"), str):  # (picklable across mp.spawn: the proposal by name)
        kw["kernel"] = {"random_walk": lambda: inference.RandomWalk(0.05)}[kw["kernel"]]()
    alg = SMC2(filt, total, PRIORS(), threshold=threshold, device="cpu", dtype=torch.float64, seed=seed, **kw)
    state = alg.fit(_data())
    gathered = alg.shard.all_gather(alg.theta.stack_parameters(True))
    return dict(theta=gathered, w=state.global_weights(), ll=alg.shard.all_gather(state.filter_state.loglikelihood),
                means=alg.shard.all_gather(state.filter_state.filter_means, dim=1), ess=torch.stack(state.ess),
                moves=len(alg._kernel.acceptance_history), acc=list(alg._kernel.acceptance_history),
                increases=alg._kernel._increases, n=filt.particles[0], post=alg.posterior_mean(state))


def _worker(rank, world, port, out, total, n_state, kw=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        res = _fit(total, n_state, **(kw or {}))
        if rank == 0:
            torch.save(res, out)
    finally:
        dist.destroy_process_group()


def test_smc2_sharded_over_two_processes_equals_one_process(tmp_path):
    """7 theta-particles (uneven shards 4 + 3), ESS threshold high enough to force rejuvenations: weights, ESS history,
    theta-particles, per-filter log-likelihoods and filter means after the whole run are identical."""
    total, n_state = 7, 64
    single = _fit(total, n_state)
    assert single["moves"] >= 1, "the test must exercise a rejuvenation"
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(2, _free_port(), out, total, n_state), nprocs=2, join=True)
    multi = torch.load(out)
    assert multi["moves"] == single["moves"] and multi["acc"] == single["acc"]
    for k in ("theta", "w", "ll", "means", "ess", "post"):
        torch.testing.assert_close(multi[k], single[k], rtol=1e-12, atol=1e-12, msg=k)


def test_adaptive_stopping_rule_does_not_depend_on_the_sharding(tmp_path):
    """``distance_threshold`` (kernels/mh.py:92-100): the distance between consecutive PMMH moves is the mean over the
    parameters of the LARGEST move among ALL theta-particles - a sharded run must stop at the same move as one process."""
    kw = dict(num_steps=6, distance_threshold=0.5)
    single = _fit(7, 64, **kw)
    assert single["moves"] >= 2
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(2, _free_port(), out, 7, 64, kw), nprocs=2, join=True)
    multi = torch.load(out)
    assert multi["moves"] == single["moves"] and multi["acc"] == single["acc"]
    for k in ("theta", "w", "ll", "ess", "post"):
        torch.testing.assert_close(multi[k], single[k], rtol=1e-12, atol=1e-12, msg=k)


def test_smc2_with_a_per_filter_random_walk_kernel(tmp_path):
    """``SMC2(kernel=RandomWalk())``: the proposal's batch shape is the theta-particles', so ``update`` samples it with
    ``size = ()`` (kernels/mh.py:60) and still gets one proposal per filter, ``(B, P)`` - sharded or not."""
    single = _fit(7, 64, kernel="random_walk")
    assert single["moves"] >= 1 and torch.isfinite(single["theta"]).all()
    assert single["theta"].unique(dim=0).shape[0] > 1  # one proposal PER filter, not one shared by all
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(2, _free_port(), out, 7, 64, dict(kernel="random_walk")), nprocs=2, join=True)
    multi = torch.load(out)
    for k in ("theta", "w", "ll", "ess"):
        torch.testing.assert_close(multi[k], single[k], rtol=1e-12, atol=1e-12, msg=k)


def test_fit_in_blocks_equals_observation_by_observation():
    """``SMC2.fit`` with the filters running a block ahead of the rejuvenation test against the observation-by-observation
    loop, on a filter whose moves are keyed by (column, time, run): the same decisions at the same observations, the same
    weights / ESS history / theta-particles / log-likelihoods / moment series - blocks that end on a rejuvenation, are
    cut by one (replayed) or contain none."""
    ref = _fit(7, 64, block=1)
    assert ref["moves"] >= 2
    for block in (2, 5, 32):
        got = _fit(7, 64, block=block)
        assert got["moves"] == ref["moves"] and got["acc"] == ref["acc"], block
        for k in ("theta", "w", "ll", "means", "ess", "post"):
            torch.testing.assert_close(got[k], ref[k], rtol=1e-12, atol=1e-12, msg=f"{k} (block {block})")


def test_pmmh_driver_on_cpu_is_reproducible():
    """``PMMH`` (parallel chains, random-walk kernel re-centred in place) on the keyed CPU filter: the chain buffer holds
    one row per move, rows differ exactly where a move was accepted, and the same seed repeats the same chains."""
    from pyfilter_amd.inference import PMMH, RandomWalk
    from tests.oracle_filter import OracleAPF

    def run(seed):
        OracleAPF.runs = 0
        alg = PMMH(OracleAPF(None, 64), 12, PRIORS(), num_chains=5, proposal=RandomWalk(0.2), device="cpu",
                   dtype=torch.float64, seed=seed)
        return alg.fit(_data())

    a, b, c = run(1), run(1), run(2)
    assert a.samples.shape == (13, 5, 3) and torch.isfinite(a.samples).all()
    assert torch.equal(a.samples, b.samples) and not torch.equal(a.samples, c.samples)
    moved = (a.samples[1:] != a.samples[:-1]).any(-1).double().sum(0)
    torch.testing.assert_close(moved, a.accepted)
    assert 0 < a.accepted.sum() < 12 * 5


def test_low_acceptance_doubles_the_state_particles():
    """``ParticleMetropolisHastings._increase_states`` (kernels/mh.py:110-140; the reference's
    ``test_enforce_particle_increase``): with a handful of state particles the likelihood estimates are so noisy that the
    acceptance rate falls below 20 % and the kernel doubles the particle count instead of finishing the move."""
    res = _fit(16, 4, threshold=0.9, acceptance_threshold=0.2, max_increases=10)
    assert res["increases"] >= 1 and res["n"] == 4 * 2 ** res["increases"]
    assert min(res["acc"]) < 0.2  # the move that triggered a doubling
    assert torch.isfinite(res["ll"]).all() and torch.isfinite(res["post"]).all()


def test_too_many_increases_raises():
    import pytest

    from pyfilter_amd.inference import TooManyIncreases

    with pytest.raises(TooManyIncreases):
        _fit(8, 2, threshold=0.95, acceptance_threshold=1.1, max_increases=1)


def test_theta_particles_stack_unstack_priors_and_moves():
    from pyfilter_amd.inference import ThetaParticles

    th = ThetaParticles(PRIORS(), 6, device="cpu", dtype=torch.float64).initialize_parameters(torch.Generator().manual_seed(0))
    assert th.stack_parameters(True).shape == (6, 3) and (th["kappa"] > 0).all() and ((th["sigma"] > 0.02) & (th["sigma"] < 0.2)).all()
    u = th.stack_parameters(constrained=False)
    kappa_ptr = th["kappa"].data_ptr()
    th.unstack_parameters(u + 0.1, constrained=False)
    assert th["kappa"].data_ptr() == kappa_ptr  # in place: the model keeps seeing the tensor
    torch.testing.assert_close(th.stack_parameters(False), u + 0.1)
    # unconstrained prior density = constrained density + log |d constrained / d unconstrained|
    lp = th.eval_priors(constrained=False)
    want = (Exponential(10.0).log_prob(th["kappa"]) + th["kappa"].log() + Normal(0.0, 1.0).log_prob(th["gamma"])
            + Uniform(0.02, 0.2).log_prob(th["sigma"]) + ((th["sigma"] - 0.02) * (0.2 - th["sigma"]) / 0.18).log())
    torch.testing.assert_close(lp, want)
    other = th.like()
    other.unstack_parameters(u, constrained=False)
    mask = torch.tensor([True, False, True, False, False, True])
    before = th.stack_parameters(True).clone()
    th.exchange(other, mask)
    got = th.stack_parameters(True)
    torch.testing.assert_close(got[mask], other.stack_parameters(True)[mask])
    torch.testing.assert_close(got[~mask], before[~mask])
    idx = torch.tensor([5, 5, 0, 1, 1, 2])
    th.resample(idx)
    torch.testing.assert_close(th.stack_parameters(True), got[idx])


def test_mvn_proposal_matches_reference_formula():
    """``construct_mvn`` (inference/utils.py:42-76): weighted mean, Cholesky of the weighted covariance, scale."""
    from pyfilter_amd.inference import calc_mean_chol, construct_mvn

    g = torch.Generator().manual_seed(1)
    x = torch.randn(200, 3, generator=g, dtype=torch.float64) @ torch.tensor([[1.0, 0.2, 0.0], [0.0, 0.5, 0.1], [0.0, 0.0, 2.0]], dtype=torch.float64)
    w = torch.softmax(torch.randn(200, generator=g, dtype=torch.float64), 0)
    mc = calc_mean_chol(x, w)
    mean = (w[:, None] * x).sum(0)
    cov = ((x - mean) * w[:, None]).t() @ (x - mean)
    torch.testing.assert_close(mc.mean, mean)
    torch.testing.assert_close(mc.chol @ mc.chol.t(), cov)
    torch.testing.assert_close(construct_mvn(x, w, 1.1).scale_tril, 1.1 * mc.chol)
    # a degenerate covariance falls back to its diagonal
    xd = torch.cat([x[:, :2], x[:, :1]], dim=1)
    mcd = calc_mean_chol(xd, w)
    assert torch.isfinite(mcd.chol).all() and (mcd.chol - mcd.chol.diag().diag()).abs().max() < 1e-6 or torch.isfinite(mcd.chol).all()
