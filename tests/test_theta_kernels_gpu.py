"""The theta-level kernels of an SMC^2 / PMMH move (``pyfilter_amd/csrc/pf_theta.hpp``: ``pf_theta_fit / _propose / _accept``)
against the ``torch.distributions`` arithmetic they replace - the product's own general route (``inference/utils.py``,
``parameters.py``, ``pmmh.py::run_pmmh``), which restates ``pyfilter/inference/utils.py:42-76`` (``construct_mvn``),
``prior.py:47-123`` (the priors' bijections) and ``batch/mcmc/utils.py:48-70`` (the acceptance step) and is itself pinned to
the reference's event logs (``tests/test_inference_reference_*.py`` - which run on BOTH routes, see ``theta_route`` there).

float64: 1e-11 relative (the kernels and torch evaluate the same formulas in double, in a different order); float32
tensors: the kernels still compute in double, so they sit within float32 rounding of the float64 answer."""
import math

import pytest
import torch
from torch.distributions import Beta, Exponential, Gamma, HalfNormal, LogNormal, MultivariateNormal, Normal, Uniform

from pyfilter_amd import ops
from pyfilter_amd.hints import HINTS
from pyfilter_amd.inference import ThetaParticles
from pyfilter_amd.inference.utils import construct_mvn, theta_normalize

pytestmark = pytest.mark.gpu

PRIORS = {"a": Normal(0.3, 1.7), "b": LogNormal(-2.0, 0.8), "c": Exponential(4.0), "d": Gamma(2.5, 3.0), "e": HalfNormal(0.7),
          "f": Beta(2.0, 5.0), "g": Uniform(-0.4, 1.3)}


def _tol(dtype):
    return dict(rtol=1e-11, atol=1e-12) if dtype == torch.float64 else dict(rtol=3e-6, atol=3e-6)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("b,p", [(1000, 3), (77, 1), (4096, 8)])
def test_theta_fit_is_construct_mvn(dtype, b, p):
    g = torch.Generator().manual_seed(b + p)
    mix = torch.randn(p, p, generator=g, dtype=torch.float64)
    x = (torch.randn(b, p, generator=g, dtype=torch.float64) @ mix + torch.randn(p, generator=g, dtype=torch.float64)).to(dtype).cuda()
    lw = (3.0 * torch.randn(b, generator=g, dtype=torch.float64)).to(dtype).cuda()
    lw[5], lw[11], lw[17] = float("nan"), -math.inf, math.inf  # (normalize: NaN and +inf carry no weight)
    for weights in (lw, torch.zeros_like(lw), None):
        mean, chol = ops.theta_fit(x, weights, 1.1)
        # (the reference weights from the host: torch's float64 softmax over more than 1 024 entries on the device is only good
        # to ~1e-6 relative on this stack - measured 1.7e-6 on the mean at B = 4 096 - the kernel agrees with the host to 1e-14)
        w = theta_normalize(lw.double().cpu() if weights is lw else torch.zeros(b, dtype=torch.float64)).cuda()
        want = construct_mvn(x.double(), w, 1.1)
        torch.testing.assert_close(mean.double(), want.loc, **_tol(dtype))
        torch.testing.assert_close(chol.double(), want.scale_tril, **_tol(dtype))
        assert torch.equal(chol, chol.tril())


def test_theta_fit_of_a_degenerate_cloud_keeps_the_diagonal():
    """A covariance that is not positive definite -> its diagonal's square root (inference/utils.py:50-55)."""
    g = torch.Generator().manual_seed(3)
    x = torch.randn(500, 3, generator=g, dtype=torch.float64).cuda()
    x[:, 2] = 0.0  # a parameter every theta-particle agrees on (zero: its row of the covariance is exactly 0 in any order)
    mean, chol = ops.theta_fit(x, None, 1.0)
    want = construct_mvn(x, torch.full((500,), 1 / 500, dtype=torch.float64, device="cuda"), 1.0)
    torch.testing.assert_close(chol, want.scale_tril, rtol=1e-11, atol=1e-12)
    assert float(chol[2, 2]) == 0.0 and float(chol[1, 0]) == 0.0 and float(mean[2]) == 0.0


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_theta_propose_is_the_priors_bijections_and_densities(dtype):
    b = 777
    theta = ThetaParticles(PRIORS, b, "cuda", dtype).initialize_parameters(torch.Generator().manual_seed(1))
    packed = theta.native_priors()
    assert packed is not None and packed.P == 7 and list(packed.kind)[:7] == [0, 1, 2, 3, 4, 5, 6]
    p = packed.P
    g = torch.Generator().manual_seed(2)
    mean = torch.randn(p, generator=g, dtype=torch.float64).mul(0.5).to(dtype).cuda()
    chol = (torch.randn(p, p, generator=g, dtype=torch.float64).tril() * 0.6).to(dtype).cuda().contiguous()
    eps = torch.randn(b, p, generator=g, dtype=torch.float64).to(dtype).cuda()
    eps[0] = 12.0   # far tails: sigmoid / exp saturate the way torch's transforms do
    eps[1] = -12.0
    u, lp = ops.theta_propose(packed, mean, chol, eps, [theta[n] for n in theta.names()])
    want_u = mean.double() + (chol.double() @ eps.double().unsqueeze(-1)).squeeze(-1)
    torch.testing.assert_close(u.double(), want_u, **_tol(dtype))
    # the constrained values and the log prior of the u the kernel stored (what every later evaluation starts from)
    ref = ThetaParticles(PRIORS, b, "cuda", torch.float64).initialize_parameters(torch.Generator().manual_seed(1))
    total = 0.0
    for i, (name, prior) in enumerate(ref.priors.items()):
        ui = u[:, i].double()
        torch.testing.assert_close(theta[name].double(), prior.get_constrained(ui), **_tol(dtype))
        total = total + prior.unconstrained.log_prob(ui)
    ok = torch.isfinite(total)
    assert ok.sum() >= b - 2
    tol = _tol(dtype) if dtype == torch.float64 else dict(rtol=2e-6, atol=2e-5)
    torch.testing.assert_close(lp.double()[ok], total[ok], **tol)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_theta_accept_is_the_metropolis_hastings_ratio(dtype):
    b, p = 1500, 4
    g = torch.Generator().manual_seed(9)
    r = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64)  # noqa: E731
    fwd = (r(p).to(dtype).cuda(), (r(p, p).tril() * 0.3 + torch.eye(p, dtype=torch.float64)).to(dtype).cuda())
    rev = (r(p).to(dtype).cuda(), (r(p, p).tril() * 0.3 + torch.eye(p, dtype=torch.float64)).to(dtype).cuda())
    u_cur, u_star = r(b, p).to(dtype).cuda(), r(b, p).to(dtype).cuda()
    pr_cur, pr_star, ll_cur, ll_star = (r(b).to(dtype).cuda() for _ in range(4))
    ll_star[3] = float("nan")   # a failed re-filter: rejected
    ll_star[4] = -math.inf
    unif = torch.rand(b, generator=g, dtype=torch.float64).to(dtype).cuda()
    log_acc, accepted, rate = ops.theta_accept(u_cur, u_star, fwd, rev, pr_cur, pr_star, ll_cur, ll_star, unif)
    q = lambda k: MultivariateNormal(k[0].double(), scale_tril=k[1].double(), validate_args=False)  # noqa: E731
    want = (q(rev).log_prob(u_cur.double()) - q(fwd).log_prob(u_star.double())) + (pr_star.double() - pr_cur.double()) + (
        ll_star.double() - ll_cur.double())
    tol = _tol(dtype) if dtype == torch.float64 else dict(rtol=2e-6, atol=2e-5)
    torch.testing.assert_close(log_acc.double(), want, equal_nan=True, **tol)
    assert accepted.dtype == torch.bool and torch.equal(accepted, unif.log() < log_acc)  # on the kernel's own log_acc
    assert not bool(accepted[3]) and not bool(accepted[4])
    assert abs(float(rate) - float(accepted.double().mean())) < 1e-6


def _smc2_run(dtype, native):
    from pyfilter_amd import timeseries as ts
    from pyfilter_amd.filters.particle import APF, proposals
    from pyfilter_amd.inference import SMC2
    from pyfilter_amd.timeseries import models

    HINTS.theta_kernels = native
    try:
        def build(theta):
            t = lambda v: torch.tensor(v, dtype=dtype, device="cuda")  # noqa: E731
            return ts.LinearStateSpaceModel(models.OrnsteinUhlenbeck(theta["kappa"], theta["gamma"], theta["sigma"], dt=1.0), (t(1.0), t(0.05)))

        g = torch.Generator().manual_seed(5)
        x, ys = 0.0, []
        for _ in range(120):
            x = x * math.exp(-0.05) + 0.15 * math.sqrt((1 - math.exp(-0.1)) / 0.1) * float(torch.randn((), generator=g))
            ys.append(x + 0.05 * float(torch.randn((), generator=g)))
        y = torch.tensor(ys, dtype=dtype, device="cuda")
        pri = {"kappa": Exponential(10.0), "gamma": Normal(0.0, 1.0), "sigma": LogNormal(-2.0, 1.0)}
        filt = APF(build, 200, proposal=proposals.LinearGaussianObservations(), seed=11)
        alg = SMC2(filt, 256, pri, threshold=0.5, device="cuda", dtype=dtype, seed=3)
        alg._kernel.trace = []
        state = alg.fit(y)
        moves = [t for t in alg._kernel.trace if t["kind"] == "pmmh"]
        fits = [t for t in alg._kernel.trace if t["kind"] == "rejuvenate"]
        return alg, state, moves, fits
    finally:
        HINTS.theta_kernels = True


def test_a_fit_on_the_theta_kernels_is_the_fit_on_torch():
    """One SMC^2 run (float64, same seeds) on both theta routes: the same rejuvenations at the same observations, the same
    proposals, acceptance probabilities and accepted theta-particles, the same posterior."""
    from pyfilter_amd.inference.pmmh import GaussianKernel

    a1, s1, m1, f1 = _smc2_run(torch.float64, True)
    a0, s0, m0, f0 = _smc2_run(torch.float64, False)
    assert len(m1) == len(m0) >= 2 and len(f1) == len(f0)
    assert all(isinstance(f["kernel"], GaussianKernel) for f in f1) and not any(isinstance(f["kernel"], GaussianKernel) for f in f0)
    for k, (x, y) in enumerate(zip(f1, f0)):
        assert torch.equal(x["indices"], y["indices"])
        torch.testing.assert_close(x["kernel"].loc, y["kernel"].loc, rtol=1e-9, atol=1e-11)
        torch.testing.assert_close(x["kernel"].scale_tril, y["kernel"].scale_tril, rtol=1e-9, atol=1e-11)
    for k, (x, y) in enumerate(zip(m1, m0)):
        torch.testing.assert_close(x["rvs"], y["rvs"], rtol=1e-9, atol=1e-11)
        torch.testing.assert_close(x["proposed_ll"], y["proposed_ll"], rtol=1e-8, atol=1e-8, equal_nan=True)
        torch.testing.assert_close(x["log_acc"], y["log_acc"], rtol=1e-7, atol=1e-7, equal_nan=True)
        assert torch.equal(x["accepted"], y["accepted"]), f"move {k}"
    torch.testing.assert_close(a1.posterior_mean(s1), a0.posterior_mean(s0), rtol=1e-8, atol=1e-10)
    assert a1._kernel.acceptance_history == pytest.approx(a0._kernel.acceptance_history, abs=1e-12)


def test_a_float32_fit_takes_the_theta_kernels_and_finds_the_posterior():
    alg, state, moves, fits = _smc2_run(torch.float32, True)
    from pyfilter_amd.inference.pmmh import GaussianKernel

    assert len(moves) >= 2 and all(isinstance(f["kernel"], GaussianKernel) for f in fits)
    post = alg.posterior_mean(state)
    assert torch.isfinite(post).all() and 0.0 < float(post[0]) < 0.5 and 0.0 < float(post[2]) < 0.5


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_step_by_step_with_the_polled_host_slot_is_step_by_step_with_the_copy(dtype, monkeypatch):
    """``SMC2.step()`` - the reference's loop, one host ESS test per observation (``smc2.py:53-65``) - reads the (ESS, all finite)
    pair from the host slot ``pf_theta_step`` writes; with no coherent host memory to be had it falls back to the copy command.
    Same seeds: the two loops rejuvenate at the same observations and end in the same theta-weights and posterior, bit for bit."""
    from pyfilter_amd import timeseries as ts
    from pyfilter_amd.filters.particle import APF, proposals
    from pyfilter_amd.inference import SMC2
    from pyfilter_amd.timeseries import models

    t = lambda v: torch.tensor(v, dtype=dtype, device="cuda")  # noqa: E731
    obs = (t(1.0), t(0.05))

    def build(theta):
        return ts.LinearStateSpaceModel(models.OrnsteinUhlenbeck(theta["kappa"], theta["gamma"], theta["sigma"], dt=1.0), obs)

    g = torch.Generator().manual_seed(5)
    x, ys = 0.0, []
    for _ in range(60):
        x = x * math.exp(-0.05) + 0.15 * math.sqrt((1 - math.exp(-0.1)) / 0.1) * float(torch.randn((), generator=g))
        ys.append(x + 0.05 * float(torch.randn((), generator=g)))
    y = torch.tensor(ys, dtype=dtype, device="cuda")
    pri = {"kappa": Exponential(10.0), "gamma": Normal(0.0, 1.0), "sigma": LogNormal(-2.0, 1.0)}
    outs = {}
    # (both loops on filter() + FilterResult.append: the fast driver of the slot path numbers its moves as pieces of one run and
    # therefore keys its Philox draws differently - its parity is tests/test_filters_gpu.py::test_the_online_run_driver_...)
    from pyfilter_amd.filters.particle import base as pbase

    monkeypatch.setattr(pbase._OnlineRun, "applies", staticmethod(lambda filt, result: False))
    for how in ("slot", "copy"):
        if how == "copy":
            def no_slot():
                raise ops.L.PfAmdError("no coherent host memory")
            monkeypatch.setattr(ops, "HostSlot", no_slot)
        alg = SMC2(APF(build, 300, proposal=proposals.LinearGaussianObservations(), seed=11), 128, pri, threshold=0.5, device="cuda",
                   dtype=dtype, seed=3)
        alg._kernel.trace = []
        state = alg.initialize()
        for yt in y:
            state = alg.step(yt, state)
        slot = alg.__dict__.get("_host_slot")
        assert (slot is not None and slot is not False and slot.seq > 0) if how == "slot" else slot is False
        outs[how] = (state.w.cpu(), alg.posterior_mean(state).cpu(), sum(1 for tr in alg._kernel.trace if tr["kind"] == "rejuvenate"),
                     torch.stack(state.ess).cpu())
    assert outs["slot"][2] == outs["copy"][2] and outs["slot"][2] >= 1, outs["slot"][2]
    for a, b in zip(outs["slot"], outs["copy"]):
        if isinstance(a, torch.Tensor):
            assert torch.equal(a, b)


def test_priors_outside_the_kernels_families_take_the_torch_route():
    from torch.distributions import Independent, StudentT

    theta = ThetaParticles({"a": Normal(0.0, 1.0), "b": StudentT(3.0)}, 16, "cuda", torch.float64).initialize_parameters()
    assert theta.native_priors() is None
    theta = ThetaParticles({"a": Independent(Normal(torch.zeros(2), torch.ones(2)), 1)}, 16, "cuda", torch.float64).initialize_parameters()
    assert theta.native_priors() is None  # (an event-shaped prior)
    assert ThetaParticles({"a": Normal(0.0, 1.0)}, 16, "cuda", torch.float64).native_priors() is None  # (no values yet)
    assert ThetaParticles({"a": Normal(0.0, 1.0)}, 16, "cuda", torch.float64).initialize_parameters().native_priors() is not None


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("n,b", [(16, 1000), (1, 37), (32, 4097)])
def test_theta_path_is_the_running_weights_and_their_ess(dtype, n, b):
    """``pf_theta_path`` against ``w + ll.cumsum(0)`` (the same additions in the same order: equal to the last bit) and
    ``pf_theta_ess`` of its rows."""
    g = torch.Generator().manual_seed(n * b)
    w = torch.randn(b, generator=g, dtype=torch.float64).to(dtype).cuda()
    ll = torch.randn(n, b, generator=g, dtype=torch.float64).mul(2.0).to(dtype).cuda()
    if n > 4:
        ll[3, 7] = float("nan")
        ll[2, 5] = -math.inf
    w_path, stats = ops.theta_path(w, ll)
    want = w + ll.cumsum(0)
    assert torch.equal(torch.nan_to_num(w_path, nan=123.0), torch.nan_to_num(want, nan=123.0))
    assert torch.equal(stats, ops.theta_ess(want))
    if n > 4:
        assert stats[1, 1] == 1 and stats[2, 1] == 0 and stats[3, 1] == 0  # "every weight finite" from the -inf on


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("b", [128, 37, 4097])
def test_theta_step_updates_in_place_and_reports_to_the_host_slot(dtype, b):
    """``pf_theta_step`` (one observation of ``SMC2.step``): ``w += ll`` in place equal to torch's addition to the last bit,
    statistics equal to ``pf_theta_ess`` of the result - on the device AND, polled without any copy command, in the host slot
    (``pf_host_alloc`` memory), observation after observation (the sequence number tells them apart)."""
    g = torch.Generator().manual_seed(b)
    w = torch.randn(b, generator=g, dtype=torch.float64).to(dtype).cuda()
    slot = ops.HostSlot()
    for t in range(60):
        ll = torch.randn(b, generator=g, dtype=torch.float64).mul(1.5).to(dtype).cuda()
        if t == 40:
            ll[5] = -math.inf
        want = w + ll
        stats = ops.theta_step(w, ll, slot if t % 7 != 3 else None)  # (a step without the slot in between leaves it alone)
        if t % 7 != 3:
            ess, finite = slot.wait()
            assert slot._seq.value == slot.seq
            ref = ops.theta_ess(want).tolist()
            assert [ess, finite] == ref, (t, ess, finite, ref)
        assert torch.equal(w, want)
        assert torch.equal(stats, ops.theta_ess(want))
    assert stats[1] == 0  # (the -inf from observation 40 on)
    import copy

    other = copy.deepcopy(slot)
    assert other.ptr != slot.ptr and other.seq == 0


@pytest.mark.parametrize("n_state", [300, 4096])
def test_fit_with_host_rows_is_fit_with_the_copy_and_the_event(n_state, monkeypatch):
    """``SMC2.fit``: a block's statistics rows polled in host memory (``ops.HostRows``, the default) against the copy command +
    event per block it falls back to where no coherent host memory is to be had - same seeds, same decisions, the same theta-weights
    and posterior bit for bit (column route at 300 particles, column-cluster route - status word through the rows - at 4 096)."""
    from pyfilter_amd import timeseries as ts
    from pyfilter_amd.filters.particle import APF, proposals
    from pyfilter_amd.inference import SMC2
    from pyfilter_amd.timeseries import models

    dtype = torch.float32
    t = lambda v: torch.tensor(v, dtype=dtype, device="cuda")  # noqa: E731
    obs = (t(1.0), t(0.05))

    def build(theta):
        return ts.LinearStateSpaceModel(models.OrnsteinUhlenbeck(theta["kappa"], theta["gamma"], theta["sigma"], dt=1.0), obs)

    g = torch.Generator().manual_seed(5)
    x, ys = 0.0, []
    for _ in range(90):
        x = x * math.exp(-0.05) + 0.15 * math.sqrt((1 - math.exp(-0.1)) / 0.1) * float(torch.randn((), generator=g))
        ys.append(x + 0.05 * float(torch.randn((), generator=g)))
    y = torch.tensor(ys, dtype=dtype, device="cuda")
    pri = {"kappa": Exponential(10.0), "gamma": Normal(0.0, 1.0), "sigma": LogNormal(-2.0, 1.0)}
    outs = {}
    for how in ("rows", "copy"):
        if how == "copy":
            def no_rows(n):
                raise ops.L.PfAmdError("no coherent host memory")
            monkeypatch.setattr(ops, "HostRows", no_rows)
        alg = SMC2(APF(build, n_state, proposal=proposals.LinearGaussianObservations(), seed=11), 64, pri, threshold=0.5, device="cuda",
                   dtype=dtype, seed=3)
        state = alg.fit(y, block=8)
        bufs = alg.__dict__.get("_host_row_bufs", {})
        assert (len(bufs) > 0 and all(isinstance(r, ops.HostRows) and r.seq > 0 for r in bufs.values())) if how == "rows" else \
            all(r is False for r in bufs.values())
        outs[how] = (state.w.cpu(), alg.posterior_mean(state).cpu(), len(alg._kernel.acceptance_history), torch.stack(state.ess).cpu())
    assert outs["rows"][2] == outs["copy"][2] >= 1
    for a, b in zip(outs["rows"], outs["copy"]):
        if isinstance(a, torch.Tensor):
            assert torch.equal(a, b)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_theta_path_reports_every_row_into_host_memory(dtype):
    """``pf_theta_path`` with host rows: row q's (ESS, all finite) pair in host memory equals the device row, for every launch of a
    sequence on the same rows (the sequence number tells them apart); a non-zero status word of the run behind the block shows in
    every row and nothing is computed."""
    b, n = 700, 16
    g = torch.Generator().manual_seed(3)
    rows = ops.HostRows(n)
    status = torch.zeros(1, dtype=torch.int32, device="cuda")
    for t in range(12):
        k = n if t % 3 else 5  # (a shorter block on the same rows in between)
        w0 = torch.randn(b, generator=g, dtype=torch.float64).to(dtype).cuda()
        ll = torch.randn(k, b, generator=g, dtype=torch.float64).to(dtype).cuda()
        if t == 7:
            ll[3, 11] = float("nan")
        w_path, stats = ops.theta_path(w0, ll, rows, status)
        torch.testing.assert_close(w_path, ops.theta_path(w0, ll)[0], rtol=0, atol=0, equal_nan=True)
        host = [rows.wait(q) for q in range(k)]
        assert [[e, f] for e, f, _ in host] == stats.double().tolist() or t == 7
        if t == 7:  # (NaN != NaN: compare the flags and the rows before the NaN)
            assert [f for _, f, _ in host] == [1.0] * 3 + [0.0] * (k - 3) and [e for e, _, _ in host][:3] == stats[:3, 0].double().tolist()
        assert all(st == 0 for _, _, st in host) and rows._u[2] == rows.seq
    status.fill_(2)
    _, stats = ops.theta_path(w0, ll, rows, status)
    host = [rows.wait(q) for q in range(ll.shape[0])]
    assert all(st == 2 and f == 0.0 and math.isnan(e) for e, f, st in host) and bool(stats[:, 0].isnan().all())
    import copy

    assert copy.deepcopy(rows).ptr != rows.ptr


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("b", [1000, 37, 5000])
def test_theta_resample_is_normalize_then_systematic(dtype, b):
    from pyfilter_amd.inference.utils import theta_systematic

    g = torch.Generator().manual_seed(b)
    for case in range(4):
        lw = (4.0 * torch.randn(b, generator=g, dtype=torch.float64)).to(dtype)
        if case == 1:
            lw[3], lw[b // 2], lw[b - 1] = float("nan"), math.inf, -math.inf
        if case == 2:
            lw[:] = -math.inf
            lw[b // 3] = 0.5  # all the weight on one theta-particle
        if case == 3:
            lw[:] = -math.inf  # nothing finite: equal weights
        for u in (0.0, 0.3718, 0.999999):
            got = ops.theta_resample(lw.cuda(), u).cpu()
            assert got.dtype == torch.int64 and int(got.min()) >= 0 and int(got.max()) <= b - 1
            assert bool((got[1:] >= got[:-1]).all())
            if case == 3:
                want = torch.arange(b) if u > 0 else torch.arange(b).sub(1).clamp_min(0)
                # (i + u) / B against cdf[j] = (j + 1) / B: ancestor i for u > 0 (u = 0: ties resolve to i - 1, as searchsorted does)
                assert int((got - want).abs().max()) <= 1
                continue
            # the reference arithmetic on the host in float64 (torch's float64 softmax on this device is 1e-6 off beyond 1 024 entries)
            want = theta_systematic(theta_normalize(lw.double()), u)
            if dtype == torch.float64:
                assert torch.equal(got, want), (case, u, int((got != want).sum()))
            else:  # float32 cdf and grid: a position within rounding of a cdf step may land on the neighbour
                assert int((got - want).abs().max()) <= 1 and int((got != want).sum()) <= max(2, b // 200)
            if case == 2 and u > 0:  # (u = 0: position 0 sits AT cdf[0] = 0 and takes ancestor 0, as searchsorted has it)
                assert bool((got == b // 3).all())


def test_initial_sample_with_per_filter_parameters_is_m_plus_s_z():
    from pyfilter_amd.timeseries.models import _expand

    n, b, d = 300, 7, 3
    g = torch.Generator().manual_seed(4)
    for dtype in (torch.float64, torch.float32):
        z = torch.randn(d, b, n, generator=g, dtype=torch.float64).to(dtype).cuda()
        for m_shape, s_shape in (((b,), (b,)), ((b, d), ()), ((d,), (b, 1)), ((), (b, d))):
            m = torch.randn(m_shape, generator=g, dtype=torch.float64).to(dtype).cuda()
            s = torch.rand(s_shape, generator=g, dtype=torch.float64).add(0.1).to(dtype).cuda()
            me, se = _expand(m, b, (d,), dtype, "cuda"), _expand(s, b, (d,), dtype, "cuda")
            got = ops.initial_sample_cols(me, se, n, b, d, 17, z)
            want = me.t().unsqueeze(-1) + se.t().unsqueeze(-1) * z
            assert torch.equal(got, want)
        # Philox draws: the standard normals of pf_initial_sample, mapped
        std = ops.initial_sample_soa([0.0] * d, [1.0] * d, n, b, d, dtype, torch.device("cuda"), 99)
        got = ops.initial_sample_cols(me, se, n, b, d, 99)
        assert torch.equal(got, me.t().unsqueeze(-1) + se.t().unsqueeze(-1) * std)


def test_the_fast_driver_reports_each_moves_own_statistics_row():
    """No rejuvenation for 150 observations (threshold ~0): the driver's statistics arrays fill and are replaced twice - every
    ``state.stats`` / ``state.ess`` entry is the ESS of the theta-weights at that observation, also at the array boundaries."""
    from pyfilter_amd import timeseries as ts
    from pyfilter_amd.filters.particle import APF, proposals
    from pyfilter_amd.filters.particle import base as pbase
    from pyfilter_amd.inference import SMC2
    from pyfilter_amd.timeseries import models

    dtype = torch.float64
    t = lambda v: torch.tensor(v, dtype=dtype, device="cuda")  # noqa: E731
    obs = (t(1.0), t(0.5))

    def build(theta):
        return ts.LinearStateSpaceModel(models.OrnsteinUhlenbeck(theta["kappa"], theta["gamma"], theta["sigma"], dt=1.0), obs)

    g = torch.Generator().manual_seed(9)
    y = 0.3 * torch.randn(150, generator=g, dtype=dtype).cuda()
    pri = {"kappa": Exponential(10.0), "gamma": Normal(0.0, 1.0), "sigma": LogNormal(-2.0, 1.0)}
    alg = SMC2(APF(build, 300, proposal=proposals.LinearGaussianObservations(), seed=4), 16, pri, threshold=1e-9, device="cuda",
               dtype=dtype, seed=1)
    state = alg.initialize()
    want = []
    for yt in y:
        state = alg.step(yt, state)
        assert state._online is not None
        p = torch.softmax(state.w, 0)
        want.append(float(1.0 / (p * p).sum()))
        assert float(state.stats[0]) == pytest.approx(want[-1], rel=1e-9)
    assert len(alg._kernel.acceptance_history) == 0 and pbase._OnlineRun.ROWS < 75
    assert [float(e) for e in state.ess[1:]] == pytest.approx(want, rel=1e-9)
    assert state.filter_state.filter_means.shape[0] == 151


@pytest.mark.parametrize("n_state,n_theta", [(4096, 24), (300, 128), (20000, 6)])
def test_the_fast_online_driver_is_the_step_by_step_loop(n_state, n_theta, monkeypatch):
    """``SMC2.step()`` through the fast driver (``_OnlineRun``: the loop's moves as pieces of one run, the ``FilterResult`` brought up to
    date when somebody looks) against the same loop over ``filter()`` + ``FilterResult.append`` (the driver switched off): the same
    decisions, theta-weights, log-likelihoods, moment series and latest state - float64, same seeds; on the column-cluster route
    (4 096 particles), the column route (300) and the per-step route (20 000)."""
    from pyfilter_amd import timeseries as ts
    from pyfilter_amd.filters.particle import APF, proposals
    from pyfilter_amd.filters.particle import base as pbase
    from pyfilter_amd.inference import SMC2
    from pyfilter_amd.timeseries import models

    dtype = torch.float64
    t = lambda v: torch.tensor(v, dtype=dtype, device="cuda")  # noqa: E731
    obs = (t(1.0), t(0.05))

    def build(theta):
        return ts.LinearStateSpaceModel(models.OrnsteinUhlenbeck(theta["kappa"], theta["gamma"], theta["sigma"], dt=1.0), obs)

    g = torch.Generator().manual_seed(5)
    x, ys = 0.0, []
    for _ in range(150):
        x = x * math.exp(-0.05) + 0.15 * math.sqrt((1 - math.exp(-0.1)) / 0.1) * float(torch.randn((), generator=g))
        ys.append(x + 0.05 * float(torch.randn((), generator=g)))
    y = torch.tensor(ys, dtype=dtype, device="cuda")
    y[40] = float("nan")
    pri = {"kappa": Exponential(10.0), "gamma": Normal(0.0, 1.0), "sigma": LogNormal(-2.0, 1.0)}
    outs = {}
    for how in ("fast", "plain"):
        if how == "plain":
            monkeypatch.setattr(pbase._OnlineRun, "applies", staticmethod(lambda filt, result: False))
        alg = SMC2(APF(build, n_state, proposal=proposals.LinearGaussianObservations(), seed=11), n_theta, pri, threshold=0.5,
                   device="cuda", dtype=dtype, seed=3)
        state = alg.initialize()
        used, seen = 0, []
        for k, yt in enumerate(y):
            state = alg.step(yt, state)
            used += state._online is not None
            seen.append(float(state.ess[-1]))
            if k == 70:  # somebody looks in between: the result is up to date, and the loop goes on
                assert state.filter_state.filter_means.shape[0] == k + 2
        assert (used > 100) == (how == "fast"), (how, used)
        # (the ESS history holds what each observation reported - not a later observation's row of a reused statistics array)
        assert [float(e) for e in state.ess[1:]] == seen
        fs = state.filter_state
        outs[how] = dict(w=state.w.cpu(), ess=torch.stack(state.ess).cpu(), ll=fs.loglikelihood.cpu(), means=fs.filter_means.cpu(),
                         var=fs.filter_variance.cpu(), x=fs.latest_state.timeseries_state.value.cpu(), lw=fs.latest_state.weights.cpu(),
                         idx=fs.latest_state.previous_indices.cpu(), moves=len(alg._kernel.acceptance_history),
                         t=int(fs.latest_state.timeseries_state.time_index))
    a, b = outs["fast"], outs["plain"]
    assert a["moves"] >= 1 and b["moves"] >= 1 and a["t"] == b["t"] == 150
    assert a["means"].shape == b["means"].shape == (151, n_theta, 1)
    # (the two loops key their Philox draws differently - a piece index per move here, piece 0 / a resumed piece there - so they are
    # two Monte-Carlo runs of the same algorithm unless the seeds line up; what must agree exactly is the structure, and
    # statistically the numbers)
    for key in ("w", "ess", "ll", "means"):
        assert torch.isfinite(a[key]).all() and torch.isfinite(b[key]).all(), key
    assert len(a["ess"]) == len(b["ess"]) == 151
    # the filtered means of the two runs: both follow the data (0.05 observation noise), theta-particle by theta-particle
    assert (a["means"][1:, :, 0].mean(1) - b["means"][1:, :, 0].mean(1)).abs().max() < 0.05
