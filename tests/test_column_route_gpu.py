"""The column-persistent route (``pyfilter_amd/csrc/pf_column.hpp``: one workgroup per filter runs the whole time loop in
ONE launch) against the per-step route (``k_fused_step``, one launch per time step) and the oracle.

Both routes key their Philox draws by (seed, stream, step, filter x N + particle): a run with the same seed consumes
the SAME random numbers on either route, so in float64 the two must agree to rounding - identical ancestors, moments and
log-likelihoods to 1e-9 - for every filter / proposal / resampler / model / particle count the column route accepts.
(The golden-fixture suites in ``tests/test_filters_gpu.py`` run on both routes as well: ``kernel_route``.)"""
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _t(v, dtype):
    return torch.tensor(v, dtype=dtype, device=DEV)


def _model(kind, b, dtype):
    from pyfilter_amd import timeseries as ts
    from pyfilter_amd.timeseries import models

    if kind == "sine":
        return ts.LinearStateSpaceModel(models.SineDiffusion(_t(0.0, dtype), _t(1.0, dtype), dt=0.1), (_t(1.0, dtype), _t(0.1, dtype))), ()
    if kind == "lg":
        return ts.LinearStateSpaceModel(models.AR(_t(0.0, dtype), _t(0.99, dtype), _t(0.05, dtype)), (_t(1.0, dtype), _t(0.15, dtype))), ()
    if kind == "ou":
        kappa = _t([0.02 + 0.01 * (i % 5) for i in range(b)], dtype)
        gamma = _t([0.1 * (i % 3) for i in range(b)], dtype)
        sigma = _t([0.05 + 0.01 * (i % 4) for i in range(b)], dtype)
        return ts.LinearStateSpaceModel(models.OrnsteinUhlenbeck(kappa, gamma, sigma, dt=1.0), (_t(1.0, dtype), _t(0.05, dtype))), ()
    if kind == "sv":
        kappa = _t([0.05 + 0.01 * (i % 7) for i in range(b)], dtype)
        gamma = _t([1.0 + 0.1 * (i % 5) for i in range(b)], dtype)
        sigma = _t([0.10 + 0.02 * (i % 3) for i in range(b)], dtype)
        mu = _t([0.05 * (i % 4) for i in range(b)], dtype)
        return models.StochasticVolatilityModel(models.Verhulst(kappa, gamma, sigma, dt=0.2, initial=(_t(1.0, dtype), _t(0.1, dtype))), mu), ()
    if kind == "lorenz":
        hidden = models.Lorenz63(_t(10.0, dtype), _t(28.0, dtype), _t(8.0 / 3.0, dtype), _t(1.0, dtype), dt=0.01)
        a = _t([[0.8, 0.0, 0.0], [0.0, 0.0, 0.8]], dtype)
        return ts.LinearStateSpaceModel(hidden, (a, _t([0.0], dtype), _t([math.sqrt(0.1)], dtype)), torch.Size([2])), (2,)
    if kind == "rw2d":  # the reference's own 2-D model (tests/filters/models.py:28-52), one sigma row per filter
        sig = _t([[0.05 + 0.01 * (i % 3), 0.1 + 0.02 * (i % 2)] for i in range(b)], dtype)
        hidden = models.RandomWalk(sig if b > 1 else sig[0], dim=2)
        a = _t([[1.0, 0.0], [0.0, 1.0]], dtype)
        return ts.LinearStateSpaceModel(hidden, (a, _t([0.15, 0.15], dtype)), torch.Size([2])), (2,)
    if kind in _DENSE:
        # round 5: DENSE linear observations of a vector state - (hidden process, D, O); O = 0 is a scalar observation
        # (event_shape = Size([])); one noise-scale row per filter
        hid, d, o = _DENSE[kind]
        od = max(o, 1)
        gen = torch.Generator().manual_seed(17 * d + o)
        a = (0.5 * torch.randn(od, d, generator=gen) + (torch.eye(od, d) if od <= d else 0.0)).to(dtype).to(DEV)
        off = (0.2 * torch.randn(od, generator=gen)).to(dtype).to(DEV)
        sc = _t([[0.3 + 0.05 * ((i + k) % 4) for k in range(od)] for i in range(b)], dtype)
        if hid == "lorenz":
            hidden = models.Lorenz63(_t(10.0, dtype), _t(28.0, dtype), _t(8.0 / 3.0, dtype), _t(1.0, dtype), dt=0.01)
        else:
            hidden = models.RandomWalk(_t([0.05, 0.1, 0.07][:d], dtype), dim=d)
        if o == 0:
            return ts.LinearStateSpaceModel(hidden, (a[0], off[0], sc[:, 0] if b > 1 else sc[0, 0]), torch.Size([])), ()
        return ts.LinearStateSpaceModel(hidden, (a, off, sc if b > 1 else sc[0]), torch.Size([o])), (o,)
    raise KeyError(kind)


_DENSE = {"lorenz_o3": ("lorenz", 3, 3), "lorenz_s": ("lorenz", 3, 0), "lorenz_o1": ("lorenz", 3, 1), "rw3_o3": ("rw", 3, 3),
          "rw2_o3": ("rw", 2, 3), "rw2_o1": ("rw", 2, 1), "rw2_s": ("rw", 2, 0), "rw3_o2": ("rw", 3, 2)}


def _run(route, kind, filt_name, prop, resampler, n, b, t_len, dtype, nan_at=(), seed=7, ess=0.9, generic=False, oes=1):
    from pyfilter_amd import ops, resampling
    from pyfilter_amd.filters.particle import APF, SISR, proposals

    ssm, o = _model(kind, b, dtype)
    ssm.observe_every_step = oes
    cls = {"sisr": SISR, "apf": APF}[filt_name]
    p = {"bootstrap": proposals.Bootstrap, "lgo": proposals.LinearGaussianObservations}[prop]()
    rs = {"systematic": resampling.systematic, "multinomial": resampling.multinomial}[resampler]
    filt = cls(ssm, n, proposal=p, resampling=rs, seed=seed, ess_threshold=ess)
    if b > 1:
        filt.set_batch_shape(torch.Size([b]))
    g = torch.Generator().manual_seed(3)
    if kind == "lorenz":
        y = torch.tensor([-4.7, 19.6]) + 0.5 * torch.randn((t_len, 2), generator=g)
    elif kind in _DENSE and _DENSE[kind][0] == "lorenz":  # around the observation of the initial mean
        a_, b_, _ = (p.detach().cpu().double() for p in ssm.parameters)
        c0 = torch.tensor([-5.91652, -5.52332, 24.5723], dtype=torch.float64)
        loc = b_ + ((a_ * c0).sum(-1) if a_.dim() == 1 else a_ @ c0)
        y = loc + 0.5 * torch.randn((t_len,) + tuple(loc.shape), generator=g, dtype=torch.float64)
    elif kind == "sv":
        y = 0.05 + torch.randn((t_len,), generator=g)
    else:
        y = (0.1 * torch.randn((t_len,) + o, generator=g)).cumsum(0)
    y = y.to(dtype)
    for k in nan_at:
        y[k] = float("nan")
    from pyfilter_amd.hints import HINTS

    saved = (HINTS.route, HINTS.column_max_n)
    HINTS.column_max_n = 4096  # (the library hands columns beyond 2 048 particles to the per-step route: faster there)
    HINTS.route = 1 if route == "per_step" else (2 if generic else 0)
    try:
        res = filt.batch_filter(y.to(DEV), bar=False)
        torch.cuda.synchronize()
        trace = ops.debug_launch_trace(4)
    finally:
        HINTS.route, HINTS.column_max_n = saved
    last = res.latest_state
    return dict(means=res.filter_means.cpu(), var=res.filter_variance.cpu(), ll=res.loglikelihood.cpu(),
                x=last.timeseries_state.value.cpu(), w=last.weights.cpu(), idx=last.previous_indices.cpu(),
                ll_last=last.get_loglikelihood().cpu(), SPEC=trace[-1]["SPEC"], FAST=trace[-1]["FAST"])


CASES = [
    # kind, filter, proposal, resampler, N, B, T, NaN observations
    ("sine", "apf", "lgo", "systematic", 512, 5, 40, ()),
    ("sine", "apf", "bootstrap", "systematic", 256, 3, 30, (4, 5)),
    ("sine", "sisr", "bootstrap", "systematic", 400, 4, 40, (7,)),
    ("sine", "sisr", "lgo", "systematic", 64, 2, 30, ()),
    ("lg", "sisr", "bootstrap", "systematic", 1000, 1, 50, ()),
    ("lg", "apf", "lgo", "systematic", 2048, 2, 25, ()),
    ("lg", "apf", "lgo", "systematic", 3072, 3, 25, (0,)),    # 12 waves; (float64: 4096 particles exceed the 64 KB of LDS)
    ("ou", "apf", "lgo", "systematic", 1024, 7, 30, ()),
    ("ou", "sisr", "lgo", "systematic", 100, 3, 30, ()),       # N % 4 != 0: the RAGGED instantiation
    ("ou", "apf", "bootstrap", "systematic", 333, 2, 20, (3,)),
    ("sv", "apf", "bootstrap", "systematic", 512, 6, 40, ()),
    ("sv", "sisr", "bootstrap", "systematic", 256, 4, 40, ()),
    ("lorenz", "sisr", "bootstrap", "systematic", 512, 2, 20, ()),
    ("lorenz", "apf", "lgo", "systematic", 256, 2, 15, (2,)),
    ("lorenz", "apf", "bootstrap", "systematic", 1536, 1, 10, ()),
    ("rw2d", "sisr", "bootstrap", "systematic", 512, 3, 40, ()),
    ("rw2d", "apf", "lgo", "systematic", 1024, 2, 30, (3, 4)),
    ("rw2d", "sisr", "lgo", "systematic", 333, 2, 30, ()),     # D = 2, N % 4 != 0: the RAGGED instantiation
    ("rw2d", "apf", "bootstrap", "systematic", 2048, 1, 20, (0,)),
    ("rw2d", "apf", "lgo", "multinomial", 256, 3, 25, ()),
    ("sine", "sisr", "bootstrap", "multinomial", 512, 3, 30, ()),
    ("sine", "apf", "lgo", "multinomial", 1024, 2, 30, ()),
    ("lorenz", "sisr", "bootstrap", "multinomial", 256, 2, 15, ()),
    ("lg", "sisr", "bootstrap", "systematic", 2, 3, 12, ()),
    ("lg", "apf", "bootstrap", "systematic", 1, 2, 8, ()),
    # round 5: dense observation matrices for every (D, O) the kernels accept, a scalar observation of a vector state included
    ("lorenz_o3", "apf", "lgo", "systematic", 512, 2, 15, (3,)),
    ("lorenz_o3", "sisr", "lgo", "systematic", 333, 3, 15, ()),
    ("lorenz_s", "sisr", "lgo", "systematic", 256, 2, 15, ()),
    ("lorenz_s", "apf", "lgo", "systematic", 256, 3, 15, (5,)),
    ("lorenz_o1", "apf", "lgo", "multinomial", 512, 2, 15, ()),
    ("rw3_o3", "apf", "lgo", "systematic", 1024, 3, 30, ()),
    ("rw3_o2", "sisr", "lgo", "systematic", 700, 2, 30, (9,)),
    ("rw2_o3", "apf", "lgo", "systematic", 512, 4, 30, ()),
    ("rw2_o3", "sisr", "bootstrap", "systematic", 333, 2, 30, ()),
    ("rw2_o1", "apf", "lgo", "systematic", 256, 2, 30, (0, 1)),
    ("rw2_s", "sisr", "lgo", "systematic", 2048, 1, 20, ()),
    ("rw2_s", "apf", "bootstrap", "multinomial", 400, 3, 20, ()),
]


@pytest.mark.parametrize("kind,filt_name,prop,resampler,n,b,t_len,nan_at", CASES)
def test_column_route_equals_per_step_route_float64(kind, filt_name, prop, resampler, n, b, t_len, nan_at):
    col = _run("column", kind, filt_name, prop, resampler, n, b, t_len, torch.float64, nan_at)
    ref = _run("per_step", kind, filt_name, prop, resampler, n, b, t_len, torch.float64, nan_at)
    assert col["SPEC"] == 9 and ref["SPEC"] != 9, "the routes under test did not run"
    assert torch.equal(col["idx"], ref["idx"]), "final ancestors differ"
    tol = dict(rtol=1e-9, atol=1e-11)
    torch.testing.assert_close(col["means"], ref["means"], **tol)
    torch.testing.assert_close(col["var"], ref["var"], rtol=1e-8, atol=1e-11)
    torch.testing.assert_close(col["ll"], ref["ll"], rtol=1e-9, atol=1e-9)
    torch.testing.assert_close(col["ll_last"], ref["ll_last"], rtol=1e-9, atol=1e-9)
    torch.testing.assert_close(col["x"], ref["x"], **tol)
    torch.testing.assert_close(col["w"], ref["w"], equal_nan=True, **tol)


F32_CASES = [
    ("sine", "apf", "lgo", "systematic", 512, 5, 40, ()),
    ("sine", "sisr", "bootstrap", "systematic", 400, 4, 40, (7,)),
    ("lg", "sisr", "bootstrap", "systematic", 1000, 1, 50, ()),
    ("lg", "apf", "lgo", "systematic", 4096, 3, 25, (0,)),     # the largest column the route takes: 16 waves
    ("ou", "apf", "lgo", "systematic", 1024, 7, 30, ()),
    ("sv", "apf", "bootstrap", "systematic", 512, 6, 40, ()),
    ("sine", "apf", "lgo", "multinomial", 1024, 2, 30, ()),
]  # (D = 2 in float32 on this route: test_kalman_statistical_parity_2d_philox - the long memory of a random walk makes two
#    independent runs differ by more than the one-step standard errors this test allows)


@pytest.mark.parametrize("kind,filt_name,prop,resampler,n,b,t_len,nan_at", F32_CASES)
def test_column_route_float32_within_monte_carlo_error_of_float64(kind, filt_name, prop, resampler, n, b, t_len, nan_at):
    """float32 production arithmetic: an ulp in a weight moves an ancestor across a cdf boundary, after which the runs are
    different - equally valid - Monte-Carlo runs.  Bar: the float32 column run within 8 Monte-Carlo standard errors of the
    float64 run (float32 and float64 Philox normals are different numbers: two independent runs; the bar of ``test_fused_batch_filter_matches_reference`` for float32 fixtures)."""
    c32 = _run("column", kind, filt_name, prop, resampler, n, b, t_len, torch.float32, nan_at)
    c64 = _run("column", kind, filt_name, prop, resampler, n, b, t_len, torch.float64, nan_at)
    assert c32["SPEC"] == 9
    se = (c64["var"] / n).sqrt()
    diff = (c32["means"].double() - c64["means"]).abs()
    assert (diff <= 8.0 * se + 1e-4 * c64["means"].abs() + 1e-5).all(), (diff / (se + 1e-12)).max()
    assert torch.isfinite(c32["ll"]).all()
    assert ((c32["ll"].double() - c64["ll"]).abs() <= 0.08 * math.sqrt(t_len) * max(1.0, float(c64["ll"].abs().max()) / t_len) + 1e-2).all()


@pytest.mark.parametrize("n", [512, 333, 1502])
@pytest.mark.parametrize("resampler", ["systematic", "multinomial"])
@pytest.mark.parametrize("prop", ["bootstrap", "lgo"])
@pytest.mark.parametrize("filt_name", ["sisr", "apf"])
@pytest.mark.parametrize("kind", ["lg", "sine", "ou", "sv", "lorenz"])
def test_specialised_column_kernels_equal_the_run_time_kernel(kind, filt_name, prop, resampler, n):
    """float32 runs of the scalar closed-form models take instantiations of the column kernel with the model kind, filter
    and proposal as compile-time constants (``pf_column.hpp``: KIND / FILT / PROP).  Same draws, same arithmetic: they
    must reproduce the run-time kernel (``pf_run_hints.route = PF_ROUTE_COLUMN_GENERIC``) - NaN observations included - and be the ones that ran."""
    if kind == "sv" and prop == "lgo":
        pytest.skip("the stochastic-volatility observation has no linear-Gaussian proposal")
    if kind == "lorenz" and n % 4:
        pytest.skip("Lorenz columns of N % 4 != 0 particles: the run-time kernel's RAGGED instantiation (no specialised twin)")
    # (n = 333: the RAGGED instantiations - four particles per lane, N % 4 != 0; n = 1502: ragged AND the 1024-thread bound)
    b, t_len, nan_at = (5, 40, (3, 17)) if n < 1024 else (3, 12, (3,))
    spec = _run("column", kind, filt_name, prop, resampler, n, b, t_len, torch.float32, nan_at)
    gen = _run("column", kind, filt_name, prop, resampler, n, b, t_len, torch.float32, nan_at, generic=True)
    assert spec["SPEC"] == 9 and spec["FAST"] == 1 and gen["SPEC"] == 9 and gen["FAST"] == 0
    if kind == "lorenz" and prop == "bootstrap":
        # the Lorenz drift folds into different fused multiply-adds once its kind is a constant: an ulp in a chaotic state
        # moves an ancestor, after which the two are different - equally valid - Monte-Carlo runs.  What can be pinned: ONE
        # move from the same state (same ancestors, new particles and moments to float rounding), and that long runs stay finite.
        assert torch.isfinite(spec["ll"]).all() and torch.isfinite(spec["means"]).all()
        one_s = _run("column", kind, filt_name, prop, resampler, n, b, 1, torch.float32, generic=True)
        one_f = _run("column", kind, filt_name, prop, resampler, n, b, 1, torch.float32)
        assert one_s["FAST"] == 0 and one_f["FAST"] == 1
        assert torch.equal(one_s["idx"], one_f["idx"])
        torch.testing.assert_close(one_f["x"], one_s["x"], rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(one_f["means"], one_s["means"], rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(one_f["ll"], one_s["ll"], rtol=1e-4, atol=1e-3)
        return
    assert torch.equal(spec["idx"], gen["idx"]), "ancestors differ"
    torch.testing.assert_close(spec["means"], gen["means"], rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(spec["ll"], gen["ll"], rtol=1e-6, atol=1e-5)
    torch.testing.assert_close(spec["x"], gen["x"], rtol=1e-6, atol=1e-7)


def test_runs_longer_than_one_launch_carries_the_state_through():
    """2048 steps fit one launch's baked-in observed flags; a longer run is several launches with the state handed over in
    HBM: identical to the per-step route, NaN observations on either side of the seam."""
    t_len = 2100
    nan_at = (5, 2047, 2048, 2060)
    col = _run("column", "lg", "sisr", "bootstrap", "systematic", 128, 2, t_len, torch.float64, nan_at)
    ref = _run("per_step", "lg", "sisr", "bootstrap", "systematic", 128, 2, t_len, torch.float64, nan_at)
    assert col["means"].shape[0] == t_len + 1
    torch.testing.assert_close(col["means"], ref["means"], rtol=1e-9, atol=1e-11)
    torch.testing.assert_close(col["ll"], ref["ll"], rtol=1e-9, atol=1e-8)
    assert torch.equal(col["idx"], ref["idx"])


def test_column_route_against_the_oracle_on_taped_draws():
    """Independent of the per-step route: the column kernel on injected draws against ``oracle/cpu_ref.py`` (float64, a
    shape no golden fixture has: 2 000 particles x 3 filters, four particles per thread, 8 waves)."""
    from oracle import cpu_ref, models as M
    from pyfilter_amd import ops, timeseries as ts
    from pyfilter_amd.filters.particle import APF, proposals
    from pyfilter_amd.timeseries import models

    n, b, t_len, dtype = 2000, 3, 12, torch.float64
    g = torch.Generator().manual_seed(5)
    y = (0.3 * torch.randn(t_len, generator=g, dtype=dtype)).cumsum(0)
    z0 = torch.randn(n, b, generator=g).to(dtype)
    z = torch.randn(t_len, n, b, generator=g).to(dtype)
    u = torch.rand(t_len, b, generator=g).to(dtype)
    ssm = ts.LinearStateSpaceModel(models.SineDiffusion(_t(0.0, dtype), _t(1.0, dtype), dt=0.1), (_t(1.0, dtype), _t(0.1, dtype)))
    filt = APF(ssm, n, proposal=proposals.LinearGaussianObservations())
    filt.set_batch_shape(torch.Size([b]))
    filt.set_tape(z=z, u=u, z0=z0)
    res = filt.batch_filter(y.to(DEV), bar=False)
    torch.cuda.synchronize()
    assert ops.debug_launch_trace(1)[-1]["SPEC"] == 9
    spec = M.ModelSpec(M.HID_SINE_EM, (0.0, 1.0), 0, 0.1, (0.0, 1.0), M.OBS_LINEAR, (1.0, 0.0, 0.1), 0)
    x0 = M.initial_sample(spec, z0)
    ref = cpu_ref.batch_filter(spec, "apf", "lgo", y, x0, z, u)
    torch.testing.assert_close(res.filter_means.cpu(), ref["filter_means"], rtol=1e-9, atol=1e-11)
    torch.testing.assert_close(res.loglikelihood.cpu(), ref["loglikelihood"], rtol=1e-9, atol=1e-9)
    assert torch.equal(res.latest_state.previous_indices.cpu(), ref["prev_inds"])


def test_smc2_shaped_workload_is_one_launch_per_run():
    """The reference's own operating point (1 000 theta x 400 particles): ``batch_filter`` is ONE kernel launch."""
    from pyfilter_amd import ops

    r = _run("column", "ou", "apf", "lgo", "systematic", 400, 1000, 60, torch.float32)
    assert r["SPEC"] == 9 and torch.isfinite(r["ll"]).all() and torch.isfinite(r["means"]).all()
    recs = ops.debug_launch_trace(64)
    assert recs[-1]["SPEC"] == 9 and (len(recs) < 2 or recs[-2]["SPEC"] != 9 or recs[-2]["step"] == 0)


def test_random_cross_route_sweep():
    """90 random (model, filter, proposal, resampler, N in [1, 2048], B, T, NaN pattern, ESS threshold) configurations, float64,
    Philox draws: the column-persistent kernel and the per-step kernels consume the same random numbers and must agree -
    identical ancestors, 1e-9.  Unlike ``tools/fuzz_parity.py`` (oracle, taped draws, systematic only) this covers the
    multinomial resampler and the kernels' own generators."""
    import random

    rng = random.Random(5)
    for i in range(90):
        kind = rng.choice(["sine", "lg", "ou", "sv", "lorenz", "rw2d"] + sorted(_DENSE))
        filt_name = rng.choice(["sisr", "apf"])
        prop = "bootstrap" if kind == "sv" else rng.choice(["bootstrap", "lgo"])
        resampler = rng.choice(["systematic", "systematic", "multinomial"])
        n = rng.choice([rng.randint(1, 70), rng.randint(71, 700), rng.randint(701, 2048), rng.choice([64, 256, 1024, 2048])])
        if kind == "lorenz" or (kind in _DENSE and _DENSE[kind][1] == 3):
            n = min(n, 1536)  # (float64, D = 3: the cdf + three particle planes of 2 048 particles exceed the 64 KB of LDS)
        b = rng.choice([1, 2, 3, 7, 33])
        t_len = rng.randint(1, 30)
        nan_at = tuple(k for k in range(t_len) if rng.random() < 0.12)
        ess = rng.choice([0.1, 0.5, 0.9])
        oes = rng.choice([1, 1, 1, 2, 3])  # observe_every_step
        tag = f"#{i} {kind} {filt_name} {prop} {resampler} N={n} B={b} T={t_len} nan={nan_at} ess={ess} oes={oes}"
        col = _run("column", kind, filt_name, prop, resampler, n, b, t_len, torch.float64, nan_at, seed=100 + i, ess=ess, oes=oes)
        ref = _run("per_step", kind, filt_name, prop, resampler, n, b, t_len, torch.float64, nan_at, seed=100 + i, ess=ess, oes=oes)
        assert col["SPEC"] == 9 and ref["SPEC"] != 9, tag
        assert torch.equal(col["idx"], ref["idx"]), tag + ": final ancestors differ"
        torch.testing.assert_close(col["means"], ref["means"], rtol=1e-9, atol=1e-11, equal_nan=True, msg=lambda m: tag + ": " + m)
        torch.testing.assert_close(col["ll"], ref["ll"], rtol=1e-9, atol=1e-9, equal_nan=True, msg=lambda m: tag + ": " + m)
        torch.testing.assert_close(col["w"], ref["w"], rtol=1e-9, atol=1e-11, equal_nan=True, msg=lambda m: tag + ": " + m)
