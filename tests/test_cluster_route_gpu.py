"""The column-CLUSTER route (``pyfilter_amd/csrc/pf_cluster.hpp``: a filter of 2 049 .. 16 384 particles held in the registers of
``ceil(N / 1024)`` workgroups for a whole run, one record per wave and step handed between them) against the per-step route
(``k_fused_step``) and the oracle.

* float64: the routes key their Philox draws alike, so a run with the same seed consumes the SAME numbers on either - identical
  ancestors, moments and log-likelihoods to 1e-9 for every filter / proposal / model / size the route accepts: columns that end
  inside a member workgroup or a wave, batches that are no multiple of the eight XCDs, batches of more workgroups than the chip
  holds (consecutive launches), missing observations, ``observe_every_step > 1``, degenerate weights (the ancestors of a member lie
  many chunks apart: several staged windows per step).  ``tools/fuzz_parity.py`` with ``FUZZ_CLUSTER=1`` runs the same route
  against the ORACLE on taped draws (``test_cluster_fuzz_against_the_oracle``).
* float32: the production instantiations (model kind / filter / proposal folded at compile time) on their own Philox draws,
  teacher-forced oracle in float64 and float32 from the kernel's own state to the next - the procedure of
  ``tests/test_column_production_gpu.py`` (``own_draws_check``), launch trace ``SPEC == 10``."""
import importlib.util
import os

import pytest
import torch

from tests.test_column_production_gpu import own_draws_check
from tests.test_column_route_gpu import _DENSE, _model

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _run(route, kind, filt_name, prop, n, b, t_len, dtype=torch.float64, nan_at=(), seed=11, ess=0.9, oes=1, jumpy=False):
    from pyfilter_amd import ops, resampling
    from pyfilter_amd.filters.particle import APF, SISR, proposals
    from pyfilter_amd.hints import HINTS

    ssm, o = _model(kind, b, dtype)
    ssm.observe_every_step = oes
    cls = {"sisr": SISR, "apf": APF}[filt_name]
    p = {"bootstrap": proposals.Bootstrap, "lgo": proposals.LinearGaussianObservations}[prop]()
    filt = cls(ssm, n, proposal=p, resampling=resampling.systematic, seed=seed, ess_threshold=ess)
    if b > 1:
        filt.set_batch_shape(torch.Size([b]))
    g = torch.Generator().manual_seed(5)
    if kind == "lorenz":
        y = torch.tensor([-4.7, 19.6]) + 0.5 * torch.randn((t_len, 2), generator=g)
    elif kind in _DENSE and _DENSE[kind][0] == "lorenz":
        a_, b_, _ = (q.detach().cpu().double() for q in ssm.parameters)
        c0 = torch.tensor([-5.91652, -5.52332, 24.5723], dtype=torch.float64)
        loc = b_ + ((a_ * c0).sum(-1) if a_.dim() == 1 else a_ @ c0)
        y = loc + 0.5 * torch.randn((t_len,) + tuple(loc.shape), generator=g, dtype=torch.float64)
    elif kind == "sv":
        y = 0.05 + torch.randn((t_len,), generator=g)
    else:
        y = (0.1 * torch.randn((t_len,) + o, generator=g)).cumsum(0)
        if jumpy:  # observations far out in the particles' tail: a handful of particles carry the weight of a step
            y = y + 0.9 * torch.tensor([(-1.0) ** k for k in range(t_len)]).reshape((t_len,) + (1,) * len(o))
    y = y.to(dtype)
    for k in nan_at:
        y[k] = float("nan")
    saved = HINTS.route
    HINTS.route = {"per_step": 1, "cluster": 4, "spread": 5}[route]  # (4: PF_ROUTE_CLUSTER_ALWAYS, 5: PF_ROUTE_CLUSTER_SPREAD)
    try:
        res = filt.batch_filter(y.to(DEV), bar=False)
        torch.cuda.synchronize()
        trace = ops.debug_launch_trace(4)
    finally:
        HINTS.route = saved
    last = res.latest_state
    return dict(means=res.filter_means.cpu(), var=res.filter_variance.cpu(), ll=res.loglikelihood.cpu(),
                x=last.timeseries_state.value.cpu(), w=last.weights.cpu(), idx=last.previous_indices.cpu(),
                SPEC=trace[-1]["SPEC"], FAST=trace[-1]["FAST"])


CASES = [
    # kind, filter, proposal, N, B, T, options
    ("sine", "apf", "lgo", 8192, 3, 12, {}),
    ("sine", "apf", "bootstrap", 4096, 9, 10, dict(nan_at=(3, 4))),        # B no multiple of 8
    ("lg", "sisr", "bootstrap", 4096, 5, 25, dict(ess=0.5)),              # SISR: moves with and without resampling
    ("lg", "sisr", "lgo", 2052, 2, 12, dict(ess=0.97)),                  # the column ends 4 particles into its third member
    ("ou", "apf", "lgo", 8196, 7, 8, dict(nan_at=(0,))),                 # ... and 4 particles into a ninth
    ("ou", "apf", "bootstrap", 3000, 2, 10, {}),                         # ends inside a wave
    ("sv", "apf", "bootstrap", 5120, 9, 10, {}),
    ("sv", "sisr", "bootstrap", 16384, 2, 8, dict(ess=0.9)),              # the largest column: 64 chunks, 16 members
    ("lorenz", "sisr", "bootstrap", 3000, 2, 10, dict(ess=0.7)),
    ("lorenz", "apf", "lgo", 4096, 1, 8, dict(nan_at=(2,))),             # float64, D = 3: 106 KB of LDS, one member per CU
    ("rw2d", "apf", "lgo", 16380, 2, 6, {}),
    ("rw3_o3", "sisr", "lgo", 12288, 2, 6, dict(nan_at=(1,), ess=0.5)),
    ("lorenz_s", "apf", "bootstrap", 8192, 2, 6, {}),                    # scalar observation of a vector state
    ("sine", "sisr", "bootstrap", 8192, 2, 10, dict(oes=3, ess=0.8)),     # observe_every_step: propagate-only moves in the loop
    ("sine", "apf", "bootstrap", 8192, 3, 10, dict(jumpy=True)),         # degenerate weights: several windows per step
    ("lg", "sisr", "bootstrap", 16384, 2, 10, dict(jumpy=True, ess=0.97)),
    ("lg", "apf", "lgo", 4096, 300, 4, {}),                              # 1 200 member workgroups: more than resident - two launches
]


@pytest.mark.parametrize("kind,filt_name,prop,n,b,t_len,opt", CASES)
def test_cluster_route_equals_per_step_route_float64(kind, filt_name, prop, n, b, t_len, opt):
    ref = _run("per_step", kind, filt_name, prop, n, b, t_len, **opt)
    got = _run("cluster", kind, filt_name, prop, n, b, t_len, **opt)
    assert got["SPEC"] == 10 and ref["SPEC"] != 10, (got["SPEC"], ref["SPEC"])
    assert torch.isfinite(got["ll"]).all(), "a cluster launch gave up waiting (NaN log-likelihood = its error word)"
    assert torch.equal(got["idx"], ref["idx"]), f"{(got['idx'] != ref['idx']).sum().item()} ancestors differ"
    torch.testing.assert_close(got["x"], ref["x"], rtol=1e-9, atol=1e-11)
    torch.testing.assert_close(got["w"], ref["w"], rtol=1e-9, atol=1e-9)
    torch.testing.assert_close(got["means"], ref["means"], rtol=1e-9, atol=1e-11)
    torch.testing.assert_close(got["var"], ref["var"], rtol=1e-7, atol=1e-11)
    torch.testing.assert_close(got["ll"], ref["ll"], rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("kind,filt_name,prop,n,b,t_len,opt", [c for c in CASES if c[4] <= 9])
def test_cluster_members_on_different_xcds_float64(kind, filt_name, prop, n, b, t_len, opt):
    """``PF_ROUTE_CLUSTER_SPREAD``: the members of a filter on eight different XCDs, the exchange on agent-scope write-through stores
    and L1-bypassing loads only - the form a run falls back to when its members do not share an XCD - must give what the same-XCD
    fast path gives, bit for bit (the arithmetic is the same; only the memory path differs)."""
    fast = _run("cluster", kind, filt_name, prop, n, b, t_len, **opt)
    got = _run("spread", kind, filt_name, prop, n, b, t_len, **opt)
    assert got["SPEC"] == 10 and torch.isfinite(got["ll"]).all()
    for key in ("idx", "x", "w", "means", "var", "ll"):
        assert torch.equal(got[key], fast[key]) or torch.allclose(got[key], fast[key], rtol=0, atol=0, equal_nan=True), key


@pytest.mark.parametrize("kind,dtype,n", [("sine", torch.float32, 8192), ("lorenz", torch.float64, 4096), ("rw2d", torch.float64, 5120)])
def test_cluster_runs_replayed_from_a_captured_graph(kind, dtype, n):
    """``observe_every_step = 2`` keeps a run on the general fused driver, whose repeated runs of one configuration replay a captured
    hipGraph: the cluster launches (and, for float64 vector states, their > 64 KB of dynamic LDS) are captured like any other
    kernel - run for run the same numbers as the per-step route's replayed graph (float64) / finite and fresh (float32)."""
    from pyfilter_amd import resampling
    from pyfilter_amd.filters.particle import APF, proposals
    from pyfilter_amd.hints import HINTS

    outs = {}
    for route in ("per_step", "cluster"):
        ssm, o = _model(kind, 2, dtype)
        ssm.observe_every_step = 2
        filt = APF(ssm, n, proposal=proposals.Bootstrap(), resampling=resampling.systematic, seed=21)
        filt.set_batch_shape(torch.Size([2]))
        g = torch.Generator().manual_seed(8)
        y = (torch.tensor([-4.7, 19.6]) + 0.5 * torch.randn((6, 2), generator=g)) if kind == "lorenz" else (0.1 * torch.randn((6,) + o, generator=g)).cumsum(0)
        saved = HINTS.route
        HINTS.route = 1 if route == "per_step" else 4
        try:
            outs[route] = [filt.batch_filter(y.to(dtype).to(DEV), bar=False) for _ in range(4)]
            torch.cuda.synchronize()
            assert any(pl.graph is not None for pl in filt._fused_plans.values()), "the run was not replayed from a graph"
        finally:
            HINTS.route = saved
    for rep in range(4):
        a, b = outs["cluster"][rep], outs["per_step"][rep]
        assert torch.isfinite(a.loglikelihood).all()
        if dtype == torch.float64:
            assert torch.equal(a.latest_state.previous_indices, b.latest_state.previous_indices)
            torch.testing.assert_close(a.filter_means, b.filter_means, rtol=1e-9, atol=1e-11)
            torch.testing.assert_close(a.loglikelihood, b.loglikelihood, rtol=1e-9, atol=1e-9)
    assert not torch.equal(outs["cluster"][1].filter_means, outs["cluster"][2].filter_means), "replays must draw fresh numbers"


@pytest.mark.parametrize("pieces", [[(3, 0), (2, 1), (1, 0), (None, 1)], [(1, 1), (4, 0), (None, 1)], [(1, 1)] * 8])
@pytest.mark.parametrize("kind,filt_name,prop,n,b", [("sine", "apf", "lgo", 4096, 3), ("lg", "sisr", "bootstrap", 8192, 2),
                                                      ("lorenz", "sisr", "bootstrap", 3000, 2)])
def test_a_run_issued_in_pieces_that_alternate_between_the_cluster_and_the_per_step_kernels(kind, filt_name, prop, n, b, pieces):
    """``pf_filter_run`` pieces on ONE argument block with mixed ``finalize``: a piece that is not self-contained runs on the
    per-step kernels, a self-contained one of a filter of this size on the cluster kernel (started at ``t0 > 0``: it flushes the
    increment its predecessor left pending, and leaves the column record a per-step successor starts from) - and online moves, one
    cluster launch each.  Same draws (keyed by the absolute step): the one-piece run's numbers."""
    from pyfilter_amd import ops, resampling
    from pyfilter_amd.filters.particle import APF, SISR, proposals
    from pyfilter_amd.hints import HINTS

    outs = {}
    for mode in ("one", "pieces"):
        ssm, o = _model(kind, b, torch.float64)
        p = {"bootstrap": proposals.Bootstrap, "lgo": proposals.LinearGaussianObservations}[prop]()
        filt = {"sisr": SISR, "apf": APF}[filt_name](ssm, n, proposal=p, resampling=resampling.systematic, seed=31, ess_threshold=0.8)
        filt.set_batch_shape(torch.Size([b]))
        g = torch.Generator().manual_seed(9)
        y = (torch.tensor([-4.7, 19.6]) + 0.5 * torch.randn((8, 2), generator=g)) if kind == "lorenz" else (0.1 * torch.randn((8,) + o, generator=g)).cumsum(0)
        y[5] = float("nan")
        gen = torch.Generator().manual_seed(4)
        filt.set_tape(u=torch.rand((8, b), generator=gen, dtype=torch.float64))  # (the offsets: tape-indexed, so every piece reads its own)
        saved = HINTS.route
        HINTS.route = 1 if mode == "one" else 4
        try:
            if mode == "pieces":
                filt._move_by_move = pieces
            res = filt.batch_filter(y.double().to(DEV), bar=False)
            torch.cuda.synchronize()
            specs = {r["SPEC"] for r in ops.debug_launch_trace(64)[-8:]}
        finally:
            HINTS.route = saved
        outs[mode] = (res, specs)
    assert 10 in outs["pieces"][1], outs["pieces"][1]
    a, r = outs["pieces"][0], outs["one"][0]
    assert torch.equal(a.latest_state.previous_indices, r.latest_state.previous_indices)
    torch.testing.assert_close(a.filter_means, r.filter_means, rtol=1e-9, atol=1e-11)
    torch.testing.assert_close(a.filter_variance, r.filter_variance, rtol=1e-7, atol=1e-11)
    torch.testing.assert_close(a.loglikelihood, r.loglikelihood, rtol=1e-9, atol=1e-9)
    torch.testing.assert_close(a.latest_state.weights, r.latest_state.weights, rtol=1e-9, atol=1e-9)


def test_cluster_fuzz_against_the_oracle(monkeypatch):
    """``tools/fuzz_parity.py`` with the cluster route's sizes: 20 random (model, filter, proposal, threshold, B, T, NaN pattern,
    observe_every_step, driver) configurations in float64 on taped draws - means / log-likelihood to 1e-9 of the ORACLE, identical
    ancestors."""
    from pyfilter_amd.hints import HINTS

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_parity.py")
    spec = importlib.util.spec_from_file_location("fuzz_parity_cluster", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    monkeypatch.setattr("sys.argv", ["fuzz_parity.py", "20", "21"])
    monkeypatch.setenv("FUZZ_CLUSTER", "1")
    saved = (HINTS.route, HINTS.tile_target, HINTS.column_max_n)
    try:
        assert mod.main() == 0
    finally:
        HINTS.route, HINTS.tile_target, HINTS.column_max_n = saved


def _own_cases():
    out = []
    for kind in ("lg", "sine", "ou"):
        for filt_name in ("sisr", "apf"):
            for prop in ("bootstrap", "lgo"):
                out.append((kind, filt_name, prop, 4096 if kind == "lg" else (8192 if kind == "ou" else 2052)))
    # not specialised, float32 all the same: stochastic volatility, Lorenz-63, the 2-D random walk
    out += [("sv", "apf", "bootstrap", 4096), ("sv", "sisr", "bootstrap", 3000), ("lorenz", "apf", "lgo", 4096),
            ("lorenz", "sisr", "bootstrap", 2052), ("rw2d", "apf", "lgo", 4096), ("rw2d", "sisr", "bootstrap", 8192)]
    return out


@pytest.mark.parametrize("kind,filt_name,prop,n", _own_cases())
def test_production_cluster_kernels_match_oracle_on_their_own_draws(kind, filt_name, prop, n):
    own_draws_check(kind, filt_name, prop, n, 10, 1 if kind in ("lg", "sine", "ou") else 0, bt=(3, 7))


def test_auto_route_takes_the_cluster_kernel_where_it_pays():
    """The default hints (``HINTS.cluster``): 2 049 .. 16 384 particles and at most two launches' worth of member workgroups -
    beyond that, and below, the other routes."""
    from pyfilter_amd.hints import HINTS

    assert HINTS.kernel_route() == 3 and HINTS.cluster_takes(8192, 128) and HINTS.cluster_takes(8192, 256)
    assert not HINTS.cluster_takes(8192, 257) and not HINTS.cluster_takes(2048, 8) and not HINTS.cluster_takes(16388, 2)
    assert not HINTS.cluster_takes(8190, 2)
    from pyfilter_amd import ops, resampling, timeseries as ts
    from pyfilter_amd.filters.particle import APF, proposals
    from pyfilter_amd.timeseries import models

    t = lambda v: torch.tensor(v, dtype=torch.float32, device=DEV)  # noqa: E731
    ssm = ts.LinearStateSpaceModel(models.AR(t(0.0), t(0.99), t(0.05)), (t(1.0), t(0.15)))
    y = (0.1 * torch.randn(6)).cumsum(0).to(DEV)
    for n, b, spec in ((8192, 4, 10), (2048, 4, 9), (8192, 300, None), (32768, 2, None)):
        f = APF(ssm, n, proposal=proposals.LinearGaussianObservations(), resampling=resampling.systematic, seed=3)
        f.set_batch_shape(torch.Size([b]))
        res = f.batch_filter(y, bar=False)
        torch.cuda.synchronize()
        got = ops.debug_launch_trace(1)[-1]["SPEC"]
        assert (got == spec) if spec is not None else (got not in (9, 10)), (n, b, got)
        assert torch.isfinite(res.loglikelihood).all()


# ---- round 6: the route is safe to take by default ---------------------------------------------------------------------------------
def _concurrent_filter(seed, b=128, n=8192):
    from pyfilter_amd import resampling, timeseries as ts
    from pyfilter_amd.filters.particle import APF, proposals
    from pyfilter_amd.timeseries import models

    t = lambda v: torch.tensor(v, dtype=torch.float32, device=DEV)  # noqa: E731
    ssm = ts.LinearStateSpaceModel(models.AR(t(0.0), t(0.99), t(0.05)), (t(1.0), t(0.15)))
    f = APF(ssm, n, proposal=proposals.LinearGaussianObservations(), resampling=resampling.systematic, seed=seed)
    f.set_batch_shape(torch.Size([b]))
    return f


def test_two_threads_on_two_streams_run_cluster_filters_concurrently():
    """different threads may each drive their own filter (the reference's thread-local ``InferenceContext``,
    ``pyfilter/inference/context.py:41-48``, ``tests/inference/test_context.py:183-194``: a thread pool).  Two Python threads, each
    on its own ``torch.cuda.Stream``, each filtering 128 x 8 192 particles for T = 200 on the DEFAULT hints - every run a
    column-cluster launch of 1 024 workgroups, i.e. each launch alone fills the chip's resident slots.  The grouped workgroup ids
    (pf_cluster.hpp) make the resident workgroups of either launch whole filters: both finish, finite, and equal - bit for bit, the
    draws are keyed by (seed, step, particle) - to the same filters run alone; no launch gives up (``cluster_fallbacks``)."""
    import threading

    from pyfilter_amd import ops
    from pyfilter_amd.hints import HINTS

    assert HINTS.kernel_route() == 3 and HINTS.cluster_takes(8192, 128)
    g = torch.Generator().manual_seed(17)
    y = (0.1 * torch.randn(200, generator=g)).cumsum(0).to(DEV)

    def run_alone(seed):
        f = _concurrent_filter(seed)
        r = f.batch_filter(y, bar=False)
        torch.cuda.synchronize()
        assert ops.debug_launch_trace(1)[-1]["SPEC"] == 10
        return r.loglikelihood.cpu(), r.filter_means.cpu()

    alone = {seed: run_alone(seed) for seed in (41, 42)}
    out, errors = {}, []
    start = threading.Barrier(2)

    def worker(seed):
        try:
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                f = _concurrent_filter(seed)
                start.wait()
                reps = [f.batch_filter(y, bar=False) for _ in range(3)]  # (three runs each: the launches overlap whatever the start skew)
                stream.synchronize()
                out[seed] = (reps, getattr(f, "cluster_fallbacks", 0), ops.debug_launch_trace(1)[-1]["SPEC"])
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(seed,)) for seed in (41, 42)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    for seed in (41, 42):
        reps, fallbacks, spec = out[seed]
        assert spec == 10 and fallbacks == 0, (spec, fallbacks)
        for r in reps:
            assert torch.isfinite(r.loglikelihood).all() and torch.isfinite(r.filter_means).all()
        # the first run of a fresh filter object consumes the same draw epoch as the run alone did
        assert torch.equal(reps[0].loglikelihood.cpu(), alone[seed][0])
        assert torch.equal(reps[0].filter_means.cpu(), alone[seed][1])


@pytest.mark.parametrize("driver", ["batch_filter", "filter_block", "general"])
def test_a_cluster_launch_that_gives_up_falls_back_to_the_per_step_route(driver, monkeypatch):
    """``pf_run_hints.cluster_patience = -1``: a member that does not find its siblings' records at its first look gives up - the
    launch reports through ``pf_filter_args.status`` (NaN log-likelihoods, bit 0), and every driver re-issues the piece on the
    per-step route from the same incoming state on the same draws: the result IS the per-step route's."""
    import warnings

    from pyfilter_amd.hints import HINTS

    g = torch.Generator().manual_seed(23)
    y = (0.1 * torch.randn(12, generator=g)).cumsum(0).to(DEV)
    y[5] = float("nan")

    def run(route, patience):
        monkeypatch.setattr(HINTS, "route", route)
        monkeypatch.setattr(HINTS, "cluster_patience", patience)
        f = _concurrent_filter(7, b=6, n=8192)
        if driver == "general":
            f._time_kernels = True  # (the general fused driver: pf_filter_run_timed)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            if driver == "filter_block":
                state = f.initialize()
                res, ll, _ = f.filter_block(y, state)
            else:
                res = f.batch_filter(y, bar=False)
        torch.cuda.synchronize()
        return res, getattr(f, "cluster_fallbacks", 0)

    ref, _ = run(1, 0)
    got, fallbacks = run(0, -1)
    assert fallbacks == 1, "the forced give-up did not happen (or was not noticed)"
    assert torch.isfinite(got.loglikelihood).all()
    assert torch.equal(got.loglikelihood, ref.loglikelihood)
    assert torch.equal(got.filter_means, ref.filter_means)
    assert torch.equal(got.latest_state.timeseries_state.value, ref.latest_state.timeseries_state.value)
    assert torch.equal(got.latest_state.previous_indices, ref.latest_state.previous_indices)
    ok, fallbacks = run(0, 0)  # ... and with the default patience the same filter takes the cluster kernel and nothing gives up
    assert fallbacks == 0 and torch.isfinite(ok.loglikelihood).all()
    torch.testing.assert_close(ok.loglikelihood, ref.loglikelihood, rtol=2e-3, atol=2e-2)  # (float32: another summation order)


def test_smc2_survives_cluster_launches_that_give_up(monkeypatch):
    """SMC2.fit (pipelined blocks: the status word travels with the block's statistics) and SMC2.step (watched online moves: the
    word rides through ``pf_theta_step`` into the host slot) with ``cluster_patience = -1``: every cluster launch gives up, every
    piece is re-issued on the per-step route - the run equals the one whose cluster launches all completed, decision for decision.  (float64: a watched move's log-likelihood joins the running total in
    ``pf_theta_step`` - the reference's ``+=`` of the increment in the tensors' type, filters/result.py:130 - where an unwatched
    move's kernel adds its double-precision increment before rounding: one float32 ulp apart, the same number in float64.)"""
    import warnings

    from pyfilter_amd import resampling, timeseries as ts
    from pyfilter_amd.filters.particle import APF, proposals
    from pyfilter_amd.hints import HINTS
    from pyfilter_amd.inference import SMC2
    from pyfilter_amd.timeseries import models
    from torch.distributions import Exponential, LogNormal, Normal

    g = torch.Generator().manual_seed(3)
    y = (0.05 * torch.randn(40, generator=g, dtype=torch.float64)).cumsum(0).to(DEV)
    priors = {"kappa": Exponential(10.0), "gamma": Normal(0.0, 1.0), "sigma": LogNormal(-2.0, 1.0)}
    obs_a, obs_s = torch.tensor(1.0, device=DEV, dtype=torch.float64), torch.tensor(0.05, device=DEV, dtype=torch.float64)

    def build(theta):
        return ts.LinearStateSpaceModel(models.OrnsteinUhlenbeck(theta["kappa"], theta["gamma"], theta["sigma"], dt=1.0), (obs_a, obs_s))

    def fit(cluster, patience, how):
        monkeypatch.setattr(HINTS, "cluster", cluster)
        monkeypatch.setattr(HINTS, "cluster_patience", patience)
        filt = APF(build, 4096, proposal=proposals.Bootstrap(), resampling=resampling.systematic, seed=5)
        alg = SMC2(filt, 16, priors, threshold=0.5, device=torch.device(DEV), dtype=torch.float64, seed=9)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            if how == "fit":
                state = alg.fit(y, block=8)
            else:
                state = alg.initialize()
                for t in range(y.shape[0]):
                    state = alg.step(y[t], state)
        torch.cuda.synchronize()
        return state, getattr(filt, "cluster_fallbacks", 0)

    for how in ("fit", "step"):
        ref, fb0 = fit(True, 0, how)    # every cluster launch completes
        got, fb1 = fit(True, -1, how)   # every cluster launch gives up: every piece re-issued on the per-step route
        assert fb0 == 0 and fb1 > 0, (how, fb0, fb1)
        # the same draws either way (the routes key Philox alike; a repeated PMMH move rewinds the theta-level stream): the two runs
        # are the same particle system up to the routes' float64 summation order - same decisions, numbers to 1e-8
        assert len(got.ess) == len(ref.ess), how
        torch.testing.assert_close(torch.stack(got.ess), torch.stack(ref.ess), rtol=1e-8, atol=1e-8, msg=how)
        torch.testing.assert_close(got.w, ref.w, rtol=1e-8, atol=1e-8, msg=how)
        torch.testing.assert_close(got.filter_state.loglikelihood, ref.filter_state.loglikelihood, rtol=1e-8, atol=1e-8, msg=how)


def test_four_streams_of_cluster_launches_from_one_thread():
    """Four HIP streams, one host thread: twelve cluster launches (filters of different sizes, each launch sized to the chip's resident
    slots) issued round robin without waiting - whatever the hardware interleaves, the grouped workgroup ids keep whole filters
    resident: every run finite, equal to the same filter run alone, nothing gives up."""
    from pyfilter_amd.hints import HINTS

    assert HINTS.kernel_route() == 3
    g = torch.Generator().manual_seed(29)
    y = (0.1 * torch.randn(60, generator=g)).cumsum(0).to(DEV)
    shapes = [(128, 8192), (200, 4096), (64, 16384), (256, 4100)]
    alone = []
    for i, (b, n) in enumerate(shapes):
        f = _concurrent_filter(70 + i, b=b, n=n)
        alone.append(f.batch_filter(y, bar=False).loglikelihood.cpu())
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in shapes]
    filters = [_concurrent_filter(70 + i, b=b, n=n) for i, (b, n) in enumerate(shapes)]
    outs = [[] for _ in shapes]
    for rep in range(3):
        for i, st in enumerate(streams):
            with torch.cuda.stream(st):
                filters[i]._defer_status_once = True  # (no wait between the launches: the status words are read at the end)
                outs[i].append(filters[i].batch_filter(y, bar=False))
    torch.cuda.synchronize()
    for i in range(len(shapes)):
        assert getattr(filters[i], "cluster_fallbacks", 0) == 0
        for r in outs[i]:
            watch = getattr(r, "_cluster_watch", None)
            assert watch is not None and int(watch[0].item()) == 0, "a launch gave up"
            assert torch.isfinite(r.loglikelihood).all()
        assert torch.equal(outs[i][0].loglikelihood.cpu(), alone[i]), shapes[i]


def _process_worker(rank, barrier, out_dir, seed):
    import os

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    g = torch.Generator().manual_seed(17)
    y = (0.1 * torch.randn(200, generator=g)).cumsum(0).to(DEV)
    f = _concurrent_filter(seed + rank)
    f.batch_filter(y[:4], bar=False)  # (load the library, build the plan)
    torch.cuda.synchronize()
    f = _concurrent_filter(seed + rank)
    barrier.wait()
    reps = [f.batch_filter(y, bar=False).loglikelihood.cpu() for _ in range(3)]
    torch.cuda.synchronize()
    torch.save({"ll": reps, "fallbacks": getattr(f, "cluster_fallbacks", 0)}, os.path.join(out_dir, f"r{rank}.pt"))


def test_two_processes_share_the_gpu_on_the_cluster_route(tmp_path):
    """Two PROCESSES on the one GPU, each filtering 128 x 8 192 x T = 200 on the default hints at the same time - two tenants whose
    cluster launches each fill the chip's resident slots and know nothing of each other (the residency query cannot): both finish,
    finite, equal to the same filters run alone, without a fallback."""
    import torch.multiprocessing as mp

    g = torch.Generator().manual_seed(17)
    y = (0.1 * torch.randn(200, generator=g)).cumsum(0).to(DEV)
    alone = [_concurrent_filter(90 + r).batch_filter(y, bar=False).loglikelihood.cpu() for r in range(2)]
    torch.cuda.synchronize()
    ctx = mp.get_context("spawn")
    barrier = ctx.Barrier(2)
    procs = [ctx.Process(target=_process_worker, args=(r, barrier, str(tmp_path), 90)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0, f"worker exited with {p.exitcode}"
    for r in range(2):
        got = torch.load(str(tmp_path / f"r{r}.pt"))
        assert got["fallbacks"] == 0
        for ll in got["ll"]:
            assert torch.isfinite(ll).all()
        assert torch.equal(got["ll"][0], alone[r])
