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
