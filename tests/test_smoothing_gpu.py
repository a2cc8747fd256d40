"""Recorded states on the fused route, smoothing and checkpoint interoperability
against the reference's golden fixtures (``oracle/make_golden.py``: per-step states, ``smooth(states, "fl")``, FFBS
statistics, ``FilterResult.state_dict()``), through ``libpfamd.so``."""
import math

import pytest
import torch

from oracle import cpu_ref
from oracle.cases import FUSED_CASES as CASES, build_spec  # (lg1d_o2_*: the torch route, tests/test_torch_route_golden.py)
from pyfilter_amd import ops
from tests.helpers import DT, build_filter_from_case, load_golden, moves_after

pytestmark = pytest.mark.gpu
F64 = [c["name"] for c in CASES]
TOL = dict(rtol=1e-9, atol=1e-11)


@pytest.mark.parametrize("name", F64)
def test_fused_run_records_every_state(name):
    """``record_states=True`` on the fused route: the kernels keep the state history (``pf_filter_args.ring``); every
    recorded state equals the reference's per-step state - particles, weights, log-likelihood and identical ancestors
    (incl. the carried ancestors of SISR steps that did not resample)."""
    case = next(c for c in CASES if c["name"] == name)
    g = load_golden(name, "f64")
    filt = build_filter_from_case(case, g, torch.float64, "cuda", record_states=True)
    res = filt.batch_filter(g["y"].cuda(), bar=False)
    thinned = case.get("observe_every_step", 1) != 1
    # the fused route, one history slot per move from the first recorded one on
    assert filt._last_run["plan"].ring == moves_after(g, g["y"].shape[0]) - moves_after(g, 1) + 1
    states = res.states
    assert len(states) == g["y"].shape[0] + 1
    torch.testing.assert_close(states[0].timeseries_state.value.cpu(), g["x0"], **TOL)
    for t, st in enumerate(states[1:]):
        assert int(st.timeseries_state.time_index) == moves_after(g, t + 1)
        assert torch.equal(st.previous_indices.cpu(), g["step_idx"][t]), f"ancestors differ at step {t}"
        torch.testing.assert_close(st.timeseries_state.value.cpu(), g["step_x"][t], **TOL)
        torch.testing.assert_close(st.weights.cpu(), g["step_w"][t], equal_nan=True, **TOL)
        torch.testing.assert_close(st.get_loglikelihood().cpu(), g["step_ll"][t], **TOL)
        torch.testing.assert_close(st.get_mean().cpu(), g["filter_means"][t + 1], **TOL)
    torch.testing.assert_close(res.filter_means.cpu(), g["filter_means"], **TOL)
    torch.testing.assert_close(res.loglikelihood.cpu(), g["loglikelihood"], **TOL)
    if thinned:  # (the recorded states skip moves: no ancestor chain to follow)
        return
    # fixed-lag smoothing: pure ancestor chasing over the recorded history
    sm = filt.smooth(states, "fl")
    assert sm.shape == g["smooth_fl"].shape
    torch.testing.assert_close(sm.cpu(), g["smooth_fl"], **TOL)


@pytest.mark.parametrize("keep", [1, 3, 7])
def test_bounded_state_history(keep):
    """``record_states=<int>`` keeps the last ``keep`` states (a ring of keep + 1 slots); a smoother over them equals the
    tail of the full-history smoother."""
    case = next(c for c in CASES if c["name"] == "sine_sisr_lgo")
    g = load_golden(case["name"], "f64")
    t_len = g["y"].shape[0]
    filt = build_filter_from_case(case, g, torch.float64, "cuda", record_states=keep)
    res = filt.batch_filter(g["y"].cuda(), bar=False)
    states = res.states
    assert len(states) == keep
    for k, st in enumerate(states):
        t = t_len - keep + k
        assert torch.equal(st.previous_indices.cpu(), g["step_idx"][t])
        torch.testing.assert_close(st.timeseries_state.value.cpu(), g["step_x"][t], **TOL)
    if keep > 1:
        xs = [g["step_x"][t] for t in range(t_len - keep, t_len)]
        inds = [g["step_idx"][t] for t in range(t_len - keep, t_len)]
        torch.testing.assert_close(filt.smooth(states, "fl").cpu(), cpu_ref.smooth_fl(xs, inds), **TOL)


def test_recorded_states_with_unobserved_substeps():
    """``observe_every_step = 3``: the fused run reports the states of the moves that consumed an observation - or every
    move with ``record_intermediary_states`` - exactly like the step-by-step driver."""
    import os

    from pyfilter_amd import timeseries as ts
    from pyfilter_amd.filters.particle import SISR, proposals
    from pyfilter_amd.timeseries import models

    t = lambda v: torch.tensor(v, dtype=torch.float64, device="cuda")  # noqa: E731
    n, b, t_obs = 512, 2, 5
    gen = torch.Generator().manual_seed(9)
    y = (0.4 * torch.randn(t_obs, generator=gen, dtype=torch.float64)).cuda()
    moves = 1 + 3 * (t_obs - 1)
    z = torch.randn((moves, n, b), generator=gen, dtype=torch.float64)
    u = torch.rand((moves, b), generator=gen, dtype=torch.float64)
    z0 = torch.randn((n, b), generator=gen, dtype=torch.float64)

    def run(inter, fused):
        ssm = ts.LinearStateSpaceModel(models.AR(t(0.0), t(0.9), t(0.3)), (t(1.0), t(0.2)), observe_every_step=3)
        f = SISR(ssm, n, proposal=proposals.Bootstrap(), ess_threshold=0.6, record_states=True, record_intermediary_states=inter)
        f.set_batch_shape(torch.Size([b]))
        f.set_tape(z=z, u=u, z0=z0)
        from pyfilter_amd.hints import HINTS

        HINTS.fused_batch = fused
        try:
            return f.batch_filter(y, bar=False)
        finally:
            HINTS.fused_batch = True

    for inter in (False, True):
        a, r = run(inter, True), run(inter, False)
        assert len(a.states) == len(r.states) == (moves + 1 if inter else t_obs + 1)
        for sa, sr in zip(a.states, r.states):
            assert int(sa.timeseries_state.time_index) == int(sr.timeseries_state.time_index)
            assert torch.equal(sa.previous_indices, sr.previous_indices)
            torch.testing.assert_close(sa.timeseries_state.value, sr.timeseries_state.value, **TOL)
            torch.testing.assert_close(sa.weights, sr.weights, **TOL)
        torch.testing.assert_close(a.filter_means, r.filter_means, **TOL)
        torch.testing.assert_close(a.loglikelihood, r.loglikelihood, **TOL)


@pytest.mark.parametrize("name", ["lg1d_sisr_boot", "sine_apf_lgo", "lorenz_sisr_boot", "sv_apf_boot", "rw2d_sisr_boot"])
def test_ffbs_matches_oracle_on_identical_uniforms_and_reference_statistics(name):
    """Backward simulation (``pf_smooth_ffbs``): (i) against the oracle's restatement of ``_do_sample_ffbs`` with the same
    uniforms - identical trajectories (float64); (ii) against the reference's own FFBS statistics (fixture)."""
    case = next(c for c in CASES if c["name"] == name)
    g = load_golden(name, "f64")
    spec = build_spec(case, torch.float64)
    n, b, t_len = case["N"], case["B"], g["y"].shape[0]
    filt = build_filter_from_case(case, g, torch.float64, "cuda", record_states=True)
    res = filt.batch_filter(g["y"].cuda(), bar=False)
    states = res.states
    gen = torch.Generator().manual_seed(5)
    u_back = torch.rand((t_len, n, b), generator=gen, dtype=torch.float64)
    # the last state's resampling uniform: the tape row behind the run's own (the filter's resampler reads the tape)
    u_all = torch.cat([g["u_tape"].double(), g["ffbs_u_last"].double().reshape(1, b)])
    filt.set_tape(z=g["z_tape"].double(), u=u_all, z0=g["z0"].double())
    filt.set_smoothing_tape(u_back)
    from pyfilter_amd import resampling

    last_w = states[-1].weights
    filt._resampler = lambda w, normalized=False: resampling.systematic(w, normalized=normalized, u=u_all[-1].cuda())
    sm = filt.smooth(states, "ffbs").cpu()

    xs = [s.timeseries_state.value.cpu() for s in states]
    ws = [s.weights.cpu() for s in states]
    W = cpu_ref.normalize(ws[-1].clone())
    start = cpu_ref.batched_gather(xs[-1], cpu_ref.systematic(W, normalized=True, u=u_all[-1].reshape(-1, 1)), 0)
    ref = cpu_ref.smooth_ffbs(spec, xs, ws, start, u_back)
    same = (sm == ref)
    frac = 1.0 - same.double().mean().item()
    assert frac <= 2e-4, f"{frac:.2e} of the backward draws differ from the oracle"
    se = (g["ffbs_var"] / n).sqrt() * math.sqrt(2.0)
    mean = sm.mean(dim=1)
    assert ((mean - g["ffbs_mean"]).abs() <= 7.0 * se + 1e-9).all(), ((mean - g["ffbs_mean"]).abs() / (se + 1e-12)).max()
    # Philox draws (no tape): a different, equally valid backward pass
    filt.set_smoothing_tape(None)
    sm2 = filt.smooth(states, "ffbs").cpu()
    assert not torch.equal(sm2, sm)
    assert ((sm2.mean(dim=1) - g["ffbs_mean"]).abs() <= 7.0 * se + 1e-9).all()


def _unflatten(g, prefix="sd"):
    out = {}
    for k, v in g.items():
        if not k.startswith(prefix + "::"):
            continue
        node = out
        parts = k.split("::")[1:]
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = v
    return out


@pytest.mark.parametrize("name,dt", [("lg1d_apf_lgo", "f64"), ("lorenz_sisr_boot", "f64"), ("sv_sisr_boot", "f64"), ("lg1d_apf_lgo", "f32")])
def test_reference_checkpoint_loads_and_continues(name, dt):
    """A ``FilterResult.state_dict()`` written by the reference after 12 observations loads into this library's result,
    the filter continues on the remaining observations and lands on the reference's final numbers; the state_dict this
    library writes at the same point has the reference's keys, shapes and dtypes."""
    from oracle.make_golden import STATE_DICT_AT as k0

    case = next(c for c in CASES if c["name"] == name)
    g = load_golden(name, dt)
    dtype = DT[dt]
    sd = _unflatten(g)
    assert set(sd) == {"tensor_tuples", "state", "log_likelihood"}
    to_dev = lambda d: {k: (to_dev(v) if isinstance(v, dict) else v.cuda()) for k, v in d.items()}  # noqa: E731
    sd = to_dev(sd)

    filt = build_filter_from_case(case, g, dtype, "cuda")
    result = filt.initialize_with_result()
    result.load_state_dict({k: (dict(v) if isinstance(v, dict) else v) for k, v in sd.items()})
    assert result.filter_means.shape[0] == k0 + 1
    state = result.latest_state
    assert int(state.timeseries_state.time_index) == k0
    tol = TOL if dt == "f64" else dict(rtol=2e-4, atol=2e-5)
    torch.testing.assert_close(state.timeseries_state.value.cpu(), g["step_x"][k0 - 1], rtol=0, atol=0)
    cont = filt.batch_filter(g["y"][k0:].cuda(), bar=False, init_state=state)
    last = cont.latest_state
    if dt == "f64":
        assert torch.equal(last.previous_indices.cpu(), g["step_idx"][-1])
        torch.testing.assert_close(last.timeseries_state.value.cpu(), g["step_x"][-1], **tol)
        torch.testing.assert_close(last.weights.cpu(), g["step_w"][-1], equal_nan=True, **tol)
        torch.testing.assert_close(cont.filter_means.cpu(), g["filter_means"][k0:], **tol)
    else:
        se = (g["filter_variance"][k0:] / case["N"]).sqrt()
        assert ((cont.filter_means.cpu() - g["filter_means"][k0:]).abs() <= 6.0 * se + 1e-5).all()

    # the reverse direction: what this library writes after the same 12 observations
    mine = build_filter_from_case(case, g, dtype, "cuda").batch_filter(g["y"][:k0].cuda(), bar=False).state_dict()

    def same_layout(a, b, path=""):
        assert list(a.keys()) == list(b.keys()), (path, list(a.keys()), list(b.keys()))
        for k in a:
            if isinstance(a[k], dict):
                same_layout(a[k], b[k], path + "/" + k)
            else:
                ta, tb = torch.as_tensor(a[k]), torch.as_tensor(b[k])
                assert ta.shape == tb.shape and ta.dtype == tb.dtype, (path + "/" + k, ta.shape, tb.shape, ta.dtype, tb.dtype)

    same_layout(mine, sd)
    if dt == "f64":
        torch.testing.assert_close(mine["tensor_tuples"]["tensor_deque_None__filter_means"].cpu(),
                                   sd["tensor_tuples"]["tensor_deque_None__filter_means"].cpu(), **TOL)
        torch.testing.assert_close(mine["state"]["_x"]["value"].cpu(), sd["state"]["_x"]["value"].cpu(), **TOL)
        assert torch.equal(mine["state"]["_prev_inds"].cpu(), sd["state"]["_prev_inds"].cpu())
