"""Pins the oracle (``oracle/cpu_ref.py``) against golden vectors produced by the unmodified reference
(``oracle/make_golden.py``): per-step states, ancestors, log-likelihoods and the final FilterResult moments."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import cpu_ref
from oracle.cases import CASE_BY_NAME, CASES, CLUSTER_CASES, build_spec

DT = {"f64": torch.float64, "f32": torch.float32}
PARAMS = [(c["name"], d) for c in CASES for d in c["dtypes"]]
# (+ the reference's runs at column-cluster sizes, round 6: filtering arrays only)
FILTER_PARAMS = PARAMS + [(c["name"], d) for c in CLUSTER_CASES for d in c["dtypes"]]


def load(golden_dir, name, dt):
    with np.load(os.path.join(golden_dir, f"{name}_{dt}.npz")) as f:
        return {k: torch.from_numpy(f[k]) for k in f.files}


@pytest.mark.parametrize("name,dt", FILTER_PARAMS)
def test_filter_matches_reference(golden_dir, name, dt):
    case = CASE_BY_NAME[name]
    dtype = DT[dt]
    g = load(golden_dir, name, dt)
    spec = build_spec(case, dtype)

    out = cpu_ref.batch_filter(
        spec, case["filter"], case["proposal"], g["y"], g["x0"], g["z_tape"].to(dtype), g["u_tape"].to(dtype),
        ess_threshold=case["ess_threshold"], record_steps=True,
    )

    # ancestors: exact
    assert torch.equal(out["step_idx"], g["step_idx"]), "ancestor indices differ from the reference"
    tol = dict(rtol=1e-12, atol=1e-13) if dt == "f64" else dict(rtol=2e-5, atol=2e-6)
    for k in ("step_x", "step_w", "step_ll"):
        torch.testing.assert_close(out[k], g[k], equal_nan=True, **tol)
    torch.testing.assert_close(out["filter_means"], g["filter_means"], **tol)
    torch.testing.assert_close(out["filter_variance"], g["filter_variance"], **tol)
    torch.testing.assert_close(out["loglikelihood"], g["loglikelihood"], **tol)


@pytest.mark.parametrize("dt", ["f64", "f32"])
def test_primitives_match_reference(golden_dir, dt):
    g = load(golden_dir, "primitives", dt)
    # the reference's own known-answer arrangement (tests/test_resampling.py:31-47)
    idx = cpu_ref.systematic(g["ka_w"].moveaxis(0, 1), u=g["ka_u"], normalized=True).moveaxis(0, 1)
    assert torch.equal(idx, g["ka_idx"])

    for nm in "abc":
        lw = g[f"norm_{nm}_in"].clone()
        W = cpu_ref.normalize(lw)
        assert torch.equal(lw, g[f"norm_{nm}_inplace"])  # in-place nan_to_num semantics
        assert torch.equal(W, g[f"norm_{nm}_W"])
        assert torch.equal(cpu_ref.get_ess(W, normalized=True), g[f"norm_{nm}_ess"])
        assert torch.equal(cpu_ref.systematic(W, normalized=True, u=g[f"norm_{nm}_u"]), g[f"norm_{nm}_idx"])
        v = g[f"norm_{nm}_v"]
        torch.testing.assert_close(cpu_ref.log_likelihood(v, W), g[f"norm_{nm}_ll_w"], rtol=0, atol=0, equal_nan=True)
        torch.testing.assert_close(cpu_ref.log_likelihood(v), g[f"norm_{nm}_ll"], rtol=0, atol=0)


def _recorded(g, case, dtype):
    """The reference's T + 1 recorded states from a fixture: (xs, ws, prev_inds) lists."""
    n, b = case["N"], case["B"]
    xs = [g["x0"]] + list(g["step_x"].unbind(0))
    ws = [torch.zeros(n, b, dtype=dtype)] + list(g["step_w"].unbind(0))
    idx0 = torch.arange(n).unsqueeze(-1).expand(n, b)
    return xs, ws, [idx0] + list(g["step_idx"].unbind(0))


@pytest.mark.parametrize("name,dt", PARAMS)
def test_fixed_lag_smoothing_matches_reference(golden_dir, name, dt):
    """``smooth(states, "fl")`` of the reference (particle/base.py:136-152) is pure index chasing: exact."""
    case = next(c for c in CASES if c["name"] == name)
    if case.get("observe_every_step", 1) != 1:
        pytest.skip("recorded states of a thinned run skip moves: no ancestor chain to follow")
    g = load(golden_dir, name, dt)
    xs, _, inds = _recorded(g, case, DT[dt])
    assert torch.equal(cpu_ref.smooth_fl(xs, inds), g["smooth_fl"])


@pytest.mark.parametrize("name", ["lg1d_sisr_boot", "sine_apf_lgo", "lorenz_sisr_boot", "sv_apf_boot", "rw2d_sisr_boot"])
def test_ffbs_oracle_against_reference_statistics(golden_dir, name):
    """``smooth(states, "ffbs")``: the reference's ``Categorical`` draws cannot be injected, so the oracle (inverse CDF on
    the same logits) is pinned statistically - the per-time mean over trajectories of 4 independent backward passes of
    the reference (fixture) against the oracle's, within Monte-Carlo error."""
    case = next(c for c in CASES if c["name"] == name)
    g = load(golden_dir, name, "f64")
    spec = build_spec(case, torch.float64)
    xs, ws, _ = _recorded(g, case, torch.float64)
    n, b = case["N"], case["B"]
    gen = torch.Generator().manual_seed(11)
    W = cpu_ref.normalize(ws[-1].clone())
    start_idx = cpu_ref.systematic(W, normalized=True, u=g["ffbs_u_last"].double().reshape(-1, 1))
    start = cpu_ref.batched_gather(xs[-1], start_idx, 0)
    draws = [cpu_ref.smooth_ffbs(spec, xs, ws, start, torch.rand((len(xs) - 1, n, b), generator=gen, dtype=torch.float64))
             for _ in range(4)]
    traj = torch.stack(draws)
    mean = traj.mean(dim=(0, 2))
    # both sides average 4 N trajectories that share ancestors heavily: allow 6 standard errors of N effective draws
    se = (g["ffbs_var"] / n).sqrt() * math.sqrt(2.0)
    assert ((mean - g["ffbs_mean"]).abs() <= 6.0 * se + 1e-9).all(), ((mean - g["ffbs_mean"]).abs() / (se + 1e-12)).max()
    # the last state is the resampled one on both sides, given the same uniform: exact
    torch.testing.assert_close(traj[0, -1].mean(dim=0), traj[1, -1].mean(dim=0), rtol=0, atol=0)
