"""Pins the oracle (``oracle/cpu_ref.py``) against golden vectors produced by the unmodified reference
(``oracle/make_golden.py``): per-step states, ancestors, log-likelihoods and the final FilterResult moments."""
import os

import numpy as np
import pytest
import torch

from oracle import cpu_ref
from oracle.cases import CASES, build_spec

DT = {"f64": torch.float64, "f32": torch.float32}
PARAMS = [(c["name"], d) for c in CASES for d in c["dtypes"]]


def load(golden_dir, name, dt):
    with np.load(os.path.join(golden_dir, f"{name}_{dt}.npz")) as f:
        return {k: torch.from_numpy(f[k]) for k in f.files}


@pytest.mark.parametrize("name,dt", PARAMS)
def test_filter_matches_reference(golden_dir, name, dt):
    case = next(c for c in CASES if c["name"] == name)
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
