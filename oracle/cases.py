"""
TEST INFRASTRUCTURE — the golden-fixture case table shared by ``oracle/make_golden.py`` (generator, this container
only) and ``tests/`` (consumers; they only need the spec, the inputs come from the ``.npz``).

Model constants follow SURVEY.md §8(c)/(d): AR(1) of tests/filters/models.py:10-15, the README sine diffusion,
the Verhulst SV model of examples/stochastic-volatility.ipynb, Lorenz-63 of examples/lorenz.ipynb and the OU model
of tests/inference/models.py:12-19, and the reference's own 2-D acceptance model - the random walk with sigma = (0.05, 0.1),
A = I2, s = 0.15 of tests/filters/models.py:28-52 (``rw2d``; B in {1, 3} and 10 % NaN rows as tests/filters/test_particle.py:44-63
runs it).
"""
import math

import torch

from . import models as M

_LORENZ_INIT = ([-5.91652, -5.52332, 24.5723], [math.sqrt(10.0)] * 3)
_LORENZ_A = [[0.8, 0.0, 0.0], [0.0, 0.0, 0.8]]

CASES = [
    # name, model, filter, proposal, N, B, T
    dict(name="lg1d_sisr_boot", model="lg1d", filter="sisr", proposal="bootstrap", N=1000, B=1, T=25,
         ess_threshold=0.9, seed=101, dtypes=("f64", "f32")),
    dict(name="lg1d_apf_lgo", model="lg1d", filter="apf", proposal="lgo", N=256, B=3, T=20,
         ess_threshold=0.9, seed=102, dtypes=("f64", "f32")),
    dict(name="sine_apf_lgo", model="sine", filter="apf", proposal="lgo", N=512, B=2, T=20,
         ess_threshold=0.9, seed=103, dtypes=("f64", "f32")),
    dict(name="sine_sisr_lgo", model="sine", filter="sisr", proposal="lgo", N=256, B=3, T=20,
         ess_threshold=0.5, seed=104, dtypes=("f64",)),
    dict(name="sine_apf_boot_nan", model="sine", filter="apf", proposal="bootstrap", N=256, B=2, T=20,
         ess_threshold=0.9, seed=105, nan_steps=(3, 4, 11), dtypes=("f64",)),
    dict(name="sine_sisr_boot_nan", model="sine", filter="sisr", proposal="bootstrap", N=256, B=3, T=20,
         ess_threshold=0.7, seed=106, nan_steps=(2, 9), dtypes=("f64",)),
    dict(name="sv_apf_boot", model="sv_batched", filter="apf", proposal="bootstrap", N=256, B=4, T=20,
         ess_threshold=0.9, seed=107, dtypes=("f64", "f32")),
    dict(name="sv_sisr_boot", model="sv_batched", filter="sisr", proposal="bootstrap", N=256, B=4, T=20,
         ess_threshold=0.6, seed=108, dtypes=("f64",)),
    dict(name="lorenz_sisr_boot", model="lorenz", filter="sisr", proposal="bootstrap", N=256, B=2, T=15,
         ess_threshold=0.9, seed=109, dtypes=("f64", "f32")),
    dict(name="lorenz_apf_lgo", model="lorenz", filter="apf", proposal="lgo", N=128, B=2, T=15,
         ess_threshold=0.9, seed=110, dtypes=("f64",)),
    dict(name="ou_sisr_lgo_theta", model="ou_batched", filter="sisr", proposal="lgo", N=128, B=3, T=20,
         ess_threshold=0.8, seed=111, dtypes=("f64",)),
    dict(name="ou_apf_boot_theta", model="ou_batched", filter="apf", proposal="bootstrap", N=128, B=3, T=20,
         ess_threshold=0.9, seed=112, dtypes=("f64",)),
    # D = 2 / O = 2: the reference's 2-D random walk (tests/filters/models.py:28-52), its MVN / Cholesky LGO path included
    dict(name="rw2d_sisr_boot", model="rw2d", filter="sisr", proposal="bootstrap", N=512, B=3, T=25,
         ess_threshold=0.9, seed=113, dtypes=("f64", "f32")),
    dict(name="rw2d_apf_lgo", model="rw2d", filter="apf", proposal="lgo", N=256, B=3, T=25,
         ess_threshold=0.9, seed=114, nan_steps=(4, 13, 14), dtypes=("f64", "f32")),
    dict(name="rw2d_sisr_lgo", model="rw2d", filter="sisr", proposal="lgo", N=300, B=1, T=20,
         ess_threshold=0.5, seed=115, dtypes=("f64",)),
    dict(name="rw2d_apf_boot", model="rw2d", filter="apf", proposal="bootstrap", N=333, B=2, T=20,
         ess_threshold=0.9, seed=116, nan_steps=(7,), dtypes=("f64",)),
]

CASE_BY_NAME = {c["name"]: c for c in CASES}


def build_spec(case, dtype=torch.float64) -> M.ModelSpec:
    m, b = case["model"], case["B"]
    t = lambda v: torch.tensor(v, dtype=dtype)  # noqa: E731

    if m == "lg1d":  # tests/filters/models.py:10-15
        return M.ModelSpec(M.HID_LINEAR, (0.0, 0.99, 0.05), 0, 1.0, (0.0, 0.05), M.OBS_LINEAR, (1.0, 0.0, 0.15), 0)
    if m == "sine":  # README.md:44-67
        return M.ModelSpec(M.HID_SINE_EM, (0.0, 1.0), 0, 0.1, (0.0, 1.0), M.OBS_LINEAR, (1.0, 0.0, 0.1), 0)
    if m == "sv_batched":  # stochastic-volatility.ipynb, B distinct parameter rows + B distinct series
        k = case.get("param_step_scale", 1.0)  # spacing of the B parameter rows (1: the golden fixtures)
        kappa = t([0.05 + 0.01 * k * i for i in range(b)])
        gamma = t([1.0 + 0.1 * k * i for i in range(b)])
        sigma = t([0.10 + 0.02 * k * i for i in range(b)])
        mu = t([0.0 + 0.05 * k * i for i in range(b)])
        return M.ModelSpec(M.HID_VERHULST_EM, (kappa, gamma, sigma), 0, 0.2, (1.0, 0.1), M.OBS_SV, (mu,), 0)
    if m == "lorenz":  # lorenz.ipynb
        return M.ModelSpec(
            M.HID_LORENZ63_EM, (10.0, 28.0, 8.0 / 3.0, 1.0), 3, 0.01, (t(_LORENZ_INIT[0]), t(_LORENZ_INIT[1])),
            M.OBS_LINEAR, (t(_LORENZ_A), t([0.0]), t([math.sqrt(0.1)])), 2,
        )
    if m == "rw2d":  # tests/filters/models.py:28-52: x' = I2 x + (0.05, 0.1) e, x0 ~ N(0, sigma), y ~ N(I2 x, 0.15)
        sig = t([0.05, 0.1])
        return M.ModelSpec(
            M.HID_LINEAR, (torch.zeros_like(sig), torch.ones_like(sig), sig), 2, 1.0, (torch.zeros_like(sig), sig),
            M.OBS_LINEAR, (torch.eye(2, dtype=dtype), t([0.0, 0.0]), t([0.15, 0.15])), 2,
        )
    if m == "ou_batched":  # tests/inference/models.py:12-33 with theta on the batch dim
        kappa = t([0.025 * (i + 1) for i in range(b)])
        gamma = t([0.0 + 0.1 * i for i in range(b)])
        sigma = t([0.05 + 0.01 * i for i in range(b)])
        return M.ModelSpec(M.HID_OU, (kappa, gamma, sigma), 0, 1.0, (0.0, 0.1), M.OBS_LINEAR, (1.0, 0.0, 0.05), 0)
    raise KeyError(m)


def simulate(case, spec: M.ModelSpec, dtype=torch.float64) -> torch.Tensor:
    """Synthetic observations from *our own* simulator of the same model (first batch column's parameters when the
    model is theta-batched, except the SV case which simulates B distinct series -> ``y (T,B)``)."""
    g = torch.Generator().manual_seed(case["seed"] + 7)
    t_len = case["T"]
    b = case["B"]
    per_series = case["model"] == "sv_batched"
    shape = (1, b) if (per_series or case["model"] == "ou_batched") else (1, 1)
    if spec.dim > 0:
        shape = shape + (spec.dim,)
    x = M.initial_sample(spec, torch.randn(shape, generator=g, dtype=dtype))
    ys = []
    for _ in range(t_len):
        x = M.propagate(spec, x, torch.randn(shape, generator=g, dtype=dtype))
        loc, scale = M.obs_loc_scale(spec, x)
        scale = M._t(scale, loc)
        yv = loc + scale * torch.randn(loc.shape, generator=g, dtype=dtype)
        ys.append(yv[0] if per_series else yv[0, 0])
    y = torch.stack(ys)
    for s in case.get("nan_steps", ()):
        y[s] = float("nan")
    return y
