"""Shared test helpers: golden-fixture loading and construction of ``pyfilter_amd`` filters from the case table."""
import math
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DT = {"f64": torch.float64, "f32": torch.float32}


def load_golden(name, dt):
    with np.load(os.path.join(GOLDEN, f"{name}_{dt}.npz")) as f:
        return {k: torch.from_numpy(f[k]) for k in f.files}


def moves_after(g, t):
    """Moves a filter has made once it has consumed ``t`` observations of fixture ``g`` (= the time index of that state):
    ``t`` unless the model is observed every k-th step only (``move_of_obs``, oracle/make_golden.py)."""
    if t == 0 or "move_of_obs" not in g:
        return t
    return int(g["move_of_obs"][t - 1]) + 1


def build_ssm_from_case(case, dtype, device):
    from pyfilter_amd import timeseries as ts
    from pyfilter_amd.timeseries import models

    b = case["B"]
    t = lambda v: torch.tensor(v, dtype=dtype, device=device)  # noqa: E731
    m = case["model"]
    if m == "lg1d":
        hidden = models.AR(t(0.0), t(0.99), t(0.05), initial=(t(0.0), t(0.05)))
        ssm = ts.LinearStateSpaceModel(hidden, (t(1.0), t(0.15)))
    elif m == "lg1d_o2":  # a scalar state under a vector observation: the torch route's linear-Gaussian shape (oracle/cases.py)
        hidden = models.AR(t(0.0), t(0.99), t(0.05), initial=(t(0.0), t(0.05)))
        ssm = ts.LinearStateSpaceModel(hidden, (t([1.0, 0.5]), t([0.0, 0.1]), t([0.15, 0.2])), torch.Size([2]))
    elif m == "sine":
        hidden = models.SineDiffusion(t(0.0), t(1.0), dt=0.1)
        ssm = ts.LinearStateSpaceModel(hidden, (t(1.0), t(0.1)))
    elif m == "sv_batched":
        k = case.get("param_step_scale", 1.0)  # spacing of the B parameter rows (1: the golden fixtures)
        kappa = t([0.05 + 0.01 * k * i for i in range(b)])
        gamma = t([1.0 + 0.1 * k * i for i in range(b)])
        sigma = t([0.10 + 0.02 * k * i for i in range(b)])
        mu = t([0.0 + 0.05 * k * i for i in range(b)])
        hidden = models.Verhulst(kappa, gamma, sigma, dt=0.2, initial=(t(1.0), t(0.1)))
        ssm = models.StochasticVolatilityModel(hidden, mu)
    elif m == "lorenz":
        hidden = models.Lorenz63(t(10.0), t(28.0), t(8.0 / 3.0), t(1.0), dt=0.01,
                                 initial_mean=t([-5.91652, -5.52332, 24.5723]), initial_scale=t([math.sqrt(10.0)] * 3))
        a = t([[0.8, 0.0, 0.0], [0.0, 0.0, 0.8]])
        ssm = ts.LinearStateSpaceModel(hidden, (a, t([0.0]), t([math.sqrt(0.1)])), torch.Size([2]))
    elif m in ("lorenz_s", "lorenz_o1", "lorenz_o3"):  # the lorenz.ipynb process under other linear observations (oracle/cases.py)
        from oracle.cases import _LORENZ_A_O3, _LORENZ_A_S

        hidden = models.Lorenz63(t(10.0), t(28.0), t(8.0 / 3.0), t(1.0), dt=0.01,
                                 initial_mean=t([-5.91652, -5.52332, 24.5723]), initial_scale=t([math.sqrt(10.0)] * 3))
        if m == "lorenz_s":
            ssm = ts.LinearStateSpaceModel(hidden, (t(_LORENZ_A_S), t(0.0), t(math.sqrt(0.1))), torch.Size([]))
        elif m == "lorenz_o1":
            ssm = ts.LinearStateSpaceModel(hidden, (t([_LORENZ_A_S]), t([0.0]), t([math.sqrt(0.1)])), torch.Size([1]))
        else:
            ssm = ts.LinearStateSpaceModel(hidden, (t(_LORENZ_A_O3), t([0.1, -0.2, 0.3]), t([0.3, 0.4, 0.5])), torch.Size([3]))
    elif m in ("rw2d_s", "rw2d_o1"):
        from oracle.cases import _RW2D_A_S

        sig = t([0.05, 0.1])
        hidden = models.RandomWalk(sig, initial_mean=t([0.0, 0.0]), initial_scale=sig, dim=2)
        if m == "rw2d_s":
            ssm = ts.LinearStateSpaceModel(hidden, (t(_RW2D_A_S), t(0.05), t(0.15)), torch.Size([]))
        else:
            ssm = ts.LinearStateSpaceModel(hidden, (t([_RW2D_A_S]), t([0.05]), t([0.15])), torch.Size([1]))
    elif m in ("rw2d_theta", "rw2d_theta_b", "rw_rand"):  # (per-filter) sigma, A, b, s - the very tensors of the oracle's spec
        from oracle.cases import build_spec

        spec = build_spec(case, dtype)
        sig = spec.hidden_params[2].to(device)
        a, off, s = (p.to(device) for p in spec.obs_params)
        hidden = models.RandomWalk(sig, initial_mean=torch.zeros_like(sig), initial_scale=sig, dim=spec.dim)
        ssm = ts.LinearStateSpaceModel(hidden, (a, off, s), torch.Size([spec.obs_dim]) if spec.obs_dim else torch.Size([]))
    elif m == "rw2d":  # the reference's own 2-D model (tests/filters/models.py:28-52)
        sig = t([0.05, 0.1])
        hidden = models.RandomWalk(sig, initial_mean=t([0.0, 0.0]), initial_scale=sig, dim=2)
        ssm = ts.LinearStateSpaceModel(hidden, (torch.eye(2, dtype=dtype, device=device), t([0.15, 0.15])), torch.Size([2]))
    elif m == "ou_batched":
        kappa = t([0.025 * (i + 1) for i in range(b)])
        gamma = t([0.0 + 0.1 * i for i in range(b)])
        sigma = t([0.05 + 0.01 * i for i in range(b)])
        hidden = models.OrnsteinUhlenbeck(kappa, gamma, sigma, dt=1.0, initial=(t(0.0), t(0.1)))
        ssm = ts.LinearStateSpaceModel(hidden, (t(1.0), t(0.05)))
    else:
        raise KeyError(m)
    ssm.observe_every_step = int(case.get("observe_every_step", 1))
    return ssm.to(device)


def build_filter_from_case(case, g, dtype, device, tape=True, **kwargs):
    from pyfilter_amd.filters.particle import APF, SISR, proposals

    ssm = build_ssm_from_case(case, dtype, device)
    prop = {"bootstrap": proposals.Bootstrap, "lgo": proposals.LinearGaussianObservations}[case["proposal"]]()
    cls = {"sisr": SISR, "apf": APF}[case["filter"]]
    filt = cls(ssm, case["N"], proposal=prop, ess_threshold=case["ess_threshold"], **kwargs)
    filt.set_batch_shape(torch.Size([case["B"]]))
    if tape:
        filt.set_tape(z=g["z_tape"].to(dtype), u=g["u_tape"].to(dtype), z0=g["z0"].to(dtype))
    return filt
