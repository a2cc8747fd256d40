"""
TEST INFRASTRUCTURE — CPU restatement of the *model arithmetic* on the SISR/APF hot path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module.

The reference keeps its model layer in the third-party package ``stochproc`` (pinned v0.3.0,
``/root/reference/pyproject.toml:34``) whose source is absent from ``/root/reference`` -> **parity unpinned at
that boundary** (SURVEY.md §8(c)).  The closed-form definitions below are restated from the reference's own
README / notebooks / tests:

* ``HID_LINEAR``     : ``x' = alpha + beta*x + sigma*eps``          tests/filters/models.py:10-15 (AR), :29-38 (RW)
* ``HID_SINE_EM``    : ``x' = x + sin(x-gamma)*dt + sigma*eps``     README.md:44-62,  eps ~ N(0, sqrt(dt))
* ``HID_VERHULST_EM``: ``v' = v + kappa*(gamma-v)*v*dt + sigma*v*eps`` examples/stochastic-volatility.ipynb (ts.models.Verhulst)
* ``HID_LORENZ63_EM``: Lorenz-63 drift, Euler-Maruyama                examples/lorenz.ipynb (``f``), eps ~ N(0, sqrt(dt)) iid per dim
* ``HID_OU``         : exact OU discretisation                        tests/inference/models.py:12-19
* ``OBS_LINEAR``     : ``y ~ N(b + A x, s)``                          LinearStateSpaceModel, proposals/linear.py:48
* ``OBS_SV``         : ``y ~ N(mu, scale=x)``                         stochastic-volatility.ipynb ``build_obs`` at skew=0, kurt=1

Every function works on plain tensors laid out like the reference's: particles on dim 0, optional batch on dim 1,
optional state dim last (``pyfilter/filters/particle/base.py:51-62``).  Parameters are python floats, 0-d tensors
or ``(B,)`` tensors (one value per parallel filter, as SMC2 uses them: ``inference/sequential/base.py:31-34``).
"""
import math

import torch

HID_LINEAR, HID_SINE_EM, HID_VERHULST_EM, HID_LORENZ63_EM, HID_OU = 0, 1, 2, 3, 4
OBS_LINEAR, OBS_SV = 0, 1

HIDDEN_NAMES = {
    HID_LINEAR: "linear",
    HID_SINE_EM: "sine_em",
    HID_VERHULST_EM: "verhulst_em",
    HID_LORENZ63_EM: "lorenz63_em",
    HID_OU: "ou",
}


def _t(p, like):
    return p if isinstance(p, torch.Tensor) else torch.as_tensor(p, dtype=like.dtype)


class ModelSpec:
    """
    Closed description of one state-space model: the ``kernel_id`` view of a stochproc ``StateSpaceModel``.

    Args:
        hidden: one of ``HID_*``.
        hidden_params: tuple of parameters (see table above).  ``HID_LINEAR``: (alpha, beta, sigma) each scalar,
            ``(B,)`` or - for D>0 - ``(D,)``;  ``HID_SINE_EM``: (gamma, sigma); ``HID_VERHULST_EM``: (kappa, gamma,
            sigma); ``HID_LORENZ63_EM``: (s, r, b, sigma); ``HID_OU``: (kappa, gamma, sigma).
        dim: 0 for a scalar state (no trailing dim), else D.
        dt: step for the Euler-Maruyama kinds / OU (ignored by ``HID_LINEAR``).
        init: (mean, scale) of the Gaussian initial distribution; scalars or ``(D,)``.
        obs: ``OBS_LINEAR`` or ``OBS_SV``.
        obs_params: ``OBS_LINEAR``: (a, b, s) with ``a`` scalar (dim 0) or ``(O, D)``; ``OBS_SV``: (mu,).
        obs_dim: 0 for scalar observations, else O.
    """

    def __init__(self, hidden, hidden_params, dim, dt, init, obs, obs_params, obs_dim, observe_every_step=1):
        self.hidden = hidden
        self.hidden_params = tuple(hidden_params)
        self.dim = dim
        self.dt = dt
        self.init = init
        self.obs = obs
        self.obs_params = tuple(obs_params)
        self.obs_dim = obs_dim
        self.observe_every_step = observe_every_step

    @property
    def inc_scale(self):
        """Scale of the increment distribution: N(0,1) for the discrete kinds, N(0, sqrt(dt)) for Euler-Maruyama."""
        return math.sqrt(self.dt) if self.hidden in (HID_SINE_EM, HID_VERHULST_EM, HID_LORENZ63_EM) else 1.0


def mean_scale(spec: ModelSpec, x: torch.Tensor):
    """``hidden.mean_scale(x) -> (loc, scale)`` broadcast to ``x.shape`` (row M; call sites proposals/linear.py:41,
    pre_weight_funcs.py:10)."""
    p = [_t(q, x) for q in spec.hidden_params]
    k = spec.hidden
    dt = spec.dt

    if k == HID_LINEAR:
        alpha, beta, sigma = p
        loc, scale = alpha + beta * x, sigma
    elif k == HID_SINE_EM:
        gamma, sigma = p
        loc, scale = x + torch.sin(x - gamma) * dt, sigma
    elif k == HID_VERHULST_EM:
        kappa, gamma, sigma = p
        loc, scale = x + kappa * (gamma - x) * x * dt, sigma * x
    elif k == HID_LORENZ63_EM:
        s, r, b, sigma = p
        x0, x1, x2 = x[..., 0], x[..., 1], x[..., 2]
        f = torch.stack((-s * (x0 - x1), r * x0 - x1 - x0 * x2, x0 * x1 - b * x2), dim=-1)
        loc, scale = x + f * dt, sigma
    elif k == HID_OU:
        kappa, gamma, sigma = p
        e = torch.exp(-kappa * dt)
        loc = gamma + (x - gamma) * e
        scale = sigma * torch.sqrt((1.0 - torch.exp(-2.0 * kappa * dt)) / (2.0 * kappa))
    else:
        raise NotImplementedError(k)

    loc, scale = torch.broadcast_tensors(loc, _t(scale, x))
    return loc, scale


def propagate(spec: ModelSpec, x: torch.Tensor, z: torch.Tensor):
    """``hidden.propagate(x)`` with the standard-normal draws ``z`` supplied (tape).  The reference samples
    ``TransformedDistribution(inc, AffineTransform(loc, scale))`` = ``loc + scale * (z * inc_scale)``."""
    loc, scale = mean_scale(spec, x)
    eps = z * spec.inc_scale
    return loc + scale * eps


def transition_log_prob(spec: ModelSpec, x_new, loc, scale):
    """``hidden.build_density(x).log_prob(x_new)`` = Normal(0, inc_scale).log_prob((x_new-loc)/scale) - log|scale|,
    summed over the state dim when D>0 (AffineTransform with event_dim=n_dim)."""
    inc = spec.inc_scale
    eps = (x_new - loc) / scale
    lp = -(eps ** 2) / (2.0 * inc * inc) - math.log(inc) - math.log(math.sqrt(2.0 * math.pi))
    lp = lp - scale.abs().log()
    return lp.sum(-1) if spec.dim > 0 else lp


def obs_loc_scale(spec: ModelSpec, x: torch.Tensor):
    """Location/scale of ``model.build_density(x)`` (LinearStateSpaceModel: ``Normal(b + A x, s)``)."""
    if spec.obs == OBS_LINEAR:
        a, b, s = [_t(q, x) for q in spec.obs_params]
        if spec.dim == 0:
            loc = b + a * (x.unsqueeze(-1) if spec.obs_dim > 0 else x)  # (a scalar state under a vector observation: a of shape (O,))
        else:
            loc = b + (a @ x.unsqueeze(-1)).squeeze(-1)
        return loc, s
    if spec.obs == OBS_SV:
        (mu,) = [_t(q, x) for q in spec.obs_params]
        return mu + torch.zeros_like(x), x
    raise NotImplementedError(spec.obs)


def normal_log_prob(y, loc, scale):
    """torch.distributions.Normal.log_prob restated."""
    var = scale ** 2
    return -((y - loc) ** 2) / (2 * var) - scale.log() - math.log(math.sqrt(2 * math.pi))


def obs_log_prob(spec: ModelSpec, y: torch.Tensor, x: torch.Tensor):
    """``model.build_density(x).log_prob(y)`` -> ``(N,[B])``."""
    loc, scale = obs_loc_scale(spec, x)
    lp = normal_log_prob(y, loc, scale)
    return lp.sum(-1) if spec.obs_dim > 0 else lp


def initial_sample(spec: ModelSpec, z0: torch.Tensor):
    """``hidden.initial_sample((N,*B))`` with the standard normal draws supplied."""
    m, s = spec.init
    return _t(m, z0) + _t(s, z0) * z0
