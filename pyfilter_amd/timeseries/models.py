"""
Built-in model kinds: processes whose per-particle arithmetic is implemented in ``csrc/pf_models.hpp`` so the whole
SISR/APF step runs in one fused HIP kernel.  Each class is *also* a regular :class:`AffineProcess` (its
``mean_scale`` is available as PyTorch ops), so a built-in process can be combined with a user-defined observation
density on the generic path.

Closed forms restated from the reference's README / notebooks / tests (the reference's own model layer, stochproc
v0.3.0, is not part of its source tree - SURVEY.md §8(c) "parity unpinned"):

=====================  ==========================================================  =============================
class                  transition                                                  source of the definition
=====================  ==========================================================  =============================
``AR``                 ``x' = alpha + beta x + sigma e``                           tests/filters/models.py:10-15
``RandomWalk``         ``x' = x + sigma e``  (D-dimensional, diagonal)             tests/filters/models.py:29-38
``SineDiffusion``      ``x' = x + sin(x - gamma) dt + sigma e``, e~N(0, sqrt dt)   README.md:44-62
``Verhulst``           ``v' = v + kappa (gamma - v) v dt + sigma v e``             examples/stochastic-volatility.ipynb
``Lorenz63``           Lorenz-63 drift, Euler-Maruyama, D = 3                      examples/lorenz.ipynb
``OrnsteinUhlenbeck``  exact discretisation                                        tests/inference/models.py:12-19
=====================  ==========================================================  =============================
"""
from math import sqrt
from typing import Optional, Sequence, Tuple

import torch
from torch.distributions import Independent, Normal

from .. import _lib as L
from . import AffineProcess, KernelKind, StateSpaceModel, TimeseriesState, _as_tensor


def _normal_init(mean, scale, dim):
    def kernel(*_):
        m, s = torch.broadcast_tensors(_as_tensor(mean), _as_tensor(scale))
        d = Normal(m, s, validate_args=False)
        return Independent(d, 1) if dim > 0 else d

    return kernel


def _increment(scale: float, dim: int, device=None):
    d = Normal(torch.tensor(0.0, device=device), torch.tensor(float(scale), device=device), validate_args=False)
    return Independent(d.expand(torch.Size([dim])), 1) if dim > 0 else d


class _BuiltinProcess(AffineProcess):
    """Base of the built-in kinds: keeps the Gaussian initial distribution in closed form for ``pf_initial_sample``."""

    def __init__(self, kind_id, mean_scale, parameters, dim, dt, inc_scale, init_mean, init_scale):
        self.init_mean = _as_tensor(init_mean)
        self.init_scale = _as_tensor(init_scale)
        super().__init__(
            mean_scale, parameters, _increment(inc_scale, dim), _normal_init(self.init_mean, self.init_scale, dim)
        )
        self.kernel_kind = KernelKind(kind_id, dim, dt, inc_scale)
        self._dim = dim

    def to(self, device):
        super().to(device)
        self.init_mean, self.init_scale = self.init_mean.to(device), self.init_scale.to(device)
        self._initial_kernel = _normal_init(self.init_mean, self.init_scale, self._dim)
        self._event_shape = None
        return self


class AR(_BuiltinProcess):
    """AR(1): ``x' = alpha + beta x + sigma e`` with ``x_0 ~ N(alpha, sigma)`` (tests/filters/models.py:10-25)."""

    def __init__(self, alpha, beta, sigma, initial: Optional[Tuple] = None):
        init = initial or (alpha, sigma)
        super().__init__(
            L.HID_LINEAR, lambda x, a, b, s: (a + b * x.value, s), (alpha, beta, sigma), 0, 1.0, 1.0, init[0], init[1]
        )


class RandomWalk(_BuiltinProcess):
    """Random walk ``x' = x + sigma e``.  ``dim`` = 2 or 3 gives a D-dimensional diagonal walk with ``sigma`` of shape
    ``(D,)`` / ``(B, D)``; ``dim = 0`` a scalar walk (``sigma`` scalar or ``(B,)``).  ``dim=None`` infers it from an
    UNBATCHED sigma (last dimension 2 or 3): pass it explicitly when sigma carries a batch dimension - a ``(B,)`` sigma
    with B in {2, 3} is otherwise indistinguishable from a vector walk."""

    def __init__(self, sigma, initial_mean=0.0, initial_scale=None, dim: Optional[int] = None):
        sigma = _as_tensor(sigma)
        if dim is None:
            dim = sigma.shape[-1] if sigma.dim() == 1 and sigma.shape[-1] in (2, 3) else 0
            if sigma.dim() > 1:
                raise L.PfAmdError("RandomWalk: pass `dim` explicitly when sigma has a batch dimension")
        zeros, ones = torch.zeros_like(sigma), torch.ones_like(sigma)
        init_scale = sigma if initial_scale is None else initial_scale
        super().__init__(
            L.HID_LINEAR, lambda x, a, b, s: (a + b * x.value, s), (zeros, ones, sigma), dim, 1.0, 1.0,
            _as_tensor(initial_mean) + zeros, init_scale,
        )


class SineDiffusion(_BuiltinProcess):
    """The README's sine diffusion, Euler-Maruyama with step ``dt``: ``x_0 ~ N(0, 1)``."""

    def __init__(self, gamma, sigma, dt=0.1, initial=(0.0, 1.0)):
        super().__init__(
            L.HID_SINE_EM, lambda x, g, s: (x.value + torch.sin(x.value - g) * dt, s), (gamma, sigma), 0, dt, sqrt(dt),
            initial[0], initial[1],
        )


class Verhulst(_BuiltinProcess):
    """Verhulst (logistic) diffusion used as the volatility process of the SV example."""

    def __init__(self, kappa, gamma, sigma, dt=1.0, initial: Optional[Tuple] = None):
        init = initial or (gamma, sigma)
        super().__init__(
            L.HID_VERHULST_EM,
            lambda x, k, g, s: (x.value + k * (g - x.value) * x.value * dt, s * x.value),
            (kappa, gamma, sigma), 0, dt, sqrt(dt), init[0], init[1],
        )


class Lorenz63(_BuiltinProcess):
    """Stochastic Lorenz-63 (examples/lorenz.ipynb): drift ``(-s(x-y), r x - y - x z, x y - b z)``."""

    def __init__(self, s, r, b, sigma=1.0, dt=1e-2, initial_mean=(-5.91652, -5.52332, 24.5723),
                 initial_scale=(sqrt(10.0),) * 3):
        def ms(x, s_, r_, b_, sig):
            v = x.value
            f = torch.stack(
                (-s_ * (v[..., 0] - v[..., 1]), r_ * v[..., 0] - v[..., 1] - v[..., 0] * v[..., 2],
                 v[..., 0] * v[..., 1] - b_ * v[..., 2]), dim=-1,
            )
            sig = sig if sig.dim() == 0 or sig.shape[-1] == 3 else sig.unsqueeze(-1)
            return v + f * dt, sig

        super().__init__(L.HID_LORENZ63_EM, ms, (s, r, b, sigma), 3, dt, sqrt(dt), initial_mean, initial_scale)


class OrnsteinUhlenbeck(_BuiltinProcess):
    """OU process, exact discretisation over ``dt``: ``x' = gamma + (x - gamma) e^{-kappa dt} + sigma_dt e``."""

    def __init__(self, kappa, gamma, sigma, dt=1.0, initial: Optional[Tuple] = None):
        def ms(x, k, g, s):
            e = torch.exp(-k * dt)
            return g + (x.value - g) * e, s * torch.sqrt((1.0 - torch.exp(-2.0 * k * dt)) / (2.0 * k))

        init = initial or (gamma, _as_tensor(sigma) / torch.sqrt(2.0 * _as_tensor(kappa)))
        super().__init__(L.HID_OU, ms, (kappa, gamma, sigma), 0, dt, 1.0, init[0], init[1])


class StochasticVolatilityModel(StateSpaceModel):
    """``y ~ N(mu, scale = x)`` on a scalar volatility process - the SV notebook's observation density at
    ``skew = 0, kurt = 1`` (where its SinhArcsinh transform is the identity)."""

    def __init__(self, hidden: AffineProcess, mu, observe_every_step: int = 1):
        super().__init__(hidden, lambda x, m: Normal(m, x.value, validate_args=False), (mu,), observe_every_step)
        hk = getattr(hidden, "kernel_kind", None)
        if hk is not None and hk.dim == 1:
            kind = KernelKind(hk.hid_kind, hk.dim, hk.dt, hk.inc_scale)
            kind.obs_kind, kind.obs_dim = L.OBS_SV, 1
            self.kernel_kind = kind


# ----------------------------------------------------------------------------------------------------------------
# parameter rows for the kernels
# ----------------------------------------------------------------------------------------------------------------
def _expand(p: torch.Tensor, b: int, inner: Sequence[int], dtype, device) -> torch.Tensor:
    """Broadcast a parameter to ``(B, *inner)``: scalar | ``inner`` | ``(B,)`` | ``(B, *inner)`` | ``(1,)``."""
    p = p.to(device=device, dtype=dtype)
    inner = tuple(inner)
    full = (b,) + inner
    if p.numel() == 1:
        return p.reshape(()).expand(full)
    if tuple(p.shape) == full:
        return p
    per_filter, per_component = tuple(p.shape) == (b,), tuple(p.shape) == inner
    if per_filter and per_component and b > 1:
        # B == D: a (B,) vector could be one value per filter or one per state component.  The reference resolves this by
        # torch broadcasting against (N, B, D) tensors - trailing dimensions align, i.e. per component - so that is kept,
        # loudly: a per-filter parameter must be given as (B, 1) / (B, D)
        import warnings

        warnings.warn(f"parameter of shape {tuple(p.shape)} with batch = state dim = {b}: read as one value per state "
                      "component (torch broadcasting); pass shape (B, 1) for one value per filter", stacklevel=3)
    if per_component:
        return p.unsqueeze(0).expand(full)
    if per_filter:
        return p.reshape((b,) + (1,) * len(inner)).expand(full)
    if p.dim() == len(full) and p.shape[0] == b and all(ps in (1, fs) for ps, fs in zip(p.shape[1:], inner)):
        return p.expand(full)  # (B, 1, ...): explicitly one value per filter
    raise L.PfAmdError(f"cannot broadcast a parameter of shape {tuple(p.shape)} to (batch={b}, {inner})")


def pack_params(ssm: StateSpaceModel, b: int, dtype, device) -> torch.Tensor:
    """``(B, NP)`` rows ``[hp0[D] hp1[D] hp2[D] hp3[D] | A[OxD] | b[O] | s[O]]`` (``pf_model.params``)."""
    kind = ssm.kernel_kind
    if kind is None:
        raise L.PfAmdError("model has no built-in kernel kind")
    d, o = kind.dim, kind.obs_dim
    cols = []
    hp = [] if kind.is_user else list(ssm.hidden.parameters)  # (a user process keeps its parameters in its callable)
    for k in range(4):
        if k < len(hp):
            cols.append(_expand(hp[k], b, (d,), dtype, device))
        else:
            cols.append(torch.zeros((b, d), dtype=dtype, device=device))
    if kind.obs_kind == L.OBS_LINEAR:
        a, ob, os_ = ssm.parameters
        if d > 1 and ssm.n_dim == 0 and a.dim() >= 1:
            # a SCALAR observation of a vector state (event_shape = Size([])): ``a`` is the (D,) row - or (B, D), one row per
            # filter - that the reference unsqueezes to (1, D) (proposals/utils.py:248 ``c.unsqueeze(-2)``)
            a = a.unsqueeze(-2)
        cols.append(_expand(a, b, (o, d), dtype, device).reshape(b, o * d))
        cols.append(_expand(ob, b, (o,), dtype, device))
        cols.append(_expand(os_, b, (o,), dtype, device))
    else:
        (mu,) = ssm.parameters
        cols.append(torch.zeros((b, o * d), dtype=dtype, device=device))
        cols.append(_expand(mu, b, (o,), dtype, device))
        cols.append(torch.ones((b, o), dtype=dtype, device=device))
    return torch.cat([c.reshape(b, -1) for c in cols], dim=1).contiguous()
