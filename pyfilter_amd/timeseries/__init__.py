"""
Host-side mirror of the slice of ``stochproc.timeseries`` (v0.3.0, an external dependency of the reference:
``/root/reference/pyproject.toml:34``) that pyfilter's SISR/APF hot path touches - SURVEY.md §8(a) row M:

``TimeseriesState``, ``StructuralStochasticProcess``, ``AffineProcess``, ``AffineEulerMaruyama``, ``StateSpaceModel``,
``LinearStateSpaceModel`` with the members ``mean_scale / build_density / propagate / initial_sample / n_dim /
event_shape / parameters / observe_every_step``.

Two ways to define a model:

* **built-in kinds** (``pyfilter_amd.timeseries.models``): closed-form processes whose arithmetic lives in the HIP
  kernels (``csrc/pf_models.hpp``) - these run the fused step (one HIP kernel per time step);
* **user callables** (exactly the reference's way, README.md:44-67): ``mean_scale`` / observation builders are python
  callables evaluated with PyTorch-ROCm ops on the device; normalise / scan / search / gather / moments still run in
  the HIP kernels.
"""
from math import sqrt
from typing import Callable, Optional, Sequence

import torch
from torch.distributions import AffineTransform, Distribution, Independent, Normal, TransformedDistribution

from .. import _lib as L


def _as_tensor(p, device=None, dtype=None) -> torch.Tensor:
    if isinstance(p, torch.Tensor):
        return p
    return torch.as_tensor(p, dtype=dtype or torch.get_default_dtype(), device=device)


class TimeseriesState(dict):
    """State of a timeseries: time index + values (possibly a lazily evaluated sampler, as pyfilter passes
    ``density.sample``: proposals/linear.py:53)."""

    def __init__(self, time_index, values, event_shape: torch.Size):
        super().__init__()
        self.time_index = time_index if isinstance(time_index, torch.Tensor) else torch.tensor(time_index)
        self._values = values
        self.event_shape = torch.Size(event_shape)

    @property
    def value(self) -> torch.Tensor:
        if callable(self._values):
            self._values = self._values()
        return self._values

    @value.setter
    def value(self, v):
        self._values = v

    @property
    def batch_shape(self) -> torch.Size:
        v = self.value
        return v.shape[: v.dim() - len(self.event_shape)]

    def copy(self, values) -> "TimeseriesState":
        return TimeseriesState(self.time_index, values, self.event_shape)

    def propagate_from(self, values, time_increment=1) -> "TimeseriesState":
        return TimeseriesState(self.time_index + time_increment, values, self.event_shape)


class KernelKind:
    """The ``kernel_id`` of a built-in model: everything ``csrc/pf_models.hpp`` needs (see ``pf_model`` in
    ``include/pf_amd.h``)."""

    def __init__(self, hid_kind: int, dim: int, dt: float, inc_scale: float):
        self.hid_kind = hid_kind
        self.dim = max(dim, 1)
        self.dt = dt
        self.inc_scale = inc_scale
        self.obs_kind = None
        self.obs_dim = None

    @property
    def is_user(self) -> bool:
        """A user-defined affine process: its ``mean_scale`` callable is evaluated with PyTorch-ROCm ops once per step and
        handed to the fused kernels as (loc, scale) planes (``PF_HID_USER_AFFINE``)."""
        return self.hid_kind == L.HID_USER_AFFINE


class StructuralStochasticProcess:
    def __init__(self, parameters: Sequence, initial_kernel: Callable[..., Distribution], initial_parameters=None):
        self.parameters = tuple(_as_tensor(p) for p in parameters)
        self._initial_kernel = initial_kernel
        self._initial_parameters = None if initial_parameters is None else tuple(_as_tensor(p) for p in initial_parameters)
        self._event_shape = None

    @property
    def device(self) -> torch.device:
        for p in self.parameters:
            if p.is_cuda:
                return p.device
        return torch.device("cpu")

    @property
    def initial_distribution(self) -> Distribution:
        return self._initial_kernel(*(self._initial_parameters or self.parameters))

    @property
    def event_shape(self) -> torch.Size:
        if self._event_shape is None:
            self._event_shape = self.initial_distribution.event_shape
        return self._event_shape

    @property
    def n_dim(self) -> int:
        return len(self.event_shape)

    def initial_sample(self, shape=torch.Size([])) -> TimeseriesState:
        return TimeseriesState(0, self.initial_distribution.sample(torch.Size(shape)), self.event_shape)

    def build_density(self, x: TimeseriesState) -> Distribution:
        raise NotImplementedError()

    def propagate(self, x: TimeseriesState, time_increment=1) -> TimeseriesState:
        return x.propagate_from(values=self.build_density(x).sample, time_increment=time_increment)

    def to(self, device):
        self.parameters = tuple(p.to(device) for p in self.parameters)
        if self._initial_parameters is not None:
            self._initial_parameters = tuple(p.to(device) for p in self._initial_parameters)
        return self

    def cuda(self):
        return self.to("cuda")


class AffineProcess(StructuralStochasticProcess):
    """``x' = loc(x) + scale(x) * eps`` with ``(loc, scale) = mean_scale(x, *parameters)`` and ``eps`` drawn from
    ``increment_distribution``."""

    kernel_kind: Optional[KernelKind] = None

    def __init__(self, mean_scale, parameters, increment_distribution, initial_kernel, initial_parameters=None):
        super().__init__(parameters, initial_kernel, initial_parameters)
        self._mean_scale = mean_scale
        self.increment_distribution = increment_distribution

    def mean_scale(self, x: TimeseriesState, parameters=None):
        loc, scale = self._mean_scale(x, *(parameters or self.parameters))
        return torch.broadcast_tensors(loc, _as_tensor(scale, device=loc.device, dtype=loc.dtype))

    def build_density(self, x: TimeseriesState) -> Distribution:
        loc, scale = self.mean_scale(x)
        return TransformedDistribution(
            self.increment_distribution, AffineTransform(loc, scale, event_dim=self.n_dim), validate_args=False
        )

    def to(self, device):
        super().to(device)
        inc = self.increment_distribution
        base = inc.base_dist if isinstance(inc, Independent) else inc
        if isinstance(base, Normal):
            nb = Normal(base.loc.to(device), base.scale.to(device), validate_args=False)
            self.increment_distribution = Independent(nb, inc.reinterpreted_batch_ndims) if isinstance(inc, Independent) else nb
        return self


class AffineEulerMaruyama(AffineProcess):
    """Euler-Maruyama discretisation ``loc = x + f(x) dt``, ``scale = g(x)`` (README.md:44-62)."""

    def __init__(self, dynamics, parameters, increment_distribution, dt, initial_kernel, initial_parameters=None):
        self.dt = dt

        def _ms(x, *params):
            f, g = dynamics(x, *params)
            # x + f dt as ONE elementwise launch (the callable's launches set the pace of a fused move: four ~2.5 - 4 us kernels
            # for the README's sine diffusion, three with this)
            if isinstance(f, torch.Tensor) and not isinstance(dt, torch.Tensor):
                return torch.add(x.value, f, alpha=dt), g
            if isinstance(f, torch.Tensor) and isinstance(dt, torch.Tensor) and f.dtype == x.value.dtype:
                return torch.addcmul(x.value, f, dt), g
            return x.value + f * dt, g

        super().__init__(_ms, parameters, increment_distribution, initial_kernel, initial_parameters)
        self._dynamics = dynamics

    def drift_scale(self, x: TimeseriesState):
        """``(f(x), g(x), dt)`` for the fused kernels, which form ``x + f dt`` at the parent themselves
        (``pf_filter_args.user_dt``: one elementwise launch per move less) - or None when that does not apply (a tensor-valued
        ``dt``, a drift that is not a tensor of the state's dtype): the caller then takes ``mean_scale``."""
        if isinstance(self.dt, torch.Tensor) or not self.dt:
            return None
        f, g = self._dynamics(x, *self.parameters)
        if not isinstance(f, torch.Tensor) or f.dtype != x.value.dtype:
            return None
        f, g = torch.broadcast_tensors(f, _as_tensor(g, device=f.device, dtype=f.dtype))
        return f, g, float(self.dt)


def _user_affine_kind(hidden: "AffineProcess") -> Optional[KernelKind]:
    """The kernel kind of a USER-DEFINED affine process (the reference's plug-in seam: ``AffineProcess(lambda x, *p: (loc,
    scale), parameters, increment_distribution, ...)``, README.md:44-67) - available when its increments are centred
    Gaussians of one common scale (``Normal(0, s)``, possibly ``.to_event(1)``): then ``x' = loc(x) + scale(x) s e`` is what
    the fused kernels compute from the (loc, scale) planes.  Anything else stays on the step-by-step route."""
    # exactly these two classes: a subclass may override ``build_density`` / ``propagate`` (other dynamics than loc + scale * e),
    # which the kernels would silently ignore - such a process keeps the step-by-step route unless it says otherwise
    # (``fused_affine = True`` on the instance / class)
    if type(hidden) not in (AffineProcess, AffineEulerMaruyama) and not getattr(hidden, "fused_affine", False):
        return None
    inc = hidden.increment_distribution
    base = inc.base_dist if isinstance(inc, Independent) else inc
    if not isinstance(base, Normal):
        return None
    # the Normal(0, s) test needs the VALUES of the increment parameters: one host copy per increment distribution (PMMH
    # rebuilds the model around the same increments at every move; a device round trip there would make the host wait for
    # the running re-filter), remembered by the tensors' identity and in-place version
    key = (id(base.loc), base.loc._version, id(base.scale), base.scale._version)
    cached = getattr(base, "_pf_user_kind", None)
    if cached is None or cached[0] != key:
        loc, scale = base.loc.detach().reshape(-1).cpu(), base.scale.detach().reshape(-1).cpu()
        ok = loc.numel() > 0 and not bool((loc != 0).any()) and not bool((scale != scale[0]).any())
        cached = (key, float(scale[0]) if ok else None)
        try:
            base._pf_user_kind = cached
        except AttributeError:
            pass
    if cached[1] is None:
        return None
    dim = hidden.n_dim and hidden.event_shape.numel()
    if dim > L.MAX_D:
        return None
    return KernelKind(L.HID_USER_AFFINE, dim, 1.0, cached[1])


class StateSpacePath:
    """Sampled hidden states and observations (the slice of ``stochproc.timeseries.result.StateSpacePath`` that
    ``ParticleFilterCorrection.predict_path`` exposes): ``get_paths() -> (x (steps, *shape), y (steps, *shape))``."""

    def __init__(self, xs, ys):
        self._x, self._y = torch.stack(xs, 0), torch.stack(ys, 0)

    def get_paths(self):
        return self._x, self._y


class StateSpaceModel:
    """Hidden process + observation density builder ``f(x, *parameters) -> Distribution``."""

    def __init__(self, hidden: StructuralStochasticProcess, f, parameters, observe_every_step: int = 1):
        self.hidden = hidden
        self._f = f
        self.parameters = tuple(_as_tensor(p) for p in parameters)
        self.observe_every_step = observe_every_step
        self._event_shape = None
        self.kernel_kind: Optional[KernelKind] = None

    def build_density(self, x: TimeseriesState) -> Distribution:
        return self._f(x, *self.parameters)

    @property
    def event_shape(self) -> torch.Size:
        if self._event_shape is None:
            self._event_shape = self.build_density(self.hidden.initial_sample()).event_shape
        return self._event_shape

    @property
    def n_dim(self) -> int:
        return len(self.event_shape)

    def sample_states(self, steps: int, samples=torch.Size([]), x_0: Optional[TimeseriesState] = None) -> StateSpacePath:
        """``steps`` moves of the hidden process from ``x_0`` (default: an initial sample) with an observation drawn at
        every one of them - forecasting from a filter state (``particle/state.py:173-174``).  Off the hot path: plain
        torch draws on the model's device."""
        x = x_0 if x_0 is not None else self.hidden.initial_sample(samples)
        xs, ys = [], []
        for _ in range(steps):
            x = self.hidden.propagate(x)
            xs.append(x.value)
            ys.append(self.build_density(x).sample())
        return StateSpacePath(xs, ys)

    def to(self, device):
        self.hidden.to(device)
        self.parameters = tuple(p.to(device) for p in self.parameters)
        return self

    def cuda(self):
        return self.to("cuda")


class LinearStateSpaceModel(StateSpaceModel):
    """``y = b + A x + s v``; parameters ``(a, s)`` or ``(a, b, s)`` (normalised to ``(a, b, s)`` as
    proposals/linear.py:48 expects)."""

    def __init__(self, hidden, parameters, event_shape: torch.Size = torch.Size([]), observe_every_step: int = 1):
        parameters = tuple(_as_tensor(p) for p in parameters)
        if len(parameters) == 2:
            a, s = parameters
            parameters = (a, torch.zeros_like(s), s)
        obs_event = torch.Size(event_shape)
        hidden_is_1d = hidden.n_dim == 0

        def _f(x, a, b, s):
            if hidden_is_1d:  # (a scalar state under a vector observation: a of shape (O,), x broadcast along the observation axis)
                loc = b + a * (x.value.unsqueeze(-1) if len(obs_event) == 1 else x.value)
            else:
                loc = b + (a @ x.value.unsqueeze(-1)).squeeze(-1)
            d = Normal(loc, s, validate_args=False)
            return Independent(d, 1) if len(obs_event) == 1 else d

        super().__init__(hidden, _f, parameters, observe_every_step)
        self._event_shape = obs_event
        hk = getattr(hidden, "kernel_kind", None)
        if hk is None and isinstance(hidden, AffineProcess):
            hk = _user_affine_kind(hidden)
        if hk is not None:
            o = obs_event.numel() if len(obs_event) else 1
            if hk.dim <= L.MAX_D and o <= L.MAX_O and (hk.dim > 1 or o == 1):
                kind = KernelKind(hk.hid_kind, hk.dim, hk.dt, hk.inc_scale)
                kind.obs_kind, kind.obs_dim = L.OBS_LINEAR, o
                self.kernel_kind = kind


from . import models  # noqa: E402,F401
