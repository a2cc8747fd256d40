"""Minimal model layer with the members SURVEY.md §8(a) row M lists. Our own code, written from the
call sites in the reference (``proposals/linear.py``, ``proposals/bootstrap.py``, ``filters/base.py`` ...)."""
import torch
from torch.distributions import AffineTransform, Independent, Normal, TransformedDistribution

from . import result  # noqa: F401


def _as_tensor(p):
    return p if isinstance(p, torch.Tensor) else torch.as_tensor(p, dtype=torch.get_default_dtype())


class TimeseriesState(dict):
    def __init__(self, time_index, values, event_shape):
        super().__init__()
        self.time_index = time_index if isinstance(time_index, torch.Tensor) else torch.tensor(time_index)
        self._values = values
        self.event_shape = torch.Size(event_shape)

    @property
    def value(self):
        if callable(self._values):  # lazy sampling: pyfilter passes ``density.sample``
            self._values = self._values()
        return self._values

    @value.setter
    def value(self, v):
        self._values = v

    @property
    def batch_shape(self):
        v = self.value
        return v.shape[: v.dim() - len(self.event_shape)]

    def copy(self, values):
        return TimeseriesState(self.time_index, values, self.event_shape)

    def propagate_from(self, values, time_increment=1):
        return TimeseriesState(self.time_index + time_increment, values, self.event_shape)


class StructuralStochasticProcess:
    def __init__(self, parameters, initial_kernel, initial_parameters=None):
        self.parameters = tuple(_as_tensor(p) for p in parameters)
        self._initial_kernel = initial_kernel
        self._initial_parameters = (
            None if initial_parameters is None else tuple(_as_tensor(p) for p in initial_parameters)
        )

    @property
    def initial_distribution(self):
        return self._initial_kernel(*(self._initial_parameters or self.parameters))

    @property
    def event_shape(self):
        return self.initial_distribution.event_shape

    @property
    def n_dim(self):
        return len(self.event_shape)

    def initial_sample(self, shape=torch.Size([])):
        # the initial distribution is EXPANDED to the requested batch shape, not sampled `shape` times: with parameters of
        # shape (B,) - theta-particles on the filters' batch dimension (inference/sequential/base.py:31-34) - its batch
        # shape is already (B,) and pyfilter asks for (N, B) samples (filters/particle/base.py:88-90).  For an unbatched
        # distribution the two are the same draw (one torch.normal call over the (N, B, [D]) expanded parameters).
        dist = self.initial_distribution
        shape = torch.Size(shape)
        if len(shape):
            dist = dist.expand(shape)
        return TimeseriesState(0, dist.sample(), self.event_shape)

    def build_density(self, x):
        raise NotImplementedError()

    def propagate(self, x, time_increment=1):
        return x.propagate_from(values=self.build_density(x).sample, time_increment=time_increment)


class AffineProcess(StructuralStochasticProcess):
    def __init__(self, mean_scale, parameters, increment_distribution, initial_kernel, initial_parameters=None):
        super().__init__(parameters, initial_kernel, initial_parameters)
        self._mean_scale = mean_scale
        self.increment_distribution = increment_distribution

    def mean_scale(self, x, parameters=None):
        loc, scale = self._mean_scale(x, *(parameters or self.parameters))
        return torch.broadcast_tensors(loc, scale)

    def build_density(self, x):
        loc, scale = self.mean_scale(x)
        return TransformedDistribution(
            self.increment_distribution, AffineTransform(loc, scale, event_dim=self.n_dim), validate_args=False
        )


class AffineEulerMaruyama(AffineProcess):
    """``loc = x + f(x)*dt``, ``scale = g(x)``, increments supplied by the caller (README.md:44-62)."""

    def __init__(self, dynamics, parameters, increment_distribution, dt, initial_kernel, initial_parameters=None):
        self.dt = dt

        def _ms(x, *params):
            f, g = dynamics(x, *params)
            return x.value + f * dt, g

        super().__init__(_ms, parameters, increment_distribution, initial_kernel, initial_parameters)


class LinearModel(AffineProcess):
    """``x' = b + A x + s * eps`` - the process the reference's own 2-D acceptance model is built from
    (tests/filters/models.py:28-38: ``ts.LinearModel((a, sigma), inc_dist, initial_kernel)``).  ``parameters`` is
    ``(a, s)`` or ``(a, b, s)``; the initial kernel is called with ``(a, b, s)`` (the reference's lambda takes
    ``m_, _, s_``).  A scalar / vector ``a`` multiplies elementwise, a matrix acts through ``matmul``."""

    def __init__(self, parameters, increment_distribution, initial_kernel, initial_parameters=None):
        parameters = tuple(_as_tensor(p) for p in parameters)
        if len(parameters) == 2:
            a, s = parameters
            parameters = (a, torch.zeros_like(s), s)

        def _ms(x, a, b, s):
            v = x.value
            loc = b + (a @ v.unsqueeze(-1)).squeeze(-1) if a.dim() >= 2 else b + a * v
            return loc, s

        super().__init__(_ms, parameters, increment_distribution, initial_kernel, initial_parameters)


class StateSpaceModel:
    def __init__(self, hidden, f, parameters, observe_every_step=1):
        self.hidden = hidden
        self._f = f
        self.parameters = tuple(_as_tensor(p) for p in parameters)
        self.observe_every_step = observe_every_step
        self._event_shape = None

    def build_density(self, x):
        return self._f(x, *self.parameters)

    @property
    def event_shape(self):
        if self._event_shape is None:
            self._event_shape = self.build_density(self.hidden.initial_sample()).event_shape
        return self._event_shape

    @property
    def n_dim(self):
        return len(self.event_shape)


class LinearStateSpaceModel(StateSpaceModel):
    def __init__(self, hidden, parameters, event_shape, observe_every_step=1):
        parameters = tuple(_as_tensor(p) for p in parameters)
        if len(parameters) == 2:
            a, s = parameters
            parameters = (a, torch.zeros_like(s), s)
        self._obs_event_shape = torch.Size(event_shape)
        hidden_is_1d = hidden.n_dim == 0

        def _f(x, a, b, s):
            if hidden_is_1d:
                # (a scalar state under a VECTOR observation - a of shape (O,): b + a x with x broadcast along the observation
                # axis, the reading under which proposals/utils.py:243-245 ``c = c.unsqueeze(-1)`` is an (O, 1) matrix)
                loc = b + a * (x.value.unsqueeze(-1) if len(self._obs_event_shape) == 1 else x.value)
            else:
                loc = b + (a @ x.value.unsqueeze(-1)).squeeze(-1)
            d = Normal(loc, s, validate_args=False)
            return Independent(d, 1) if len(self._obs_event_shape) == 1 else d

        super().__init__(hidden, _f, parameters, observe_every_step)
        self._event_shape = self._obs_event_shape
