"""Optimal proposal for linear-Gaussian observations (``proposals/linear.py:13-89`` with ``find_optimal_density``,
``proposals/utils.py:219-267``).  On a built-in model the whole closed form (precision, <=3x3 inverse + Cholesky,
three log-densities) is evaluated per particle in registers by ``pf_sample_and_weight`` / the fused step kernel - the
reference spends 22 % of its step in a batched LU of (N, 1, 1) matrices here."""
import math

import torch
from torch.distributions import MultivariateNormal, Normal

from .... import _lib as L
from ....timeseries import AffineProcess, LinearStateSpaceModel
from .base import Proposal


class _ObservationUpdate:
    """The Gaussian algebra of the optimal proposal on the step-by-step route (user-defined affine processes; torch ops on
    the device), in *innovation form*.  Prior per particle: ``x' ~ N(m, diag(h^2))`` (``m, h`` = the user's ``mean_scale``);
    observation ``y ~ N(b + A x', diag(s^2))``.  With ``P = A diag(h^2)`` (O x D), the innovation covariance
    ``S = P A^T + diag(s^2)`` (O x O) and the gain ``G = P^T S^{-1}`` (D x O, through the Cholesky factor of S):

        x' | y  ~  N(m + G (y - b - A m),  diag(h^2) - G P)          log p(y | .) = log N(y; ., S)

    - the same posterior the reference reaches through the D x D precision matrix and its inverse
    (``proposals/utils.py:219-267``), without forming or inverting a precision.  States and observations are handled as
    vectors throughout (a scalar state / observation is a vector of length one)."""

    def __init__(self, model: LinearStateSpaceModel, h_scale: torch.Tensor):
        self.vec_x, self.vec_y = model.hidden.n_dim > 0, model.n_dim > 0
        a, b, s = model.parameters
        dt = h_scale.dtype
        a = a.to(dt)
        if self.vec_x and self.vec_y:
            self.A = a                                   # (O, D)
        elif self.vec_x:
            self.A = a.unsqueeze(-2)                     # a scalar observation of a vector state: the (D,) row
        elif self.vec_y:
            self.A = a.unsqueeze(-1)                     # a vector observation of a scalar state: the (O,) column
        else:
            self.A = a.reshape(a.shape + (1, 1))         # scalar / scalar (a may carry one value per filter)
        self.b = (b if self.vec_y else b.unsqueeze(-1)).to(dt)
        s2 = (s if self.vec_y else s.unsqueeze(-1)).to(dt).square()
        h2 = (h_scale if self.vec_x else h_scale.unsqueeze(-1)).square()          # (..., D)
        self.h2 = h2
        self.P = self.A * h2.unsqueeze(-2)                                         # (..., O, D) = A diag(h^2)
        o = self.P.shape[-2]
        self.S = self.P @ self.A.transpose(-2, -1) + torch.diag_embed(s2.expand(self.P.shape[:-2] + (o,)))
        self.S_chol = torch.linalg.cholesky_ex(self.S)[0]  # (no info check: that is a host sync per move; proposals/linear.py:84)

    def _vec_y(self, y):
        return y if self.vec_y else y.unsqueeze(-1)

    def log_marginal(self, y: torch.Tensor, at: torch.Tensor) -> torch.Tensor:
        """``log N(y; b + A at, S)`` per particle (``at``: the points the observation mean is taken at)."""
        x = at if self.vec_x else at.unsqueeze(-1)
        r = (self._vec_y(y) - self.b - (self.A @ x.unsqueeze(-1)).squeeze(-1)).unsqueeze(-1)   # (..., O, 1)
        half = torch.linalg.solve_triangular(self.S_chol.expand(r.shape[:-2] + self.S_chol.shape[-2:]), r, upper=False)
        o = r.shape[-2]
        log_det = self.S_chol.diagonal(dim1=-2, dim2=-1).log().sum(-1)
        return -0.5 * half.square().sum((-2, -1)) - log_det - 0.5 * o * math.log(2.0 * math.pi)

    def posterior(self, y: torch.Tensor, m: torch.Tensor):
        """The optimal proposal ``p(x' | x, y)`` as a torch distribution over the state's own event shape."""
        mv = m if self.vec_x else m.unsqueeze(-1)
        innov = (self._vec_y(y) - self.b - (self.A @ mv.unsqueeze(-1)).squeeze(-1)).unsqueeze(-1)
        shape = torch.broadcast_shapes(self.P.shape[:-2], innov.shape[:-2])
        P = self.P.expand(shape + self.P.shape[-2:])
        Sc = self.S_chol.expand(shape + self.S_chol.shape[-2:])
        gain_t = torch.cholesky_solve(P, Sc)                                         # S^{-1} P = G^T  (..., O, D)
        mean = mv + (gain_t.transpose(-2, -1) @ innov.expand(shape + innov.shape[-2:])).squeeze(-1)
        cov = torch.diag_embed(self.h2.expand(shape + self.h2.shape[-1:])) - P.transpose(-2, -1) @ gain_t
        if not self.vec_x:
            return Normal(mean.squeeze(-1), cov[..., 0, 0].sqrt(), validate_args=False)
        cov = 0.5 * (cov + cov.transpose(-2, -1))
        return MultivariateNormal(mean, scale_tril=torch.linalg.cholesky_ex(cov)[0], validate_args=False)  # proposals/utils.py:267


class LinearGaussianObservations(Proposal):
    _KERNEL_PROPOSAL = L.PROP_LGO

    def __init__(self, *_ignored):
        # the README calls ``LinearGaussianObservations(0)`` (README.md:78); v0.29.0's ctor takes no such argument
        super().__init__()

    def set_model(self, model):
        if not isinstance(model.hidden, AffineProcess) or not isinstance(model, LinearStateSpaceModel):
            raise ValueError("Model combination not supported!")
        return super().set_model(model)

    def sample_and_weight(self, y, prediction):
        x = prediction.get_timeseries_state()
        if self.uses_kernels:
            return self._kernel_sample_and_weight(y, x)
        # step-by-step route of a user-defined affine process (``proposals/linear.py:38-55``): draw from the optimal
        # proposal, weigh with observation density x transition density / proposal density
        mean, scale = self._model.hidden.mean_scale(x)
        x_dist = self._model.hidden.build_density(x)
        kernel = _ObservationUpdate(self._model, scale).posterior(y, mean)
        x_result = x.copy(values=mean).propagate_from(values=kernel.sample())
        return x_result, self._weight_with_kernel(y, x_dist, x_result, kernel)

    def pre_weight(self, y, x):
        if self.uses_kernels:
            return self._kernel_pre_weight(y, x)
        # the APF's first-stage weight of this proposal (``proposals/linear.py:57-86``): the observation's marginal density
        # with the transition noise integrated out - its mean taken at the CURRENT particles, as the reference does
        _, h_scale = self._model.hidden.mean_scale(x)
        return _ObservationUpdate(self._model, h_scale).log_marginal(y, x.value)

    def copy(self) -> "Proposal":
        return LinearGaussianObservations()
