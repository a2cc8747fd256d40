"""Optimal proposal for linear-Gaussian observations (``proposals/linear.py:13-89`` with ``find_optimal_density``,
``proposals/utils.py:219-267``).  On a built-in model the whole closed form (precision, <=3x3 inverse + Cholesky,
three log-densities) is evaluated per particle in registers by ``pf_sample_and_weight`` / the fused step kernel - the
reference spends 22 % of its step in a batched LU of (N, 1, 1) matrices here."""
import torch
from torch.distributions import MultivariateNormal, Normal
from torch.linalg import cholesky_ex

from .... import _lib as L
from ....timeseries import AffineProcess, LinearStateSpaceModel
from ....utils import construct_diag_from_flat
from .base import Proposal


class LinearGaussianObservations(Proposal):
    _KERNEL_PROPOSAL = L.PROP_LGO

    def __init__(self, *_ignored):
        # the README calls ``LinearGaussianObservations(0)`` (README.md:78); v0.29.0's ctor takes no such argument
        super().__init__()

    def set_model(self, model):
        if not isinstance(model.hidden, AffineProcess) or not isinstance(model, LinearStateSpaceModel):
            raise ValueError("Model combination not supported!")
        return super().set_model(model)

    # ---- generic route: the reference's tensor algebra with PyTorch-ROCm ops (user-defined affine processes) ----
    def _optimal_density(self, y, loc, h_var_inv, o_var_inv, c):
        model = self._model
        hidden_is_1d, obs_is_1d = model.hidden.n_dim == 0, model.n_dim == 0
        if hidden_is_1d:
            c = c.unsqueeze(-1)
        c_u = c if not obs_is_1d else c.unsqueeze(-2)
        c_t = c_u.transpose(-2, -1)
        o_inv_cov = construct_diag_from_flat(o_var_inv, model.event_shape)
        cov = (construct_diag_from_flat(h_var_inv, model.hidden.event_shape) + c_t.matmul(o_inv_cov).matmul(c_u)).inverse()
        t_1 = h_var_inv * loc
        if hidden_is_1d:
            t_1 = t_1.unsqueeze(-1)
        t_2 = o_inv_cov.squeeze(-1) * y.unsqueeze(-1) if obs_is_1d else o_inv_cov.matmul(y)
        mean = cov.matmul(t_1.unsqueeze(-1) + c_t.matmul(t_2.unsqueeze(-1))).squeeze(-1)
        if hidden_is_1d:
            return Normal(mean.squeeze(-1), cov[..., 0, 0].sqrt(), validate_args=False)
        return MultivariateNormal(mean, scale_tril=cholesky_ex(cov)[0], validate_args=False)

    def sample_and_weight(self, y, prediction):
        x = prediction.get_timeseries_state()
        if self.uses_kernels:
            return self._kernel_sample_and_weight(y, x)
        mean, scale = self._model.hidden.mean_scale(x)
        x_dist = self._model.hidden.build_density(x)
        a, b, s = self._model.parameters
        kernel = self._optimal_density(y - b, mean, scale.pow(-2.0), s.pow(-2.0), a)
        x_result = x.copy(values=mean).propagate_from(values=kernel.sample())
        return x_result, self._weight_with_kernel(y, x_dist, x_result, kernel)

    def pre_weight(self, y, x):
        if self.uses_kernels:
            return self._kernel_pre_weight(y, x)
        _, h_scale = self._model.hidden.mean_scale(x)
        a, b, s = self._model.parameters
        if self._model.hidden.n_dim == 0:
            a = a.unsqueeze(-1)
        obs_is_1d = self._model.n_dim == 0
        a_u = a if not obs_is_1d else a.unsqueeze(-2)
        cov = construct_diag_from_flat(s.pow(2.0), self._model.event_shape) + a_u.matmul(
            construct_diag_from_flat(h_scale.pow(2.0), self._model.hidden.event_shape)
        ).matmul(a_u.transpose(-2, -1))
        if obs_is_1d:
            return Normal(b + a.squeeze(-1) * x.value, cov[..., 0, 0].sqrt(), validate_args=False).log_prob(y)
        o_loc = b + (a_u @ x.value.unsqueeze(-1)).squeeze(-1)
        return MultivariateNormal(o_loc, scale_tril=cholesky_ex(cov)[0], validate_args=False).log_prob(y)

    def copy(self) -> "Proposal":
        return LinearGaussianObservations()
