"""Weighted-sample Gaussian fit behind the SMC^2 proposal (``pyfilter/inference/utils.py:42-76``)."""
import math
from typing import NamedTuple

import torch
from torch.distributions import MultivariateNormal
from torch.linalg import cholesky_ex


# ---- theta-level arithmetic: O(B) host-orchestration math on the theta-particles' weights, plain torch ops on whatever
# device the weights live on (the particle-level hot path is what runs in the HIP kernels) ---------------------------------
def theta_normalize(log_w: torch.Tensor) -> torch.Tensor:
    """``pyfilter.utils.normalize`` (utils.py:49-64) for the ``(B,)`` theta log-weights; the input is left untouched."""
    w = torch.nan_to_num(log_w, nan=-math.inf, posinf=-math.inf)
    p = torch.softmax(w - w.max(), dim=0)
    return torch.where(p.sum() == 0.0, torch.full_like(p, 1.0 / p.shape[0]), p) if p.numel() else p


def theta_ess(log_w: torch.Tensor) -> torch.Tensor:
    """``get_ess`` (utils.py:8-20) of the theta log-weights."""
    return theta_normalize(log_w).pow(2.0).sum().reciprocal()


def theta_systematic(W: torch.Tensor, u: torch.Tensor) -> torch.Tensor:
    """Systematic resampling of ``B`` normalised theta-weights with the uniform ``u`` (``resampling.py:24-52``): identical
    on every rank that holds the same ``W`` and ``u``."""
    b = W.shape[0]
    cdf = W.cumsum(0)
    cdf[-1:].fill_(1.0)  # (fill_: the scalar travels as a kernel argument; ``cdf[-1] = 1.0`` stages it through a host tensor)
    # (``u`` is a host scalar - the lock-step CPU stream: as a Python float it rides in the kernel arguments instead of a
    # pageable host -> device copy the host would wait for)
    u = float(u) if not (isinstance(u, torch.Tensor) and u.device == W.device) else u
    probs = (torch.arange(b, device=W.device, dtype=W.dtype) + u) / b
    return torch.searchsorted(cdf, probs).clamp_max(b - 1)


class MeanChol(NamedTuple):
    mean: torch.Tensor
    chol: torch.Tensor


def calc_mean_chol(x: torch.Tensor, w: torch.Tensor) -> MeanChol:
    """Weighted mean and lower Cholesky factor of the weighted covariance of the rows of ``x (B, P)`` (``w (B,)``
    normalised).  A covariance that is not positive definite falls back to its diagonal (utils.py:42-57) - decided on
    the device, without a host round trip."""
    mean = w @ x
    centred = x - mean
    cov = (w * centred.t()).matmul(centred)
    chol, info = cholesky_ex(cov)
    diag = cov.diagonal().clamp_min(0).sqrt().diag_embed()
    return MeanChol(mean, torch.where((info > 0).any(), diag, chol))


def construct_mvn(x: torch.Tensor, w: torch.Tensor, scale: float = 1.0) -> MultivariateNormal:
    """``MultivariateNormal(mean_w(x), scale_tril = scale * chol(cov_w(x)))`` (utils.py:60-76)."""
    mc = calc_mean_chol(x, w)
    return MultivariateNormal(mc.mean, scale_tril=scale * mc.chol, validate_args=False)
