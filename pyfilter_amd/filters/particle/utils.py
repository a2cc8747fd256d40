"""Drop-ins for ``pyfilter/filters/particle/utils.py``: ``log_likelihood`` (:7-22) and
``get_filter_mean_and_variance`` (:26-65), both as HIP reductions with fp64 accumulators."""
from typing import Optional, Tuple

import torch

from ... import ops
from ...timeseries import TimeseriesState


def log_likelihood(importance_weights: torch.Tensor, weights: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``max v + log sum_i W_i exp(v_i - max)`` over the particle axis; ``W = 1/N`` when omitted."""
    batched = importance_weights.dim() > 1
    v = ops.to_cols(importance_weights)
    W = None if weights is None else ops.to_cols(weights)
    out = ops.loglik_cols(v, W)
    return out if batched else out[0]


def get_filter_mean_and_variance(
    state: TimeseriesState, weights: torch.Tensor, covariance: bool = False, keep_dim: bool = True
) -> Tuple[torch.Tensor, torch.Tensor]:
    """Weighted mean and variance ``([B], max(D, 1))`` of a particle set (the ``covariance=True`` branch is only used
    by the out-of-scope Gaussian proposals)."""
    if covariance and state.event_shape:
        raise NotImplementedError("full covariance is only needed by GPF proposals (out of scope, SURVEY.md §2 row 8)")
    batched = weights.dim() > 1
    has_event = len(state.event_shape) > 0
    mean, var = ops.moments_soa(ops.to_soa(state.value, batched, has_event), ops.to_cols(weights))
    if not batched:
        mean, var = mean[0], var[0]
    if not keep_dim:
        mean, var = mean.squeeze(-1), var.squeeze(-1)
    return mean, var
