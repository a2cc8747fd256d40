"""
Drop-in for ``pyfilter/utils.py``: ``normalize`` (:49-64) and ``get_ess`` (:8-20), computed by the HIP kernels
``k_reduce_logw`` + ``k_normalize_write`` (one reduce pass, one write pass; the reference makes ~5 passes).
"""
import torch

from . import ops


def normalize(weights: torch.Tensor) -> torch.Tensor:
    """
    Normalizes a 1D ``(N,)`` or 2D ``(N, B)`` array of log weights over the particle axis (dim 0).

    Like the reference this **sanitises ``weights`` in place**: NaN and +inf become -inf and - because the reference
    calls ``nan_to_num_`` without ``neginf`` - -inf becomes the lowest finite value, so an all ``-inf`` column
    normalises to ``1/N`` (utils.py:57-62).
    """
    cols = ops.to_cols(weights)
    W, _, _ = ops.normalize_cols(cols, want_w=True)
    if cols.data_ptr() != weights.data_ptr():  # a layout copy was made: write the sanitised values back
        weights.copy_(ops.from_cols(cols, weights.dim() > 1))
    return ops.from_cols(W, weights.dim() > 1)


def get_ess(weights: torch.Tensor, normalized: bool = False) -> torch.Tensor:
    """ESS ``1 / sum W^2`` per column from (log) weights (utils.py:8-20)."""
    if normalized:
        return weights.pow(2.0).sum(dim=0).reciprocal()
    cols = ops.to_cols(weights)
    _, _, ess = ops.normalize_cols(cols, want_w=False, want_ess=True)
    if cols.data_ptr() != weights.data_ptr():
        weights.copy_(ops.from_cols(cols, weights.dim() > 1))
    return ess if weights.dim() > 1 else ess[0]
