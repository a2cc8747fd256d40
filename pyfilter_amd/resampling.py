"""
Drop-in for ``pyfilter/resampling.py``: ``systematic`` (:24-52), ``multinomial`` (:55-65) and ``residual`` (:68-105).

``systematic`` runs as: per-tile fp64 sums -> tile scan with an fp64 carry rounded per element (bit-identical to
torch's CPU ``cumsum`` whenever the fp64 partial sums are exact) -> LDS-window ``searchsorted`` (side=left).  Given the
same normalised weights and uniforms it returns the reference's indices bit for bit.
"""
from typing import Optional, Union

import torch

from . import _lib as L
from . import ops

_DEFAULT_SEED = 0x5EED_2024
_calls = 0


def _draw_u(b: int, like: torch.Tensor) -> torch.Tensor:
    return torch.empty(b, device=like.device, dtype=like.dtype).uniform_()


def systematic(w: torch.Tensor, normalized: bool = False, u: Optional[Union[torch.Tensor, float]] = None) -> torch.Tensor:
    """
    Systematic resampling of ``(N,)`` or ``(N, B)`` (log) weights; returns int64 indices of the same shape.

    Args:
        w: log weights (``normalized=False``; sanitised in place like the reference's ``normalize``) or normalised
            weights (``normalized=True``).
        u: optional uniforms, one per column (``(B, 1)`` / ``(B,)`` / float) - unlike the reference, ``u`` is honoured
            for 1-D input too (reference bug: resampling.py:14 drops it).
    """
    batched = w.dim() > 1
    cols = ops.to_cols(w)
    b = cols.shape[0]
    if u is None:
        uu = _draw_u(b, cols)
    elif isinstance(u, torch.Tensor):
        if u.numel() == b:
            uu = u.to(device=cols.device)
        elif u.dim() == 2 and tuple(u.shape) == tuple(cols.shape):  # one u per grid position (reference's own test)
            uu = u.to(device=cols.device)
        else:
            raise ValueError(f"u must hold one uniform per batch column ({b}), got shape {tuple(u.shape)}")
    else:
        uu = torch.full((b,), float(u), device=cols.device, dtype=cols.dtype)
    idx = ops.systematic_cols(cols, uu, normalized)
    if not normalized and cols.data_ptr() != w.data_ptr():
        w.copy_(ops.from_cols(cols, batched))
    return ops.from_cols(idx, batched).long()


def multinomial(w: torch.Tensor, normalized: bool = False, seed: Optional[int] = None) -> torch.Tensor:
    """Multinomial resampling: N iid inverse-CDF draws per column (statistical parity with ``torch.multinomial``)."""
    global _calls
    from .utils import normalize

    batched = w.dim() > 1
    W = w if normalized else normalize(w)
    cols = ops.to_cols(W)
    _calls += 1
    idx = ops.multinomial_cols(cols, _DEFAULT_SEED if seed is None else seed, step=_calls)
    return ops.from_cols(idx, batched).long()


def residual(w: torch.Tensor, normalized: bool = False, seed: Optional[int] = None) -> torch.Tensor:
    """Residual resampling (``resampling.py:68-105``): particle ``j`` first receives
    ``floor(N W_j)`` offspring - positions ``0 .. M-1`` in ancestor order, exactly the reference's ``repeat_interleave``
    - and the remaining ``N - M`` positions are multinomial draws from the residuals ``N W_j - floor(N W_j)``
    (``pf_multinomial``; statistical parity with ``torch.multinomial``).  The reference accepts 1-D weights only; here a
    batch dimension works the same way per column.  Not on the fused path: the arithmetic around the multinomial
    kernel is a handful of device-side torch ops."""
    global _calls
    from .utils import normalize

    L.require_gpu(w)
    batched = w.dim() > 1
    W = w if normalized else normalize(w)
    cols = ops.to_cols(W).double()                       # (B, N); N * W and its floor are exact in float64
    n = cols.shape[1]
    mw = cols * n
    floored = mw.floor()
    counts_cum = floored.cumsum(-1)                      # integers < 2^53: exact
    m = counts_cum[:, -1:]                               # deterministic offspring per column
    pos = torch.arange(n, device=cols.device, dtype=torch.float64).expand_as(cols).contiguous()
    det = torch.searchsorted(counts_cum, pos, right=True).clamp_(max=n - 1)   # position i -> first j with C_j > i
    res = mw - floored
    tot = res.sum(-1, keepdim=True)
    res_w = torch.where(tot > 0, res / tot.clamp_min(1e-300), torch.full_like(res, 1.0 / n)).to(W.dtype)
    _calls += 1
    drawn = ops.multinomial_cols(res_w.contiguous(), _DEFAULT_SEED if seed is None else seed, step=_calls).long()
    idx = torch.where(pos < m, det, drawn)
    return ops.from_cols(idx, batched)
