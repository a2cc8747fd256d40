"""Drop-in for ``pyfilter/filters/utils.py``."""
import torch

from .. import ops


def batched_gather(x: torch.Tensor, indices: torch.Tensor, dim: int = 0) -> torch.Tensor:
    """``x.gather`` along the particle axis with ``indices`` broadcast over the trailing state dim
    (filters/utils.py:4-21), executed by ``pf_gather``.  ``x``: ``(N, [B], [D])``, ``indices``: ``(N, [B])``."""
    if dim != 0:
        raise NotImplementedError("only the particle axis (dim 0) is resampled on this path")
    batched = indices.dim() > 1
    has_event = x.dim() > indices.dim()
    soa = ops.to_soa(x, batched, has_event)
    idx = ops.to_cols(indices.to(torch.int32))
    return ops.from_soa(ops.gather_soa(soa, idx), batched, has_event)
