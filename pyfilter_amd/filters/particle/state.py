"""State objects of the particle filters (``pyfilter/filters/particle/state.py:14-211``).

Tensors are exposed in the reference's layout - particles on dim 0, optional batch on dim 1, optional state dim last -
as *views* of the library's ``(B, N)`` / ``(D, B, N)`` buffers.  Ancestors are int64 at this API like the
reference's (``searchsorted`` output), int32 inside the kernels.
"""
from collections import OrderedDict
from typing import Any, Dict, Optional

import torch
from torch import Tensor

from ...timeseries import StateSpaceModel, TimeseriesState
from ...utils import normalize
from ..state import Correction, Prediction
from .utils import get_filter_mean_and_variance


def _rows_mask(mask: Tensor, like: Tensor) -> Tensor:
    """``mask (B,)`` shaped to select whole rows of ``like (B, ...)``."""
    return mask.reshape(mask.shape + (1,) * (like.dim() - mask.dim()))


def _masked_assign(dst: Tensor, src: Tensor, mask: Tensor):
    """``dst[mask] = src[mask]`` along dim 0, in place.  A boolean mask of the right length goes through ``torch.where`` -
    indexing with it would first count its set entries on the host (a device round trip per call; PMMH does this for every
    per-filter quantity of every move)."""
    if mask.dtype == torch.bool and mask.dim() == 1 and dst.shape == src.shape and dst.dim() >= 1 and mask.shape[0] == dst.shape[0]:
        dst.copy_(torch.where(mask.reshape(mask.shape + (1,) * (dst.dim() - 1)), src, dst))
    else:
        dst[mask] = src[mask]


class ParticleFilterPrediction(Prediction):
    def __init__(self, prev_x: TimeseriesState, weights: Tensor, normalized_weights: Tensor, indices: Tensor):
        self.prev_x = prev_x
        self.weights = weights
        self.normalized_weights = normalized_weights
        self.indices = indices

    @classmethod
    def equally_weighted(cls, x: TimeseriesState, like: Tensor) -> "ParticleFilterPrediction":
        """Freshly resampled particles: log-weights 0, normalised weights 1/N (``like``: any ``(N, [B])`` tensor)."""
        zero = torch.zeros_like(like)
        return cls(x, zero, torch.full_like(like, 1.0 / like.shape[0]), None)

    def get_timeseries_state(self) -> TimeseriesState:
        return self.prev_x

    def create_state_from_prediction(self, model: StateSpaceModel, propagate=None):
        """Propagate-only move for an unobserved step / NaN observation: weights carried, ``ll = 0`` (:38-42)."""
        x_new = propagate(self.prev_x) if propagate is not None else model.hidden.propagate(self.prev_x)
        new_ll = torch.zeros(self.normalized_weights.shape[1:], device=self.weights.device, dtype=self.weights.dtype)
        return ParticleFilterCorrection(x_new, self.weights, new_ll, self.indices)


class ParticleFilterCorrection(Correction):
    _KEYS = ("_x", "_w", "_ll", "_prev_inds", "_mean", "_var")

    def __init__(self, x: TimeseriesState, w: Tensor, ll: Tensor, prev_indices: Optional[Tensor], _moments=None,
                 _anc32=None):
        super().__init__()
        # the fused paths hand over the kernels' own int32 ``(B, N)`` ancestor buffer (``_anc32 = (buffer, batched)``)
        # instead of ``prev_indices``: the reference's int64 tensor is widened from it on first access - an online filter
        # move does not pay 8 B per particle for ancestors nobody looks at
        self._anc32 = None
        self["_x"] = x
        self["_w"] = w
        self["_ll"] = ll
        if prev_indices is not None:
            self["_prev_inds"] = prev_indices
        else:
            assert _anc32 is not None
            self._anc32 = _anc32
        # ... and the moments their kernels already reduced; otherwise they are reduced on first use
        if _moments is not None:
            self["_mean"], self["_var"] = _moments

    def __missing__(self, key):
        if key == "_prev_inds" and self._anc32 is not None:
            wide = self._view32().long()
            dict.__setitem__(self, key, wide)  # (keeps the int32 buffer valid: both show the same ancestors)
            return wide
        raise KeyError(key)

    def __setitem__(self, key, value):
        if key == "_prev_inds":
            self._anc32 = None  # set from outside: the kernels' buffer no longer is what the state shows
        super().__setitem__(key, value)

    def _view32(self) -> Tensor:
        buf, batched = self._anc32
        return buf.t() if batched else buf[0]

    def ancestors32(self) -> Tensor:
        """The ancestors as the kernels take them: int32 ``(B, N)`` - their own buffer while it is current."""
        from ... import ops

        if self._anc32 is not None:
            return self._anc32[0]
        return ops.to_cols(self["_prev_inds"].to(torch.int32))

    def _restarted(self) -> "ParticleFilterCorrection":
        """The same particles / weights / ancestors (shared tensors) with a zero log-likelihood of their own: the incoming
        state of a run whose result must not accumulate into this state's ``_ll`` (``FilterResult`` aliases it, result.py:34)."""
        other = type(self).__new__(type(self))
        dict.update(other, self)
        other._anc32 = self._anc32
        dict.__setitem__(other, "_ll", torch.zeros_like(self["_ll"]))
        return other

    def _ensure_moments(self):
        if "_mean" not in self:
            self["_mean"], self["_var"] = get_filter_mean_and_variance(self["_x"], self.normalized_weights())

    @property
    def timeseries_state(self) -> TimeseriesState:
        return self["_x"]

    @property
    def weights(self) -> Tensor:
        return self["_w"]

    @property
    def previous_indices(self) -> Tensor:
        return self["_prev_inds"]

    def get_loglikelihood(self) -> Tensor:
        return self["_ll"]

    def get_mean(self) -> Tensor:
        self._ensure_moments()
        return self["_mean"]

    def get_variance(self) -> Tensor:
        self._ensure_moments()
        return self["_var"]

    def get_covariance(self) -> Tensor:
        if len(self.timeseries_state.event_shape) == 0:
            return self.get_variance()
        w = self.normalized_weights()
        x = self.timeseries_state.value
        mean = (w.unsqueeze(-1) * x).sum(dim=0)
        c = x - mean
        return (w.view(w.shape + (1, 1)) * (c.unsqueeze(-1) @ c.unsqueeze(-2))).sum(dim=0)

    def normalized_weights(self) -> Tensor:
        """``normalize(self.weights)`` - sanitises the stored log-weights in place, as the reference does."""
        return normalize(self.weights)

    def get_timeseries_state(self) -> TimeseriesState:
        return self.timeseries_state

    def _packed(self):
        """The ``(D + 1, B, N)`` buffer a fused run left particles AND log-weights in (``ParticleFilter._filter_block_lean``) -
        while the state still shows exactly that buffer: whole-filter moves then take both in one launch."""
        xl = getattr(self, "_xl", None)
        if xl is None or not xl.is_cuda or "_x" not in self or "_w" not in self:
            return None
        d = xl.shape[0] - 1
        if self["_x"].value.data_ptr() != xl.data_ptr() or self["_w"].data_ptr() != xl[d].data_ptr() or self["_w"].dim() != 2:
            return None
        return xl

    def _adopt_packed(self, xl: Tensor):
        from ... import ops

        d = xl.shape[0] - 1
        ts = self.timeseries_state
        self["_x"] = ts.copy(values=ops.from_soa(xl[:d], True, len(ts.event_shape) > 0))
        self["_w"] = ops.from_cols(xl[d], True)
        self._xl = xl

    def resample(self, indices: Tensor):
        """Gather whole filters along the batch dim (``:150-158``): the particle planes, weights
        and ancestors move with ``pf_columns_gather`` (whole contiguous columns in the library layout)."""
        from ... import ops

        self._ensure_moments()
        ts = self.timeseries_state
        xl = self._packed()
        if xl is not None and indices.dtype == torch.int64 and indices.is_contiguous() and indices.numel() == xl.shape[1]:
            self._adopt_packed(ops.gather_columns(xl, indices))
        else:
            self["_x"] = ts.copy(values=ops.gather_filters(ts.value, indices))
            self["_w"] = ops.gather_filters(self.weights, indices)
        self["_ll"][indices] = self["_ll"][indices]
        if self._anc32 is not None and self._anc32[1]:  # the int32 buffer moves; the int64 view is rebuilt on demand
            self._anc32 = (ops.to_cols(ops.gather_filters(self._view32(), indices)), True)
            dict.pop(self, "_prev_inds", None)
        else:
            self["_prev_inds"] = ops.gather_filters(self["_prev_inds"], indices)
        self["_mean"] = self["_mean"][indices]
        self["_var"] = self["_var"][indices]

    def predict_path(self, model: StateSpaceModel, num_steps: int):
        """Forecast ``num_steps`` ahead from this state's particles (``particle/state.py:173-174``)."""
        return model.sample_states(num_steps, x_0=self.timeseries_state)

    def _gather_moments(self, indices: Tensor):
        """The part of ``FilterResult.resample(entire_history=True)`` that reaches into a recorded state (see there)."""
        self._ensure_moments()
        self["_mean"], self["_var"] = self["_mean"][indices], self["_var"][indices]

    def exchange(self, other: "ParticleFilterCorrection", mask: Tensor):
        """Overwrite the filters selected by ``mask`` with those of ``other`` (``:160-168``), ``pf_columns_exchange``."""
        from ... import ops

        self._ensure_moments()
        other._ensure_moments()
        ts = self.timeseries_state
        xl, xl_other = self._packed(), other._packed()
        if (xl is not None and xl_other is not None and xl.shape == xl_other.shape and xl.dtype == xl_other.dtype
                and mask.dtype == torch.bool and mask.is_contiguous() and mask.numel() == xl.shape[1]):
            ops.exchange_columns(xl, xl_other, mask)  # (in place: the state's views keep showing the buffer)
        else:
            new_x = ops.exchange_filters(ts.value, other.timeseries_state.value, mask)
            if new_x.data_ptr() != ts.value.data_ptr():
                self["_x"] = ts.copy(values=new_x)
            self["_w"] = ops.exchange_filters(self["_w"], other.weights, mask)
        _masked_assign(self["_ll"], other.get_loglikelihood(), mask)
        if self._anc32 is not None and self._anc32[1] and getattr(other, "_anc32", None) is not None and other._anc32[1] \
                and mask.dtype == torch.bool:
            self._anc32 = (ops.to_cols(ops.exchange_filters(self._view32(), other._view32(), mask)), True)
            dict.pop(self, "_prev_inds", None)
        else:
            self["_prev_inds"] = ops.exchange_filters(self["_prev_inds"], other.previous_indices, mask)
        # (NEW tensors, not writes into the old ones: a FilterResult's moment log may still remember this state's rows by
        # reference - MomentLog.append - and its history must not change with the state)
        for k in ("_mean", "_var"):
            if mask.dtype == torch.bool and mask.dim() == 1 and self[k].shape == other[k].shape:
                self[k] = torch.where(_rows_mask(mask, self[k]), other[k], self[k])
            else:
                fresh = self[k].clone()
                fresh[mask] = other[k][mask]
                self[k] = fresh

    def state_dict(self) -> Dict[str, Any]:
        self._ensure_moments()
        result = OrderedDict((k, self[k]) for k in self._KEYS[1:])
        for k, v in self.items():  # (anything a caller attached)
            if k not in result and isinstance(v, torch.Tensor):
                result[k] = v
        result["_x"] = {"time_index": self.timeseries_state.time_index, "value": self.timeseries_state.value}
        return result

    def load_state_dict(self, state_dict: Dict[str, Any]):
        values = state_dict["_x"]["value"]
        mine = self.timeseries_state
        assert mine.value.shape == values.shape, f"shape mismatch: {mine.value.shape} != {values.shape}"
        self["_x"] = mine.propagate_from(values=values, time_increment=int(state_dict["_x"]["time_index"]) - int(mine.time_index))
        for k in ("_w", "_ll", "_prev_inds", "_mean", "_var"):
            self[k] = state_dict[k]

    def __repr__(self):
        ts = self.timeseries_state
        return f"{self.__class__.__name__}(time_index: {ts.time_index}, event_shape: {ts.event_shape})"
