"""SISR (``pyfilter/filters/particle/sisr.py:7-56``): resample the filters whose ESS fell below the threshold,
propagate, weigh."""
import torch

from ... import _lib as L
from ... import ops
from .base import ParticleFilter
from .state import ParticleFilterCorrection, ParticleFilterPrediction
from .utils import log_likelihood


class SISR(ParticleFilter):
    _FILTER_KIND = L.FILTER_SISR

    def predict(self, state: ParticleFilterCorrection) -> ParticleFilterPrediction:
        self._refresh_parameters()  # parameter tensors are read live (in-place updates between moves)
        ts_state = state.get_timeseries_state()
        weights = state.weights
        prev_inds = state.previous_indices
        batched = weights.dim() > 1
        has_event = len(ts_state.event_shape) > 0

        w_cols = ops.to_cols(weights)
        W_cols, _, ess = ops.normalize_cols(w_cols, want_w=True, want_ess=True)  # sanitises the stored weights
        if w_cols.data_ptr() != weights.data_ptr():
            weights.copy_(ops.from_cols(w_cols, batched))
        W = ops.from_cols(W_cols, batched)

        mask = ess < self._resample_threshold  # (B,), decided per filter
        if not mask.any():  # host sync, exactly where the reference has one (sisr.py:25)
            return ParticleFilterPrediction(ts_state, weights, W, indices=prev_inds)

        n, b = w_cols.shape[1], w_cols.shape[0]
        kind = self._resampler_kind()
        if kind is not None:
            colmask = mask.to(torch.uint8).contiguous()
            anc = ops.to_cols(prev_inds.to(torch.int32)).contiguous().clone()
            if kind == L.RESAMPLE_SYSTEMATIC:
                u = self._uniforms(int(ts_state.time_index), b, w_cols)
                ops.systematic_cols(W_cols, u, normalized=True, colmask=colmask, idx=anc)
            else:
                ops.multinomial_cols(W_cols, self._run_seed, step=int(ts_state.time_index), colmask=colmask, idx=anc)
            x_soa = ops.gather_soa(ops.to_soa(ts_state.value, batched, has_event), anc, colmask)
            resampled_x = ops.from_soa(x_soa, batched, has_event)
            resampled_indices = ops.from_cols(anc, batched).long()
        else:
            # user-supplied resampler: the reference's masked route (sisr.py:29-44) with torch indexing
            sub = self._resampler(W[..., mask] if batched else W, normalized=True)
            um = mask.unsqueeze(0) if batched else mask
            resampled_indices = prev_inds.masked_scatter(um, sub) if batched else sub
            vals = ts_state.value
            if batched:
                temp = vals[sub, mask]
                um_x = um.unsqueeze(-1) if has_event else um
                resampled_x = vals.masked_scatter(um_x, temp)
            else:
                resampled_x = vals[sub]

        um = mask.unsqueeze(0) if batched else mask
        resampled_weights = weights.masked_fill(um, 0.0)
        W = W.masked_fill(um, 1.0 / n)
        return ParticleFilterPrediction(ts_state.copy(values=resampled_x), resampled_weights, W, indices=resampled_indices)

    def correct(self, y: torch.Tensor, prediction: ParticleFilterPrediction) -> ParticleFilterCorrection:
        x, weights = self.proposal.sample_and_weight(y, prediction)
        new_weights = weights + prediction.weights
        ll = log_likelihood(weights, prediction.normalized_weights)
        return ParticleFilterCorrection(x, new_weights, ll, prediction.indices)
