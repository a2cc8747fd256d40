"""Auxiliary particle filter of Pitt & Shephard (``pyfilter/filters/particle/apf.py:9-46``): first-stage weights
select the ancestors every step, second-stage weights correct for them."""
import torch

from ... import _lib as L
from ... import ops
from ..utils import batched_gather
from .base import ParticleFilter
from .state import ParticleFilterCorrection, ParticleFilterPrediction
from .utils import log_likelihood


class APF(ParticleFilter):
    _FILTER_KIND = L.FILTER_APF

    def predict(self, state: ParticleFilterCorrection) -> ParticleFilterPrediction:
        self._refresh_parameters()  # parameter tensors are read live (in-place updates between moves)
        normalized = state.normalized_weights()
        old_indices = torch.arange(normalized.shape[0], device=normalized.device)
        if self.batch_shape:
            old_indices = old_indices.unsqueeze(-1).expand(self.particles)
        return ParticleFilterPrediction(state.timeseries_state, state.weights, normalized, old_indices)

    def correct(self, y: torch.Tensor, prediction: ParticleFilterPrediction) -> ParticleFilterCorrection:
        ts_state = prediction.get_timeseries_state()
        pre_weights = self.proposal.pre_weight(y, ts_state)
        resample_weights = pre_weights + prediction.weights
        batched = resample_weights.dim() > 1

        kind = self._resampler_kind()
        if kind == L.RESAMPLE_SYSTEMATIC:
            cols = ops.to_cols(resample_weights)
            u = self._uniforms(int(ts_state.time_index), cols.shape[0], cols)
            indices = ops.from_cols(ops.systematic_cols(cols, u, normalized=False), batched).long()
        elif kind == L.RESAMPLE_MULTINOMIAL:
            cols = ops.to_cols(resample_weights)
            W, _, _ = ops.normalize_cols(cols, want_w=True)
            indices = ops.from_cols(ops.multinomial_cols(W, self._run_seed, step=int(ts_state.time_index)), batched).long()
        else:
            indices = self._resampler(resample_weights)

        resampled_x = ts_state.copy(values=batched_gather(ts_state.value, indices, 0))
        zeros = torch.zeros_like(resample_weights)
        resampled_prediction = ParticleFilterPrediction(resampled_x, zeros, zeros + 1.0 / pre_weights.shape[0], None)

        x, weights = self._proposal.sample_and_weight(y, resampled_prediction)
        weights = weights - batched_gather(pre_weights, indices, 0)
        # log p(y_t | y_{1:t-1}) ~ log(1/N sum exp w') + log sum W_{t-1} exp(pre)   (apf.py:44); the second term is
        # evaluated in the max-shifted form (equal whenever the reference's unshifted sum is finite)
        ll = log_likelihood(weights) + log_likelihood(pre_weights, prediction.normalized_weights)
        return ParticleFilterCorrection(x, weights, ll, indices)
