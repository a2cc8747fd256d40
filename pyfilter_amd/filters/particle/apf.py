"""Auxiliary particle filter of Pitt & Shephard (``pyfilter/filters/particle/apf.py:9-46``): first-stage weights
select the ancestors every step, second-stage weights correct for them.  ``batch_filter`` / ``filter`` of built-in models
never come here (one fused kernel per step); this is the ``predict`` / ``correct`` pair of the step-by-step route."""
import torch

from ... import _lib as L
from ..utils import batched_gather
from .base import ParticleFilter
from .state import ParticleFilterCorrection, ParticleFilterPrediction
from .utils import log_likelihood


class APF(ParticleFilter):
    _FILTER_KIND = L.FILTER_APF

    def predict(self, state: ParticleFilterCorrection) -> ParticleFilterPrediction:
        """Nothing moves yet (apf.py:16-23): the particles with their normalised weights and identity ancestors."""
        self._refresh_parameters()  # parameter tensors are read live (in-place updates between moves)
        W = state.normalized_weights()
        identity = torch.arange(W.shape[0], device=W.device)
        return ParticleFilterPrediction(state.timeseries_state, state.weights, W,
                                        identity.unsqueeze(-1).expand(self.particles) if self.batch_shape else identity)

    def correct(self, y: torch.Tensor, prediction: ParticleFilterPrediction) -> ParticleFilterCorrection:
        """apf.py:25-46.  ``ll_t = log(1/N sum exp w'') + log sum W_{t-1} exp(first stage)`` - the second term in the
        max-shifted form (equal whenever the reference's unshifted sum is finite)."""
        parents = prediction.get_timeseries_state()
        first_stage = self.proposal.pre_weight(y, parents)
        chosen = self._ancestors_of(first_stage + prediction.weights, int(parents.time_index))
        survivors = ParticleFilterPrediction.equally_weighted(
            parents.copy(values=batched_gather(parents.value, chosen, 0)), like=first_stage)
        x, second_stage = self._proposal.sample_and_weight(y, survivors)
        second_stage = second_stage - batched_gather(first_stage, chosen, 0)
        ll = log_likelihood(second_stage) + log_likelihood(first_stage, prediction.normalized_weights)
        return ParticleFilterCorrection(x, second_stage, ll, chosen)
