"""Bootstrap proposal (``proposals/bootstrap.py:4-17``): propose from the dynamics, weigh by the observation density."""
from .... import _lib as L
from .base import Proposal


class Bootstrap(Proposal):
    _KERNEL_PROPOSAL = L.PROP_BOOTSTRAP

    def sample_and_weight(self, y, prediction):
        x = prediction.get_timeseries_state()
        if self.uses_kernels:
            return self._kernel_sample_and_weight(y, x)
        new_x = self._model.hidden.propagate(x)
        return new_x, self._model.build_density(new_x).log_prob(y)

    def copy(self) -> "Proposal":
        return Bootstrap(self._pre_weight_func if self._custom_pre_weight else None)
