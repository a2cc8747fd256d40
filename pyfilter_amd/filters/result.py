"""``FilterResult`` (``pyfilter/filters/result.py:14-164``): filter means / variances incl. the initial state, the
running log-likelihood and the recorded states."""
from copy import deepcopy
from typing import Generic, List, TypeVar

import torch

from ..container import BaseResult, BoolOrInt, make_dequeue
from .state import Correction

TCorrection = TypeVar("TCorrection", bound=Correction)


class FilterResult(BaseResult, Generic[TCorrection]):
    def __init__(self, init_state: TCorrection, record_states: BoolOrInt, record_moments: BoolOrInt):
        super().__init__()
        # NB: aliases the initial state's ``_ll`` tensor, exactly like the reference (result.py:34)
        self._loglikelihood = init_state.get_loglikelihood()
        self.tensor_tuples.make_deque("filter_means", maxlen=record_moments)
        self.tensor_tuples.make_deque("filter_variances", maxlen=record_moments)
        self._states = make_dequeue(maxlen=record_states)
        self.append(init_state)

    @property
    def loglikelihood(self) -> torch.Tensor:
        return self._loglikelihood

    @property
    def filter_means(self) -> torch.Tensor:
        """``(timesteps + 1, [batch], latent dim)`` - row 0 is the initial state."""
        return self.tensor_tuples.get_as_tensor("filter_means")

    @property
    def filter_variance(self) -> torch.Tensor:
        return self.tensor_tuples.get_as_tensor("filter_variances")

    @property
    def states(self) -> List[TCorrection]:
        return list(self._states)

    @property
    def latest_state(self) -> TCorrection:
        return self._states[-1]

    def append(self, state: TCorrection):
        self.tensor_tuples["filter_means"].append(state.get_mean())
        self.tensor_tuples["filter_variances"].append(state.get_variance())
        self._loglikelihood.add_(state.get_loglikelihood())
        self._states.append(state)
        return self

    def _extend_fused(self, means: torch.Tensor, variances: torch.Tensor, ll_total: torch.Tensor, last_state):
        """Adopts the rows the fused kernels wrote (views, no copies): ``means`` / ``variances`` are
        ``(steps, [B], D)`` for the steps after the state already appended."""
        self.tensor_tuples["filter_means"].extend(means.unbind(0))
        self.tensor_tuples["filter_variances"].extend(variances.unbind(0))
        self._loglikelihood.add_(ll_total)
        self._states.append(last_state)
        return self

    def exchange(self, other: "FilterResult", mask: torch.Tensor):
        """Overwrites the filters selected by ``mask`` (batch dim) with those of ``other`` (result.py:76-95)."""
        self._loglikelihood[mask] = other.loglikelihood[mask]
        for old_tt, new_tt in zip(self.tensor_tuples.values(), other.tensor_tuples.values()):
            for old, new in zip(old_tt, new_tt):
                old[mask] = new[mask]
        for ns, os_ in zip(other.states, self.states):
            os_.exchange(ns, mask)
        return self

    def resample(self, indices: torch.Tensor, entire_history: bool = True):
        """Gathers whole filters along the batch dim (result.py:97-117)."""
        self._loglikelihood.copy_(self._loglikelihood[indices])
        if entire_history:
            for tt in self.tensor_tuples.values():
                for tens in tt:
                    tens.copy_(tens[indices])
        for s in self.states:
            s.resample(indices)
        return self

    def state_dict(self):
        res = super().state_dict()
        res["state"] = self.latest_state.state_dict()
        res["log_likelihood"] = self.loglikelihood
        return res

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._loglikelihood = state_dict["log_likelihood"]
        assert len(self.states) == 1, "Can only handle case when we have 1 state!"
        self.latest_state.load_state_dict(state_dict["state"])

    def copy(self) -> "FilterResult":
        return deepcopy(self)

    def __repr__(self):
        return f"FilterResult(ll: {self._loglikelihood!r}, num_observations: {self.filter_means.shape[0]})"
