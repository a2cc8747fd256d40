"""``BaseFilter``: model handling, batch shape and the generic (step-by-step) driver.

The public surface is the reference's (``pyfilter/filters/base.py:17-232``: constructor arguments, ``ssm``,
``initialize_model``, ``set_batch_shape``, ``initialize_with_result``, ``filter``, ``batch_filter``, ``copy``,
``smooth``).  The driver is organised around this library's move schedule (``filters/schedule.py``), which the fused
HIP route (``ParticleFilter._batch_filter_fused``) shares: one observation = ``pad`` propagate-only moves + one weighted
(or, for an all-NaN observation, propagate-only) move.
"""
from typing import Generic, Iterable, Sequence, TypeVar, Union

import torch
from tqdm import tqdm

from ..timeseries import StateSpaceModel
from .result import FilterResult
from .schedule import unobserved_moves_before
from .state import Correction, Prediction

TCorrection = TypeVar("TCorrection", bound=Correction)
TPrediction = TypeVar("TPrediction", bound=Prediction)
BoolOrInt = Union[bool, int]

_NAN_STRATEGIES = ("skip", "impute")


def _is_missing(y) -> bool:
    """An observation with no information: every component NaN (``filters/base.py:212``)."""
    return bool(torch.as_tensor(y).isnan().all())


class BaseFilter(Generic[TCorrection, TPrediction]):
    def __init__(
        self,
        model,
        record_states: BoolOrInt = False,
        record_moments: BoolOrInt = True,
        nan_strategy: str = "skip",
        record_intermediary_states: bool = False,
    ):
        """
        Args:
            model: a ``StateSpaceModel``, or a builder ``context -> StateSpaceModel`` resolved by ``initialize_model``.
            record_states / record_moments: how much history a ``FilterResult`` keeps (``False`` the latest entry only,
                ``True`` everything, an int that many).
            nan_strategy: ``"skip"`` (propagate only) or ``"impute"``.
            record_intermediary_states: also record the states of unobserved sub-steps.
        """
        is_model = isinstance(model, StateSpaceModel)
        if not is_model and not callable(model):
            raise ValueError("`model` must be a `StateSpaceModel` or a callable that returns one!")
        if nan_strategy not in _NAN_STRATEGIES:
            raise NotImplementedError(f"Currently cannot handle strategy '{nan_strategy}'!")
        self._model = model if is_model else None
        self._model_builder = (lambda _context, m=model: m) if is_model else model
        self._batch_shape = torch.Size([])
        self.record_states = record_states
        self.record_moments = record_moments
        self._nan_strategy = nan_strategy
        self._record_intermediary = record_intermediary_states

    # ---- model / shape ------------------------------------------------------------------------------------------
    @property
    def ssm(self) -> StateSpaceModel:
        return self._model

    def initialize_model(self, context):
        self._model = self._model_builder(context)

    @property
    def batch_shape(self) -> torch.Size:
        return self._batch_shape

    def set_batch_shape(self, batch_shape: torch.Size):
        """Number of parallel filters (``filters/base.py:93-119``): at most one batch dimension - it becomes the
        kernels' column index."""
        if len(batch_shape) > 1:
            raise NotImplementedError("Currently do not support nested batches!")
        self._batch_shape = torch.Size(batch_shape)

    # ---- the pieces a concrete filter provides ------------------------------------------------------------------
    def initialize(self) -> TCorrection:
        raise NotImplementedError()

    def predict(self, state: TCorrection) -> TPrediction:
        raise NotImplementedError()

    def correct(self, y: torch.Tensor, prediction: TPrediction) -> TCorrection:
        raise NotImplementedError()

    def copy(self) -> "BaseFilter":
        raise NotImplementedError()

    def smooth(self, states: Sequence[TCorrection], method: str) -> torch.Tensor:
        raise NotImplementedError()

    def _propagate_only(self, prediction: TPrediction) -> TCorrection:
        return prediction.create_state_from_prediction(self._model)

    # ---- drivers ------------------------------------------------------------------------------------------------
    def initialize_with_result(self, state: TCorrection = None) -> FilterResult[TCorrection]:
        return FilterResult(state if state is not None else self.initialize(), self.record_states, self.record_moments)

    def filter(self, y: torch.Tensor, correction: TCorrection, result: FilterResult = None) -> TCorrection:
        """Consumes one observation: the schedule's propagate-only moves, then the weighted move (a propagate-only one
        when ``y`` is missing).  Every state that enters ``result`` is appended as soon as it exists."""
        def record(state, wanted=True):
            if result is not None and wanted:
                result.append(state)
            return state

        prediction = self.predict(correction)
        now = int(prediction.get_timeseries_state().time_index)
        for _ in range(unobserved_moves_before(now, self._model.observe_every_step)):
            correction = record(self._propagate_only(prediction), self._record_intermediary)
            prediction = self.predict(correction)
        move = self._propagate_only if _is_missing(y) else (lambda p: self.correct(y, p))
        return record(move(prediction))

    def batch_filter(self, y: Iterable[torch.Tensor], bar=True, init_state: TCorrection = None) -> FilterResult[TCorrection]:
        """Filters the whole data set ``y (T, [O])`` one observation at a time (``filters/base.py:140-158``); particle
        filters on built-in models override this with the fused device loop."""
        state = init_state if init_state is not None else self.initialize()
        result = self.initialize_with_result(state)
        for y_t in (tqdm(y, desc=type(self).__name__) if bar else y):
            state = self.filter(y_t, state, result=result)
        return result
