"""``BaseFilter`` (``pyfilter/filters/base.py:17-232``): the host-side driver loop around predict / correct."""
from typing import Callable, Generic, Sequence, TypeVar, Union

import torch
from tqdm import tqdm

from ..timeseries import StateSpaceModel
from .result import FilterResult
from .state import Correction, Prediction

TCorrection = TypeVar("TCorrection", bound=Correction)
TPrediction = TypeVar("TPrediction", bound=Prediction)
BoolOrInt = Union[bool, int]


class BaseFilter(Generic[TCorrection, TPrediction]):
    def __init__(
        self,
        model,
        record_states: BoolOrInt = False,
        record_moments: BoolOrInt = True,
        nan_strategy: str = "skip",
        record_intermediary_states: bool = False,
    ):
        super().__init__()
        if not (isinstance(model, StateSpaceModel) or callable(model)):
            raise ValueError("`model` must be a `StateSpaceModel` or a callable that returns one!")

        if callable(model) and not isinstance(model, StateSpaceModel):
            self._model_builder, self._model = model, None
        else:
            self._model_builder, self._model = (lambda _: model), model

        self._batch_shape = torch.Size([])
        self.record_states = record_states
        self.record_moments = record_moments
        if nan_strategy not in ["skip", "impute"]:
            raise NotImplementedError(f"Currently cannot handle strategy '{nan_strategy}'!")
        self._nan_strategy = nan_strategy
        self._record_intermediary = record_intermediary_states

    @property
    def ssm(self) -> StateSpaceModel:
        return self._model

    def initialize_model(self, context):
        self._model = self._model_builder(context)

    @property
    def batch_shape(self) -> torch.Size:
        return self._batch_shape

    def set_batch_shape(self, batch_shape: torch.Size):
        """Number of parallel filters (``filters/base.py:93-119``); at most one batch dimension."""
        if len(batch_shape) > 1:
            raise NotImplementedError("Currently do not support nested batches!")
        self._batch_shape = torch.Size(batch_shape)

    def initialize(self) -> TCorrection:
        raise NotImplementedError()

    def initialize_with_result(self, state: TCorrection = None) -> FilterResult[TCorrection]:
        return FilterResult(state or self.initialize(), self.record_states, self.record_moments)

    def batch_filter(self, y: Sequence[torch.Tensor], bar=True, init_state: TCorrection = None) -> FilterResult[TCorrection]:
        """Filters the whole data set ``y (T, [O])`` (``filters/base.py:140-158``)."""
        state = init_state or self.initialize()
        result = self.initialize_with_result(state)
        for y_t in y if not bar else tqdm(y, desc=str(self.__class__.__name__)):
            state = self.filter(y_t, state, result=result)
        return result

    def copy(self) -> "BaseFilter":
        raise NotImplementedError()

    def predict(self, state: TCorrection) -> TPrediction:
        raise NotImplementedError()

    def correct(self, y: torch.Tensor, prediction: TPrediction) -> TCorrection:
        raise NotImplementedError()

    def _propagate_only(self, prediction: TPrediction) -> TCorrection:
        return prediction.create_state_from_prediction(self._model)

    def filter(self, y: torch.Tensor, correction: TCorrection, result: FilterResult = None) -> TCorrection:
        """One filter move (``filters/base.py:188-221``): predict, propagate through unobserved sub-steps, then
        correct - or only propagate when the observation is all NaN."""
        prediction = self.predict(correction)
        while prediction.get_timeseries_state().time_index % self._model.observe_every_step != 0:
            correction = self._propagate_only(prediction)
            if result is not None and self._record_intermediary:
                result.append(correction)
            prediction = self.predict(correction)

        if y.isnan().all():
            correction = self._propagate_only(prediction)
        else:
            correction = self.correct(y, prediction)

        if result is not None:
            result.append(correction)
        return correction

    def smooth(self, states: Sequence[TCorrection], method: str) -> torch.Tensor:
        raise NotImplementedError("smoothing is offline post-processing, out of the hot path's scope (SURVEY.md §2 row 5)")
