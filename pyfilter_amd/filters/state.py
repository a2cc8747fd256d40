"""Abstract filter states (``pyfilter/filters/state.py``)."""
from abc import ABC


class Prediction(ABC):
    def get_timeseries_state(self):
        raise NotImplementedError()

    def create_state_from_prediction(self, model):
        raise NotImplementedError()


class Correction(dict, ABC):
    def get_mean(self):
        raise NotImplementedError()

    def get_variance(self):
        raise NotImplementedError()

    def resample(self, indices):
        raise NotImplementedError()

    def get_loglikelihood(self):
        raise NotImplementedError()

    def exchange(self, other, mask):
        raise NotImplementedError()

    def get_timeseries_state(self):
        raise NotImplementedError()

    def state_dict(self):
        raise NotImplementedError()

    def load_state_dict(self, state_dict):
        raise NotImplementedError()
