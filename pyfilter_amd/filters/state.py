"""The two state interfaces every filter speaks (``pyfilter/filters/state.py``): a *prediction* (what ``predict`` hands to
``correct``) and a *correction* (a filter state: a ``dict`` of tensors with moments, log-likelihood, whole-filter moves and
a serialised form).  Implementations: ``filters/particle/state.py``."""
from abc import ABC


def _required(name: str):
    def method(self, *args, **kwargs):
        raise NotImplementedError(f"{type(self).__name__} does not implement {name}()")

    method.__name__ = name
    return method


class Prediction(ABC):
    pass


class Correction(dict, ABC):
    pass


for _name in ("get_timeseries_state", "create_state_from_prediction"):
    setattr(Prediction, _name, _required(_name))
for _name in ("get_mean", "get_variance", "get_loglikelihood", "get_timeseries_state", "resample", "exchange", "state_dict",
              "load_state_dict"):
    setattr(Correction, _name, _required(_name))
del _name
