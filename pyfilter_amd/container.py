"""
Result containers mirroring ``pyfilter/container.py`` + ``pyfilter/state.py``: named, optionally length-bounded
sequences of tensors that serialise as stacked tensors.  The ``state_dict`` key scheme
(``tensor_tuple__<name>`` / ``tensor_deque_<maxlen>__<name>``, container.py:113-139) is kept so checkpoints interoperate.
"""
from collections import OrderedDict, deque
from typing import Any, Deque, Dict, Iterable, Tuple, Union

import torch

BoolOrInt = Union[int, bool]


def make_dequeue(maxlen: BoolOrInt = None) -> deque:
    """``False`` -> keep only the latest entry, ``True``/``None`` -> unbounded, ``int`` -> that many (container.py:10-18)."""
    if maxlen is False:
        return deque(maxlen=1)
    if maxlen is None or isinstance(maxlen, bool):
        return deque()
    return deque(maxlen=int(maxlen))


class TensorContainer:
    _KEY = "tensor_{kind}__{name}"

    def __init__(self):
        self._tuples: Dict[str, Tuple[torch.Tensor, ...]] = OrderedDict()
        self._deques: Dict[str, Deque[torch.Tensor]] = OrderedDict()

    def make_tuple(self, name: str, values=None):
        self._tuples[name] = tuple(values) if values is not None else tuple()

    def make_deque(self, name: str, values=None, maxlen: BoolOrInt = None):
        dq = self._deques[name] = make_dequeue(maxlen)
        if values is not None:
            dq.extend(values)

    def __getitem__(self, key: str) -> Iterable[torch.Tensor]:
        if key in self._tuples:
            return self._tuples[key]
        if key in self._deques:
            return self._deques[key]
        raise KeyError(f"Could not find '{key}'!")

    def get_as_tensor(self, key: str) -> torch.Tensor:
        items = self[key]
        return torch.stack(tuple(items), dim=0) if len(items) else torch.tensor([])

    def __contains__(self, key):
        return key in self._tuples or key in self._deques

    def __len__(self):
        return len(self._tuples) + len(self._deques)

    def keys(self):
        return list(self._tuples.keys()) + list(self._deques.keys())

    def values(self):
        return list(self._tuples.values()) + list(self._deques.values())

    def items(self):
        return list(zip(self.keys(), self.values()))

    def state_dict(self) -> "OrderedDict[str, torch.Tensor]":
        out = OrderedDict()
        for name in self._tuples:
            out[self._KEY.format(kind="tuple", name=name)] = self.get_as_tensor(name)
        for name, dq in self._deques.items():
            out[self._KEY.format(kind=f"deque_{dq.maxlen}", name=name)] = self.get_as_tensor(name)
        return out

    def load_state_dict(self, state_dict: Dict[str, Any]):
        for key in [k for k in state_dict if k.startswith("tensor_tuple__") or k.startswith("tensor_deque_")]:
            value = state_dict.pop(key)
            kind, name = key[len("tensor_"):].split("__", 1)
            if kind == "tuple":
                self.make_tuple(name, value)
            else:
                maxlen = kind.split("_", 1)[1]
                self.make_deque(name, value, maxlen=None if maxlen == "None" else int(maxlen))


class BaseResult(dict):
    """Base class for result objects (pyfilter/state.py:8-48)."""

    def __init__(self):
        super().__init__()
        self.tensor_tuples = TensorContainer()

    def state_dict(self):
        return OrderedDict({"tensor_tuples": self.tensor_tuples.state_dict()})

    def load_state_dict(self, state_dict):
        self.tensor_tuples.load_state_dict(state_dict["tensor_tuples"])
