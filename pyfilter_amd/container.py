"""Named tensor sequences with the reference's serialised form (``pyfilter/container.py``): a result object keeps series such
as ``filter_means`` either as fixed tuples or as (optionally length-bounded) deques; ``state_dict`` stacks each series into
one tensor under ``tensor_tuple__<name>`` / ``tensor_deque_<maxlen>__<name>`` (container.py:113-139) - the key scheme that
lets checkpoints travel between the two libraries.  Here the series live in ONE ordered mapping; whether an entry is a tuple
or a deque is the type of the stored sequence."""
from collections import OrderedDict, deque
from typing import Dict, Iterable, Sequence, Union

import torch

BoolOrInt = Union[int, bool]


def make_dequeue(maxlen: BoolOrInt = None) -> deque:
    """The reference's retention rule (container.py:10-18): ``False`` keeps the latest entry only, ``True`` / ``None``
    everything, an ``int`` that many."""
    if maxlen is False:
        return deque(maxlen=1)
    return deque(maxlen=None if (maxlen is None or maxlen is True) else int(maxlen))


def _wire_key(name: str, seq: Sequence) -> str:
    kind = f"deque_{seq.maxlen}" if isinstance(seq, deque) else "tuple"
    return f"tensor_{kind}__{name}"


class TensorContainer:
    def __init__(self):
        self._series: "OrderedDict[str, Sequence[torch.Tensor]]" = OrderedDict()

    # ---- building ------------------------------------------------------------------------------------------------------
    def make_tuple(self, name: str, values: Iterable[torch.Tensor] = None):
        self._series[name] = () if values is None else tuple(values)

    def make_deque(self, name: str, values: Iterable[torch.Tensor] = None, maxlen: BoolOrInt = None):
        self._series[name] = make_dequeue(maxlen)
        if values is not None:
            self._series[name].extend(values)

    # ---- reading (tuples first, then deques - the reference's iteration order) -------------------------------------------
    def _ordered(self):
        tuples = [(k, v) for k, v in self._series.items() if not isinstance(v, deque)]
        return tuples + [(k, v) for k, v in self._series.items() if isinstance(v, deque)]

    def __getitem__(self, name: str) -> Sequence[torch.Tensor]:
        try:
            return self._series[name]
        except KeyError:
            raise KeyError(f"Could not find '{name}'!") from None

    def __contains__(self, name: str) -> bool:
        return name in self._series

    def __len__(self) -> int:
        return len(self._series)

    def keys(self):
        return [k for k, _ in self._ordered()]

    def values(self):
        return [v for _, v in self._ordered()]

    def items(self):
        return self._ordered()

    def get_as_tensor(self, name: str) -> torch.Tensor:
        seq = self[name]
        return torch.stack(tuple(seq), dim=0) if len(seq) else torch.tensor([])

    # ---- the wire format -------------------------------------------------------------------------------------------------
    def state_dict(self) -> "OrderedDict[str, torch.Tensor]":
        return OrderedDict((_wire_key(name, seq), self.get_as_tensor(name)) for name, seq in self._ordered())

    def load_state_dict(self, state_dict: Dict[str, torch.Tensor]):
        """Takes its entries OUT of ``state_dict`` (as the reference does) and rebuilds the series from the stacked tensors."""
        mine = [k for k in state_dict if k.startswith(("tensor_tuple__", "tensor_deque_"))]
        for key in mine:
            kind, name = key[len("tensor_"):].split("__", 1)
            stacked = state_dict.pop(key)
            if kind == "tuple":
                self.make_tuple(name, stacked)
            else:
                bound = kind.partition("_")[2]
                self.make_deque(name, stacked, maxlen=None if bound == "None" else int(bound))
