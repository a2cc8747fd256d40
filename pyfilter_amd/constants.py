"""Numeric constants of the default floating-point type, evaluated at import like ``pyfilter.constants`` does
(so they follow ``torch.set_default_dtype`` only if it is called before the import)."""
import math

import torch


def _default_finfo() -> torch.finfo:
    return torch.finfo(torch.get_default_dtype())


_fi = _default_finfo()

#: positive infinity (log-weights of impossible particles are ``-INFTY``)
INFTY = math.inf
#: machine epsilon of the default dtype and its square root
EPS2 = float(_fi.eps)
EPS = math.sqrt(EPS2)
#: largest finite value of the default dtype (``normalize`` maps ``-inf`` to ``-MAX``, see ``utils.normalize``)
MAX = float(_fi.max)
