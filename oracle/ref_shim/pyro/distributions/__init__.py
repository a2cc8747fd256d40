"""``pyro.distributions`` stub: pyro's Normal/MVN/... subclass the torch ones with identical arithmetic."""
import torch
from torch.distributions import *  # noqa: F401,F403
from torch.distributions import (  # noqa: F401
    Categorical,
    Distribution,
    Independent,
    MultivariateNormal,
    Normal,
    TransformedDistribution,
)

from . import transforms  # noqa: F401


def _to_event(self, n=None):
    """pyro adds ``.to_event`` on every distribution; same as ``Independent``."""
    if n is None:
        n = len(self.batch_shape)
    return Independent(self, n) if n > 0 else self


Distribution.to_event = _to_event
