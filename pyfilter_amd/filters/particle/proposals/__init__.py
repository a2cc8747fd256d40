from .base import Proposal
from .bootstrap import Bootstrap
from .linear import LinearGaussianObservations

__all__ = ["Proposal", "Bootstrap", "LinearGaussianObservations"]
