from . import proposals
from .apf import APF
from .base import ParticleFilter
from .sisr import SISR

__all__ = ["proposals", "APF", "SISR", "ParticleFilter"]
