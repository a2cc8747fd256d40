from . import particle
from .base import BaseFilter
from .result import FilterResult
from .state import Correction

__all__ = ["particle", "BaseFilter", "FilterResult", "Correction"]
