"""Stub of ``stochproc`` (v0.3.0 surface used by pyfilter's particle-filter hot path)."""
__version__ = "0.3.0-shim"
from . import timeseries  # noqa: F401
