"""
pyfilter_amd - an MI355X-native (gfx950) particle-filter inner loop behind pyfilter's own API.

Drop-in for the hot path of tingiskhan/pyfilter v0.29.0: ``filters.particle.{SISR, APF}``, the
``proposals.{Bootstrap, LinearGaussianObservations}`` API, ``resampling.{systematic, multinomial}``,
``utils.{normalize, get_ess}`` and ``batch_filter()``; the per-step propagate -> log-weight -> normalise -> resample
-> gather cycle runs in hand-written HIP kernels (``pyfilter_amd/csrc``) reached through the C ABI of ``libpfamd.so``
(``include/pf_amd.h``).  There is no CPU path: without a GPU or without the built library every call raises.
"""
__version__ = "0.1.0"

from torch.distributions import Distribution

from . import filters, resampling, timeseries, utils  # noqa: F401

# the reference switches argument validation off at import (pyfilter/__init__.py:8)
Distribution.set_default_validate_args(False)
