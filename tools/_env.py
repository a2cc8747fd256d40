"""Development tools only: translates THIS process's environment into the package's explicit switches - the package itself
never reads the environment.  ``PF_AMD_LIB=<path>``: load an A/B or instrumented build of the library
(tools/build_variant.sh, tools/pmc_stages.py); ``PF_NO_COLUMN / PF_COLUMN_GENERIC / PF_COLUMN_MAX_N / PF_TARGET_WGS /
PF_FORCE_SEARCH / PF_NO_FUSED_STEP / PF_NO_FUSED_BATCH / PF_NO_GRAPH``: ``pyfilter_amd.hints.HINTS``."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def setup():
    from pyfilter_amd import _lib
    from pyfilter_amd.hints import HINTS

    if os.environ.get("PF_AMD_LIB"):
        _lib.LIB_PATH = os.environ["PF_AMD_LIB"]
    return HINTS.apply_mapping(os.environ)
