import os
import sys

import pytest

from pyfilter_amd.hints import HINTS

# A driver test may start this suite in a subprocess with kernel-side choices in ITS environment (PF_TARGET_WGS: multi-round
# tiles at test sizes; PF_FORCE_SEARCH: the searching ancestor stage): the test infrastructure translates them - the package
# itself never reads the environment (pyfilter_amd/hints.py)
HINTS.apply_mapping(os.environ)
if os.environ.get("PF_AMD_LIB"):  # (an A/B build of the library under test: tools/build_variant.sh)
    from pyfilter_amd import _lib as _pf_lib

    _pf_lib.LIB_PATH = os.environ["PF_AMD_LIB"]

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_report_header(config):
    """Which binary the tests load, and whether it was built from this tree (``pf_version()`` carries the sources' sha256)."""
    try:
        import hashlib

        import __graft_entry__ as ge
        from pyfilter_amd import _lib

        with open(_lib.LIB_PATH, "rb") as f:
            digest = hashlib.sha256(f.read()).hexdigest()
        return [f"libpfamd.so sha256:{digest}", f"{_lib.version()}",
                "built from this tree's sources: " + ("yes" if ge.binary_matches_sources() else "NO - rebuild (__graft_entry__.build())")]
    except Exception as e:  # (no library yet: the tests that need it say so themselves)
        return [f"libpfamd.so: {e}"]


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture
def kernel_route(request, monkeypatch):
    """Which fused route a small filter (N <= 4096) takes: "column" - the column-persistent kernel, one launch per run
    (``pf_column.hpp``; the library's default for such shapes) - or "per_step" - one ``k_fused_step`` launch per time step
    (``HINTS.route = 1``; what larger filters always take).  Tests parametrised over it pin BOTH against the reference.
    Filters of 2 049 .. 16 384 particles (``oracle/cases.py: CLUSTER_CASES``): "cluster" - the column-cluster kernel
    (``pf_cluster.hpp``; ``HINTS.route = 4``, PF_ROUTE_CLUSTER_ALWAYS) - and "spread" - the same kernel with a filter's workgroups
    on different XCDs, i.e. its placement-independent exchange (``HINTS.route = 5``)."""
    route = getattr(request, "param", "column")
    monkeypatch.setattr(HINTS, "route", {"per_step": 1, "cluster": 4, "spread": 5}.get(route, 0))
    return route


both_routes = pytest.mark.parametrize("kernel_route", ["column", "per_step"], indirect=True)
cluster_routes = pytest.mark.parametrize("kernel_route", ["cluster", "spread", "per_step"], indirect=True)
