import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture
def kernel_route(request, monkeypatch):
    """Which fused route a small filter (N <= 4096) takes: "column" - the column-persistent kernel, one launch per run
    (``pf_column.hpp``; the library's default for such shapes) - or "per_step" - one ``k_fused_step`` launch per time step
    (``PF_NO_COLUMN=1``; what larger filters always take).  Tests parametrised over it pin BOTH against the reference."""
    route = getattr(request, "param", "column")
    if route == "per_step":
        monkeypatch.setenv("PF_NO_COLUMN", "1")
    else:
        monkeypatch.delenv("PF_NO_COLUMN", raising=False)
    return route


both_routes = pytest.mark.parametrize("kernel_route", ["column", "per_step"], indirect=True)
