import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _oracle_threads():
    """The C oracle uses OpenMP: keep its team within the cores this process may run on (a GPU box shows every
    hardware thread of the host but grants 16), or tiny sweeps spend their time waking an oversubscribed team."""
    try:
        from oracle import c_oracle as CO
        try:
            avail = len(os.sched_getaffinity(0))
        except AttributeError:
            avail = os.cpu_count() or 1
        CO.set_threads(max(1, min(avail, 16)))
    except Exception:
        pass                                      # the oracle library is not built yet: the tests that need it say so
    yield
