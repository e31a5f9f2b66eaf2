import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """No test may sit on the GPU box for long: a wedged kernel or a stuck collective ends the process after 5 minutes (the whole
    CPU suite takes seconds, the whole GPU suite under a minute) instead of eating the round's GPU budget.  pytest-timeout's
    thread method, because a thread blocked inside the HIP runtime does not see signals."""
    try:
        import pytest_timeout  # noqa: F401
    except ImportError:
        return
    for item in items:
        if item.get_closest_marker("timeout") is None:
            item.add_marker(pytest.mark.timeout(300, method="thread"))


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
