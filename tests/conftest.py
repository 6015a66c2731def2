import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Make sure the C-ABI library and the oracle are built (hipcc cross-compiles without a GPU)."""
    import __graft_entry__ as g
    so = os.path.join(ROOT, "t-route_amd", "libtrmc.so")
    osO = os.path.join(ROOT, "oracle", "libmc_oracle.so")
    if not (os.path.exists(so) and os.path.exists(osO)):
        g.build()
    yield
