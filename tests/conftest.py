import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The simulated multi-rank GPU test carries its hand-off buffers in torch tensors.  torch bundles
    # its own HIP runtime, which must be the first one loaded into the process: import it before any
    # test loads libtrmc.so (linked against /opt/rocm) when GPU tests are selected.
    expr = config.getoption("-m") or ""
    if "gpu" in expr and "not gpu" not in expr:
        try:
            import torch  # noqa: F401
        except Exception:
            pass


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Make sure the C-ABI library and the oracle are built (hipcc cross-compiles without a GPU)."""
    import __graft_entry__ as g
    so = os.path.join(ROOT, "t-route_amd", "libtrmc.so")
    osO = os.path.join(ROOT, "oracle", "libmc_oracle.so")
    if not (os.path.exists(so) and os.path.exists(osO)):
        g.build()
    yield
