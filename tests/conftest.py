import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built_libs():
    """Make sure the oracle exists (cheap gcc build); the HIP .so is built by __graft_entry__.build()."""
    from oracle import gs_oracle

    gs_oracle.build()
    lib = os.path.join(ROOT, "gscodec_studio_amd", "csrc", "libgsplat_hip.so")
    if not os.path.exists(lib):
        import __graft_entry__ as g

        g.build()
    yield
