import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle (test infrastructure).  Built on demand; never imported by the product."""
    from oracle import vc_oracle as vo
    vo.lib()
    return vo


@pytest.fixture(scope="session")
def vc():
    """The product binding; the library must exist (built by __graft_entry__.build())."""
    from verticut_amd import engine
    engine.load_library()
    return engine
