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
    """The product binding.  The library is normally there already (__graft_entry__.build()); on a box that only
    received the sources it is compiled once here (hipcc cross-compiles gfx950 without a GPU)."""
    from verticut_amd import build as vb
    from verticut_amd import engine
    if not os.path.exists(engine.LIB_PATH) or not os.path.exists(vb.DRIVER):
        vb.build(force=True)
    engine.load_library()
    return engine
