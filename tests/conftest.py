import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Build libmipt.so and the oracle once per session (hipcc cross-compiles without a GPU)."""
    import __graft_entry__
    __graft_entry__.build()
    return True


@pytest.fixture(scope="session")
def orc(built):
    from oracle import orc as _orc
    _orc.load()
    return _orc


@pytest.fixture(scope="session")
def rrt(built):
    import rust_ray_tracing_amd as _rrt
    _rrt.load()
    return _rrt
