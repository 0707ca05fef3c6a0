import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _build_once():
    import __graft_entry__ as g
    g.build()


@pytest.fixture(scope="session", autouse=True)
def built_libraries():
    """Compile liboracle.so and libpvq.so if missing (hipcc cross-compiles without a GPU)."""
    _build_once()
    yield
