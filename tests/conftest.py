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
    """The CPU oracle (test infrastructure).  Built here on demand; prebuilt on the GPU box."""
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def mirror():
    from oracle import numpy_mirror as M
    return M
