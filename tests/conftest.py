import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
# The whole suite runs with the gradient-plan validator on (read once when the library is loaded): every gradient access of every reverse
# program is checked against the live interval the gradient slab was packed by (engine_exec.cpp: grad_access_check).
os.environ.setdefault("DD_GRAD_CHECK", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip_lib():
    """The HIP library must be present and loadable for every -m gpu test: no silent fallbacks."""
    from distdiff_amd import _lib
    return _lib.lib()
