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
    """CPU restatement of the reference (test infrastructure)."""
    from oracle import pyoracle
    pyoracle.build()
    return pyoracle


@pytest.fixture(scope="session")
def engine():
    """The gfx950 engine through its C ABI; fails loudly if the HIP library is missing."""
    from poasta_amd import _lib
    _lib.lib()
    if _lib.lib().poa_device_count() <= 0:
        pytest.fail("no HIP device visible: -m gpu tests must run on the GPU box")
    from poasta_amd import aligner
    return aligner
