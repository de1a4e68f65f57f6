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
    from oracle import mw_oracle
    mw_oracle.lib()
    return mw_oracle


@pytest.fixture(scope="session")
def mw():
    """The product package with the HIP library loaded.  Fails loudly if the .so is missing."""
    import miniweatherml_amd
    from miniweatherml_amd import capi
    capi.lib()
    return miniweatherml_amd
