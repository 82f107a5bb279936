import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def swamd():
    """The host-side package (its directory name has a hyphen, so import it by string)."""
    return importlib.import_module("smith-waterman_amd")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    return oracle_lib.Oracle()


@pytest.fixture(scope="session")
def engine(swamd):
    eng = swamd.Engine(0)
    yield eng
    eng.close()
