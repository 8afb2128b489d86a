import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.load()
    return orc


@pytest.fixture(scope="session")
def device():
    """One device object for the whole GPU session (the reference keeps one context singleton)."""
    from renderer_amd import api
    dev = api.Device(0)
    yield dev
    dev.destroy()
