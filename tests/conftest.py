import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "av1-go_amd"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def O():
    """the CPU oracle (test infrastructure)."""
    from oracle import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def av1mi():
    import av1mi as m
    return m


@pytest.fixture(scope="session")
def ctx(av1mi):
    """one HIP context on device 0; fails loudly when the HIP library or the GPU is missing."""
    c = av1mi.Context(0)
    yield c
    c.close()
