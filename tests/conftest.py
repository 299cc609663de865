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
    """CPU oracle (test infrastructure) -- builds oracle/libzkm_oracle.so with gcc if needed."""
    from oracle.oracle_py import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def zkm():
    import zkm_amd
    zkm_amd.load()
    return zkm_amd


@pytest.fixture(scope="session")
def ctx(zkm):
    """GPU context; fails loudly (no CPU fallback) if the extension or the GPU is missing."""
    c = zkm.Context(0)
    yield c
    c.close()
