import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)')


@pytest.fixture(scope='session')
def oracle():
    """CPU restatement of the reference (oracle/cmvm_oracle.cc) -- the checker, never the thing under test."""
    from oracle.oracle import Oracle

    return Oracle('port')


@pytest.fixture(scope='session')
def model():
    """Sequential model of the GPU engine linked with the product's host logic (tests/model)."""
    from oracle.oracle import Oracle

    return Oracle('model')
