import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)')


def pytest_collection_modifyitems(config, items):
    """A GPU test that hangs (a kernel that never ends blocks the process inside the HIP runtime) must end the session instead
    of holding the GPU box until somebody else's limit fires: 15 minutes per test, enforced from a watchdog thread
    (pytest-timeout's signal method cannot interrupt a blocked C call).  The longest GPU test takes seconds; the first one may
    include the in-tree rebuild of the library and the first `import torch` on a fresh box."""
    if not config.pluginmanager.hasplugin('timeout'):
        return
    for item in items:
        if item.get_closest_marker('gpu') and not item.get_closest_marker('timeout'):
            item.add_marker(pytest.mark.timeout(900, method='thread'))


@pytest.fixture(scope='session')
def oracle():
    """CPU restatement of the reference (oracle/cmvm_oracle.cc) -- the checker, never the thing under test."""
    from oracle.oracle import Oracle

    return Oracle('port')


@pytest.fixture(scope='session')
def reference_oracle():
    """The reference's OWN sources (oracle/_ref/libref.so: api.cc, cmvm_core.cc, state_opr.cc, indexers.cc, bit_decompose.cc,
    mat_decompose.cc compiled where they lie, oracle/Makefile) when the build is present -- it travels to the GPU box with the
    repository snapshot --, the restatement otherwise.  The GPU parity tests compare with THIS checker, live."""
    from oracle.oracle import HERE, Oracle

    return Oracle('ref' if (HERE / '_ref' / 'libref.so').exists() else 'port')


@pytest.fixture(scope='session')
def model():
    """Sequential model of the GPU engine linked with the product's host logic (tests/model)."""
    from oracle.oracle import Oracle

    return Oracle('model')
