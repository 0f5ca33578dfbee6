import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def pkg():
    import _pkg
    return _pkg.load()


@pytest.fixture(scope='session')
def golden_dir():
    return os.path.join(ROOT, 'tests', 'golden')


@pytest.fixture(scope='session')
def built_lib():
    """The C-ABI library, built if absent (hipcc cross-compiles on CPU-only boxes)."""
    so = os.path.join(ROOT, 'efficient-nerf_amd', 'libr2l_hip.so')
    if not os.path.exists(so):
        import __graft_entry__ as g
        g.build()
    return so
