import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_skyvis():
    import numpy as NP
    return dict(NP.load(os.path.join(GOLDEN, 'golden_skyvis.npz')))


@pytest.fixture(scope='session')
def golden_beams():
    import numpy as NP
    return dict(NP.load(os.path.join(GOLDEN, 'golden_beams.npz')))


@pytest.fixture(scope='module')
def ctx():
    """A GPU context through the C-ABI.  Fails loudly (no fallback) when the library or a GPU is missing."""
    from prisim_amd import _abi
    c = _abi.Context(0)
    yield c
    c.close()
