import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session', params=['golden_skyvis.npz', 'golden_skyvis_long.npz'], ids=['hera_scale', 'mwa_scale'])
def golden_skyvis(request):
    """The reference's own statements executed on seeded inputs (tests/golden/make_golden.py): 37 sources x 9 baselines (<= 150 m) x 24
    channels of 390 kHz at 150 MHz, and 101 sources x 21 baselines (<= 2.5 km, ~1600 cycles of phase) x 32 channels of 40 kHz at 185 MHz."""
    import numpy as NP
    return dict(NP.load(os.path.join(GOLDEN, request.param)))


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
