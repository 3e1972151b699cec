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


def body_class_sample(bl, ch, dircos, pc, f32=True, kappa=None):
    """Baselines for a full-size parity check, by CLASS OF KERNEL BODY instead of an even sprinkle: three baselines (first,
    middle, last) of -- the first, a middle and the last LIFTING group of 256 (groups whose step angle max|b| max|s - s_pc| |df| / c stays below
    1/8 cycle in fp32, 1/4 in fp64: the three-shear rotation), the first, a middle and the last NON-lifting group (plain rotation; for the
    taper kernels the re-anchored bodies), the first group whose leading sources the taper culling can skip (kappa given: its smallest
    horizontal baseline resolves the pixels out), and the ragged last group.  Returns (indices, lifting-flag per group)."""
    import numpy as NP
    nbl = bl.shape[0]
    ng = (nbl + 255) // 256
    length = NP.sqrt(NP.sum(bl ** 2, axis=1))
    gmax = NP.array([length[g * 256:(g + 1) * 256].max() for g in range(ng)])
    dmax = float(NP.sqrt(NP.sum((NP.asarray(dircos) - NP.asarray(pc)[None, :]) ** 2, axis=1)).max())
    k = dmax * abs(float(ch[1] - ch[0])) / 299792458.0
    lift = gmax * k <= (0.125 if f32 else 0.25) * (1.0 - 1e-9)
    groups = {0, ng - 1}
    for idx in (NP.flatnonzero(lift), NP.flatnonzero(~lift)):
        if idx.size:
            groups |= {int(idx[0]), int(idx[idx.size // 2]), int(idx[-1])}
    if kappa is not None:
        hor = NP.sqrt(bl[:, 0] ** 2 + bl[:, 1] ** 2)
        hmin = NP.array([hor[g * 256:(g + 1) * 256].min() for g in range(ng)])
        fmin = float(min(abs(ch[0]), abs(ch[-1])))
        cand = NP.flatnonzero(kappa * (hmin * fmin / 299792458.0) ** 2 >= (18.0 if f32 else 28.0))
        if cand.size:
            groups |= {int(cand[0]), int(cand[cand.size // 2])}
    sel = []
    for g in sorted(groups):
        lo, hi = g * 256, min(nbl, g * 256 + 256)
        sel += [lo, (lo + hi) // 2, hi - 1]
    return NP.unique(NP.asarray(sel, dtype=NP.int64)), lift
