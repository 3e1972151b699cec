"""Host mirror of prisim/baseline_delay_horizon.py for the function on the sky-sum path.

``geometric_delay`` keeps the reference's signature, argument meaning and error types
(baseline_delay_horizon.py:133-241).  It is a host utility (one small matrix product) for callers that want the
delay matrix itself; the sky-sum kernels never call it -- they form tau[s,b] = (b . s)/c per (source, baseline)
in registers and never materialise the matrix.
"""
import numpy as NP

from . import geometry as GEOM

C_LIGHT = 299792458.0   # scipy.constants.c (baseline_delay_horizon.py:236)


def _baselines_nx3(baselines):
    """Coerce to (nbl, 3): missing components are zero-filled, extra columns dropped (:195-203)."""
    if not isinstance(baselines, NP.ndarray):
        raise TypeError('baselines should be a Nx3 numpy array in geometric_delay().')
    bl = NP.atleast_2d(NP.asarray(baselines, dtype=NP.float64))
    ncol = bl.shape[1]
    if ncol < 3:
        bl = NP.concatenate((bl, NP.zeros((bl.shape[0], 3 - ncol))), axis=1)
    return bl[:, :3]


def _positions(skypos, ncol, what):
    """Coerce sky positions to (nsrc, ncol), a single position may be given as a flat vector (:205-231)."""
    pos = NP.asarray(skypos, dtype=NP.float64)
    if pos.ndim == 1:
        if pos.size != ncol:
            raise ValueError('Sky position in {0} should consist of {1} elements.'.format(what, ncol))
        pos = pos[NP.newaxis, :]
    if pos.ndim != 2 or pos.shape[1] != ncol:
        raise ValueError('Sky positions should be a Nx{0} numpy array if using {1}.'.format(ncol, what))
    return pos


def geometric_delay(baselines, skypos, altaz=False, dircos=False, hadec=True, units='mks', latitude=None):
    """Geometric delays, shape (nsrc, nbl), of `baselines` (nbl x 3, metres or cm per `units`) toward `skypos` given as
    Alt-Az degrees (altaz=True), HA-Dec degrees (hadec=True, needs latitude) or ENU direction cosines (dircos=True)."""
    if int(bool(altaz)) + int(bool(dircos)) + int(bool(hadec)) != 1:
        raise ValueError('One and only one of altaz, dircos, hadec must be set to True.')
    if hadec and latitude is None:
        raise ValueError('Latitude must be specified when skypos is in HA-Dec format.')
    bl = _baselines_nx3(baselines)
    if dircos:
        dc = _positions(skypos, 3, 'direction cosines')
    else:
        angles = _positions(skypos, 2, 'altitude-azimuth or HA-Dec')
        if hadec:
            angles = GEOM.hadec2altaz(angles, latitude, 'degrees')               # :220
        dc = GEOM.altaz2dircos(angles, 'degrees')                                # :218
    speed = C_LIGHT * (100.0 if units == 'cgs' else 1.0)                         # :236-237
    return dc.dot(bl.T) / speed                                                  # :240
