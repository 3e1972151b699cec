"""Host mirror of prisim/baseline_delay_horizon.py for the functions on the sky-sum path.

``geometric_delay`` keeps the reference's signature, argument checks and error types
(baseline_delay_horizon.py:133-241).  It is a host utility (one small matrix product) for callers
that want the delay matrix itself; the sky-sum kernels never call it -- they form
tau[s,b] = (b . s)/c per (source, baseline) in registers and do not materialise the matrix.
"""
import numpy as NP

from . import geometry as GEOM

C_LIGHT = 299792458.0   # scipy.constants.c (:236)


def geometric_delay(baselines, skypos, altaz=False, dircos=False, hadec=True, units='mks', latitude=None):
    """Geometric delays (nsrc x nbl) for baselines (nbl x 3) and sky positions (:133-241)."""
    if (altaz) + (dircos) + (hadec) != 1:
        raise ValueError('One and only one of altaz, dircos, hadec must be set to True.')
    if hadec and (latitude is None):
        raise ValueError('Latitude must be specified when skypos is in HA-Dec format.')
    if units not in ('mks', 'cgs'):
        units = 'mks'
    if not isinstance(baselines, NP.ndarray):
        raise TypeError('baselines should be a Nx3 numpy array in geometric_delay().')
    if baselines.ndim == 1:
        baselines = baselines.reshape(1, -1)
    if baselines.shape[1] == 1:
        baselines = NP.hstack((baselines, NP.zeros((baselines.shape[0], 2))))
    elif baselines.shape[1] == 2:
        baselines = NP.hstack((baselines, NP.zeros((baselines.shape[0], 1))))
    elif baselines.shape[1] > 3:
        baselines = baselines[:, :3]
    skypos = NP.asarray(skypos, dtype=NP.float64)
    if altaz or hadec:
        if skypos.ndim < 2:
            if skypos.size != 2:
                raise ValueError('Sky position in altitude-azimuth or HA-Dec should consist of 2 elements.')
            skypos = skypos.reshape(1, -1)
        elif skypos.ndim > 2 or skypos.shape[1] != 2:
            raise ValueError('Sky positions should be a Nx2 numpy array if using altitude-azimuth of HA-Dec.')
        if altaz:
            dc = GEOM.altaz2dircos(skypos, 'degrees')
        else:
            dc = GEOM.altaz2dircos(GEOM.hadec2altaz(skypos, latitude, 'degrees'), 'degrees')
    else:
        if skypos.ndim < 2:
            if skypos.size != 3:
                raise ValueError('Sky position in direction cosines should consist of 3 elements.')
            skypos = skypos.reshape(1, -1)
        elif skypos.ndim > 2 or skypos.shape[1] != 3:
            raise ValueError('Sky positions should be a Nx3 numpy array if using direction cosines.')
        dc = skypos
    c = C_LIGHT if units == 'mks' else C_LIGHT * 1e2
    return NP.dot(dc, baselines.T) / c
