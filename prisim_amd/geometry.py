"""Coordinate helpers on the host (numpy) -- the pieces of ``astroutils.geometry`` the sky-sum
path calls; astroutils itself is an un-vendored, un-pinned dependency of the
reference, so the conventions are taken from the reference's own in-tree statements:

  altaz [0, 90]   == dircos [1, 0, 0] (East)    prisim/primary_beams.py:255-258
  altaz [90, 270] == dircos [0, 0, 1] (zenith)  prisim/primary_beams.py:275-278
  dircos aligned with local East, North, Up       prisim/primary_beams.py:122-123
  alt is column 0, az column 1, degrees           prisim/primary_beams.py:579
"""
import numpy as NP


def altaz2dircos(altaz, units='degrees'):
    """(alt, az) -> (l, m, n) ENU direction cosines; az measured from North through East.
    Call sites replaced: GEOM.altaz2dircos at baseline_delay_horizon.py:218, interferometry.py:6164, 6263."""
    altaz = NP.asarray(altaz, dtype=NP.float64)
    if altaz.ndim == 1:
        if altaz.size != 2:
            raise ValueError('altaz must have 2 elements (alt, az)')
        altaz = altaz.reshape(1, 2)
    if altaz.ndim != 2 or altaz.shape[1] != 2:
        raise ValueError('altaz must be an N x 2 array')
    if units == 'degrees':
        alt, az = NP.radians(altaz[:, 0]), NP.radians(altaz[:, 1])
    elif units == 'radians':
        alt, az = altaz[:, 0], altaz[:, 1]
    else:
        raise ValueError('units must be "degrees" or "radians"')
    if NP.any(NP.abs(alt) > NP.pi / 2 + 1e-12):
        raise ValueError('altitude out of range')
    ca = NP.cos(alt)
    return NP.stack((ca * NP.sin(az), ca * NP.cos(az), NP.sin(alt)), axis=1)


def dircos2altaz(dircos, units='degrees'):
    """(l, m, n) -> (alt, az); inverse of altaz2dircos (GEOM.dircos2altaz, primary_beams.py:598)."""
    dc = NP.asarray(dircos, dtype=NP.float64)
    if dc.ndim == 1:
        dc = dc.reshape(1, -1)
    if dc.shape[1] != 3:
        raise ValueError('dircos must be an N x 3 array')
    alt = NP.arcsin(NP.clip(dc[:, 2], -1.0, 1.0))
    az = NP.arctan2(dc[:, 0], dc[:, 1])
    az = NP.where(az < 0.0, az + 2 * NP.pi, az)
    out = NP.stack((alt, az), axis=1)
    return NP.degrees(out) if units == 'degrees' else out


def hadec2altaz(hadec, latitude, units='degrees'):
    """(HA, Dec) -> (alt, az) at the given latitude (GEOM.hadec2altaz, interferometry.py:6157)."""
    hadec = NP.asarray(hadec, dtype=NP.float64)
    squeeze = hadec.ndim == 1
    hadec = hadec.reshape(-1, 2)
    if units == 'degrees':
        ha, dec, lat = NP.radians(hadec[:, 0]), NP.radians(hadec[:, 1]), NP.radians(latitude)
    else:
        ha, dec, lat = hadec[:, 0], hadec[:, 1], latitude
    sin_alt = NP.sin(dec) * NP.sin(lat) + NP.cos(dec) * NP.cos(lat) * NP.cos(ha)
    alt = NP.arcsin(NP.clip(sin_alt, -1.0, 1.0))
    east = -NP.cos(dec) * NP.sin(ha)
    north = NP.sin(dec) * NP.cos(lat) - NP.cos(dec) * NP.sin(lat) * NP.cos(ha)
    az = NP.arctan2(east, north)
    az = NP.where(az < 0.0, az + 2 * NP.pi, az)
    out = NP.stack((alt, az), axis=1)
    if units == 'degrees':
        out = NP.degrees(out)
    return out[0] if squeeze else out


def altaz2hadec(altaz, latitude, units='degrees'):
    """(alt, az) -> (HA, Dec) (GEOM.altaz2hadec, interferometry.py:6122)."""
    altaz = NP.asarray(altaz, dtype=NP.float64)
    squeeze = altaz.ndim == 1
    altaz = altaz.reshape(-1, 2)
    if units == 'degrees':
        alt, az, lat = NP.radians(altaz[:, 0]), NP.radians(altaz[:, 1]), NP.radians(latitude)
    else:
        alt, az, lat = altaz[:, 0], altaz[:, 1], latitude
    sin_dec = NP.sin(alt) * NP.sin(lat) + NP.cos(alt) * NP.cos(lat) * NP.cos(az)
    dec = NP.arcsin(NP.clip(sin_dec, -1.0, 1.0))
    y = -NP.cos(alt) * NP.sin(az)
    x = NP.sin(alt) * NP.cos(lat) - NP.cos(alt) * NP.sin(lat) * NP.cos(az)
    ha = NP.arctan2(y, x)
    out = NP.stack((ha, dec), axis=1)
    if units == 'degrees':
        out = NP.degrees(out)
    return out[0] if squeeze else out


def catalog_unitvec(location, coords):
    """Unit vectors (nsrc, 3) of a catalogue in ITS OWN frame -- what prisim_catalog.unitvec carries (include/prisim_hip.h):
    'radec' (cos d cos a, cos d sin a, sin d);  'hadec' the same with the hour angle for a;  'altaz' East-North-Up direction cosines
    (altaz2dircos).  The snapshot's frame (prisim_amd/frames.py snapshot_frame) takes them to the local East-North-Up frame."""
    loc = NP.asarray(location, dtype=NP.float64).reshape(-1, 2)
    if coords == 'altaz':
        return altaz2dircos(loc, 'degrees')
    if coords not in ('radec', 'hadec'):
        raise ValueError('coords must be "radec", "hadec" or "altaz"')
    lon, lat = NP.radians(loc[:, 0]), NP.radians(loc[:, 1])
    cd = NP.cos(lat)
    return NP.stack((cd * NP.cos(lon), cd * NP.sin(lon), NP.sin(lat)), axis=1)


def frame_dircos(unitvec, rot, beta):
    """East-North-Up direction cosines s = normalise(R (u + beta)) of catalogue unit vectors u (nsrc, 3) for one snapshot's frame
    (R (3, 3), beta (3,)) -- the host statement of cat_source() in prisim_amd/csrc/catalog_kernels.hip.  Written operation by operation
    (no dot products, whose BLAS kernels may fuse multiply-adds) in the order the device uses with contraction off: +, *, sqrt and /
    are correctly rounded on both sides, so host and device agree to the last bit and select the same region of interest."""
    u = NP.asarray(unitvec, dtype=NP.float64).reshape(-1, 3)
    r = NP.asarray(rot, dtype=NP.float64).reshape(3, 3)
    b = NP.asarray(beta, dtype=NP.float64).reshape(3)
    t0, t1, t2 = u[:, 0] + b[0], u[:, 1] + b[1], u[:, 2] + b[2]
    v0 = (r[0, 0] * t0 + r[0, 1] * t1) + r[0, 2] * t2
    v1 = (r[1, 0] * t0 + r[1, 1] * t1) + r[1, 2] * t2
    v2 = (r[2, 0] * t0 + r[2, 1] * t1) + r[2, 2] * t2
    nrm = NP.sqrt((v0 * v0 + v1 * v1) + v2 * v2)
    return NP.stack((v0 / nrm, v1 / nrm, v2 / nrm), axis=1)


def roi_thresholds(roi_radius_deg):
    """(sin(90 - roi_radius), cos(roi_radius)): the region-of-interest tests on direction cosines -- 'zenith' keeps n >= the first
    (altitude >= 90 - roi_radius, interferometry.py:6216), 'pointing_center' keeps s . s_pc >= the second (angle <= roi_radius, :6211).
    math.sin / math.cos (the C library's, as libprisim_hip.so's host code uses) so that both sides hold the same two numbers."""
    import math
    return math.sin(math.radians(90.0 - float(roi_radius_deg))), math.cos(math.radians(float(roi_radius_deg)))


def roi_select(dircos, roi_center, roi_radius_deg, pc_dircos=None):
    """Indices of the sources inside the region of interest (interferometry.py:6204-6216) from their direction cosines."""
    sin_alt_min, cos_radius = roi_thresholds(roi_radius_deg)
    dc = NP.asarray(dircos, dtype=NP.float64).reshape(-1, 3)
    if roi_center == 'pointing_center':
        pc = NP.asarray(pc_dircos, dtype=NP.float64).reshape(3)
        cosd = (dc[:, 0] * pc[0] + dc[:, 1] * pc[1]) + dc[:, 2] * pc[2]
        return NP.where(cosd >= cos_radius)[0]
    return NP.where(dc[:, 2] >= sin_alt_min)[0]


# ---- HEALPix (RING) pixel centres: healpy is not available, so this is a from-the-paper
# implementation (Gorski et al. 2005, eqs. 2-9) of pix2ang for the RING scheme. ---------------

def nside2npix(nside):
    return 12 * int(nside) ** 2


def nside2resol(nside, arcmin=False):
    """sqrt(pixel area) in radians (healpy.nside2resol); run_prisim.py uses it as the pixel FWHM."""
    res = NP.sqrt(4 * NP.pi / nside2npix(nside))
    return NP.degrees(res) * 60.0 if arcmin else res


def healpix_pix2ang_ring(nside, ipix=None):
    """(theta, phi) of RING-ordered pixel centres; theta = colatitude in [0, pi], phi in [0, 2 pi)."""
    nside = int(nside)
    if nside < 1:
        raise ValueError('nside must be >= 1')
    npix = nside2npix(nside)
    p = NP.arange(npix, dtype=NP.int64) if ipix is None else NP.asarray(ipix, dtype=NP.int64)
    if NP.any((p < 0) | (p >= npix)):
        raise ValueError('pixel index out of range')
    ncap = 2 * nside * (nside - 1)
    z = NP.empty(p.shape, dtype=NP.float64)
    phi = NP.empty(p.shape, dtype=NP.float64)
    north = p < ncap
    south = p >= npix - ncap
    equat = ~(north | south)
    if NP.any(north):
        pn = p[north]
        iring = (1 + NP.floor(NP.sqrt(1.0 + 2.0 * pn)).astype(NP.int64)) // 2
        # guard against sqrt rounding
        iring = NP.where(2 * iring * (iring - 1) > pn, iring - 1, iring)
        iring = NP.where(2 * (iring + 1) * iring <= pn, iring + 1, iring)
        iphi = pn + 1 - 2 * iring * (iring - 1)
        z[north] = 1.0 - iring.astype(float) ** 2 / (3.0 * nside ** 2)
        phi[north] = (iphi - 0.5) * NP.pi / (2.0 * iring)
    if NP.any(equat):
        ip = p[equat] - ncap
        iring = ip // (4 * nside) + nside
        iphi = ip % (4 * nside) + 1
        fodd = 0.5 * (1 + ((iring + nside) & 1))
        z[equat] = (2 * nside - iring) * 2.0 / (3.0 * nside)
        phi[equat] = (iphi - fodd) * NP.pi / (2.0 * nside)
    if NP.any(south):
        ip = npix - p[south]
        iring = (1 + NP.floor(NP.sqrt(2.0 * ip - 1.0)).astype(NP.int64)) // 2
        iring = NP.where(2 * iring * (iring - 1) >= ip, iring - 1, iring)
        iring = NP.where(2 * (iring + 1) * iring < ip, iring + 1, iring)
        iphi = 4 * iring + 1 - (ip - 2 * iring * (iring - 1))
        z[south] = -1.0 + iring.astype(float) ** 2 / (3.0 * nside ** 2)
        phi[south] = (iphi - 0.5) * NP.pi / (2.0 * iring)
    return NP.arccos(NP.clip(z, -1.0, 1.0)), phi


def enu2xyz(enu, latitude, units='degrees'):
    """Local East-North-Up vectors -> equatorial (X towards hour angle 0 on the celestial equator, Y towards hour angle -6 h (East),
    Z towards the celestial pole) at the given latitude: x = -sin(lat) n + cos(lat) u, y = e, z = cos(lat) n + sin(lat) u.
    (GEOM.enu2xyz of the reference's astroutils dependency, interferometry.py:7979; un-vendored, parity unpinned: the convention is the one
    the reference's uvw rotation matrix at :7979-7985 is written for.)"""
    enu = NP.asarray(enu, dtype=NP.float64).reshape(-1, 3)
    lat = NP.radians(latitude) if units == 'degrees' else float(latitude)
    e, n, u = enu[:, 0], enu[:, 1], enu[:, 2]
    return NP.stack((-NP.sin(lat) * n + NP.cos(lat) * u, e, NP.cos(lat) * n + NP.sin(lat) * u), axis=1)


def xyz2enu(xyz, latitude, units='degrees'):
    """Inverse of enu2xyz (GEOM.xyz2enu, interferometry.py:6153): e = y, n = -sin(lat) x + cos(lat) z, u = cos(lat) x + sin(lat) z."""
    xyz = NP.asarray(xyz, dtype=NP.float64).reshape(-1, 3)
    lat = NP.radians(latitude) if units == 'degrees' else float(latitude)
    x, y, z = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    return NP.stack((y, -NP.sin(lat) * x + NP.cos(lat) * z, NP.cos(lat) * x + NP.sin(lat) * z), axis=1)
