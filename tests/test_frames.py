"""prisim_amd/frames.py against published worked examples (Meeus, "Astronomical Algorithms", 2nd ed.) -- the module cannot be pinned to
astropy here (not installed, and nothing of it may be restated from memory beyond the published polynomials), so the closed forms are
checked against numbers in print.  Tolerances are stated per test; the reference statements this stands in for are
prisim/interferometry.py:6113, 6174-6180 and scripts/run_prisim.py:1690-1692."""
import math

import numpy as NP
import pytest

from prisim_amd import frames as FR
from prisim_amd import geometry as GEOM

ARCSEC_DEG = 1.0 / 3600.0


def _unit(ra_deg, dec_deg):
    ra, dec = math.radians(ra_deg), math.radians(dec_deg)
    return NP.array([math.cos(dec) * math.cos(ra), math.cos(dec) * math.sin(ra), math.sin(dec)])


def _radec(v):
    return math.degrees(math.atan2(v[1], v[0])) % 360.0, math.degrees(math.asin(v[2] / NP.linalg.norm(v)))


def _sep_arcsec(a, b):
    return math.degrees(math.acos(min(1.0, float(NP.dot(a, b) / NP.linalg.norm(a) / NP.linalg.norm(b))))) * 3600.0


def _hms(h, m, s):
    return 15.0 * (h + m / 60.0 + s / 3600.0)


def _dms(d, m, s):
    return d + m / 60.0 + s / 3600.0


def test_precession_meeus_example_21b():
    """theta Persei, J2000.0 place with proper motion applied (2h44m12.975s, +49d13'39.90") -> mean place of 2028 Nov 13.19 TD
    (JDE 2462088.69): 2h46m11.331s, +49d20'54.54".  Meeus uses the IAU 1976 angles; the IAU 2006 angles used here differ from them by
    0.28 arcsec/century in the general precession in RA and 0.12 arcsec/century in theta, i.e. < 0.07 arcsec on the sky over these
    0.29 centuries.  Tolerance 0.1 arcsec."""
    u0 = _unit(_hms(2, 44, 12.975), _dms(49, 13, 39.90))
    p = FR.precession_matrix(2000.0, FR.jyear_of_jd(2462088.69))
    got = p.dot(u0)
    want = _unit(_hms(2, 46, 11.331), _dms(49, 20, 54.54))
    assert _sep_arcsec(got, want) < 0.1
    # the accumulated precession itself is 0.37 degrees here: the test is not vacuous
    assert _sep_arcsec(u0, want) > 1200.0
    # a rotation, and composition through J2000 is exact (run_prisim.py:1690 then interferometry.py:6174 == one step)
    assert NP.allclose(p.dot(p.T), NP.eye(3), atol=1e-15)
    p2 = FR.precession_matrix(2010.0, 2028.867).dot(FR.precession_matrix(2000.0, 2010.0))
    assert NP.allclose(p2, FR.precession_matrix(2000.0, 2028.867), atol=1e-15)


def test_precess_radec_roundtrip_and_epoch_strings():
    radec = NP.array([[10.0, -30.0], [200.0, 45.0], [359.9, 89.0]])
    out = FR.precess_radec(radec, 'J2000', 'J2026.5')
    back = FR.precess_radec(out, 2026.5, 'J2000.0')
    assert NP.allclose(back, radec, atol=1e-10)
    assert FR.jyear('J2000') == 2000.0 and FR.jyear('2000') == 2000.0 and FR.jyear(b'J2015.5') == 2015.5
    assert abs(FR.jyear('B1950') - 1949.999790) < 1e-5          # B1950.0 = JD 2433282.4235 = J1949.99979
    # 50.3 arcsec per year of general precession: J2000 -> J2026 moves an equatorial source by ~0.36 deg
    moved = _sep_arcsec(_unit(*radec[0]), _unit(*FR.precess_radec(radec[:1], 2000.0, 2026.0)[0])) / 3600.0
    assert 0.3 < moved < 0.4


def test_nutation_meeus_example_22a():
    """1987 April 10, 0h TD (JDE 2446895.5): dpsi = -3.788", deps = +9.443", eps0 = 23d26'27.407".  Full 63-term series in the book; the 31
    terms kept here: tolerance 0.03 arcsec."""
    dpsi, deps, eps0 = FR.nutation_angles(2446895.5)
    assert abs(dpsi / FR.ARCSEC - (-3.788)) < 0.03
    assert abs(deps / FR.ARCSEC - 9.443) < 0.03
    assert abs(math.degrees(eps0) - _dms(23, 26, 27.407)) < 0.001 * ARCSEC_DEG
    n = FR.nutation_matrix(2446895.5)
    assert NP.allclose(n.dot(n.T), NP.eye(3), atol=1e-15)


def test_sidereal_time_meeus_example_12a():
    """1987 April 10, 0h UT (JD 2446895.5): mean sidereal time at Greenwich 13h10m46.3668s, apparent 13h10m46.1351s.  Tolerance 1 ms of
    time (0.015 arcsec) for the mean, 3 ms for the apparent (truncated nutation)."""
    assert abs(FR.gmst_deg(2446895.5) - _hms(13, 10, 46.3668)) < 15.0 * 0.001 / 3600.0
    assert abs(FR.apparent_lst_deg(2446895.5, 0.0) - _hms(13, 10, 46.1351)) < 15.0 * 0.003 / 3600.0
    assert abs((FR.apparent_lst_deg(2446895.5, 21.4) - FR.apparent_lst_deg(2446895.5, 0.0)) % 360.0 - 21.4) < 1e-9


def test_apparent_place_meeus_example_23a():
    """theta Persei on 2028 Nov 13.19 TD: the mean place of date (2h46m11.331s, +49d20'54.54") plus nutation (+15.843", +6.218") and annual
    aberration (+30.045", +6.697") gives the apparent place 2h46m14.390s, +49d21'07.45".  Tolerance 0.1 arcsec for each piece and the sum."""
    jd = 2462088.69
    mean = _unit(_hms(2, 46, 11.331), _dms(49, 20, 54.54))
    ra0, dec0 = _radec(mean)
    # nutation alone
    ra1, dec1 = _radec(FR.nutation_matrix(jd).dot(mean))
    assert abs((ra1 - ra0) * 3600.0 - 15.843) < 0.1 and abs((dec1 - dec0) * 3600.0 - 6.218) < 0.1
    # aberration alone (book: Sun's longitude 231.328 deg, e = 0.01669649, perihelion 103.434 deg)
    lon, e, peri = FR.sun_longitude(jd)
    assert abs(math.degrees(lon) - 231.328) < 0.002 and abs(e - 0.01669649) < 1e-8 and abs(math.degrees(peri) - 103.434) < 0.001
    ra2, dec2 = _radec(mean + FR.aberration_beta(jd))
    assert abs((ra2 - ra0) * 3600.0 - 30.045) < 0.1 and abs((dec2 - dec0) * 3600.0 - 6.697) < 0.1
    # both, composed the way snapshot_frame composes them: N . normalise(u + beta)
    got = FR.nutation_matrix(jd).dot(mean + FR.aberration_beta(jd))
    want = _unit(_hms(2, 46, 14.390), _dms(49, 21, 7.45))
    assert _sep_arcsec(got, want) < 0.1


def test_snapshot_frame_date_model_is_the_hour_angle_rotation():
    """model 'date' (and hadec / altaz catalogues) is exactly HA = LST - RA -> hadec2altaz -> altaz2dircos, as one matrix.  Tolerance 3e-15: the OLD route rounds through
    degrees, asin and atan2 and back through sin / cos (a few ulp of 1 near the pole and the horizon); the matrix route has 3 roundings."""
    rng = NP.random.default_rng(3)
    radec = NP.stack((rng.uniform(0, 360, 500), NP.degrees(NP.arcsin(rng.uniform(-1, 1, 500)))), axis=1)
    lat, lst = -30.7224, 123.456
    r, beta = FR.snapshot_frame('radec', lst, lat, model='date')
    assert NP.all(beta == 0.0)
    got = GEOM.frame_dircos(GEOM.catalog_unitvec(radec, 'radec'), r, beta)
    want = GEOM.altaz2dircos(GEOM.hadec2altaz(NP.stack((lst - radec[:, 0], radec[:, 1]), axis=1), lat, units='degrees'), 'degrees')
    assert NP.max(NP.abs(got - want)) <= 3e-15
    rh, _ = FR.snapshot_frame('hadec', lst, lat)
    hadec = NP.stack((lst - radec[:, 0], radec[:, 1]), axis=1)
    assert NP.max(NP.abs(GEOM.frame_dircos(GEOM.catalog_unitvec(hadec, 'hadec'), rh, NP.zeros(3)) - want)) <= 3e-15
    ra_, _ = FR.snapshot_frame('altaz', lst, lat)
    altaz = GEOM.dircos2altaz(want)
    assert NP.max(NP.abs(GEOM.frame_dircos(GEOM.catalog_unitvec(altaz, 'altaz'), ra_, NP.zeros(3)) - want)) <= 3e-15


def test_snapshot_frame_models_differ_by_what_they_add():
    """J2000 catalogue observed in 2026: 'mean' moves the sky by the accumulated precession (0.37 deg), 'apparent' by at most another
    ~40 arcsec (nutation <= 19", aberration <= 20.5"); every R is a rotation."""
    jd = 2461300.5
    u = GEOM.catalog_unitvec(NP.array([[30.0, -30.0], [250.0, 10.0], [100.0, 70.0]]), 'radec')
    out = {}
    for model in FR.MODELS:
        r, beta = FR.snapshot_frame('radec', 75.0, -30.7224, jd=jd, epoch='J2000', model=model)
        assert NP.allclose(r.dot(r.T), NP.eye(3), atol=1e-15)
        out[model] = GEOM.frame_dircos(u, r, beta)
    for i in range(3):
        assert 500.0 < _sep_arcsec(out['date'][i], out['mean'][i]) < 1500.0        # 26.7 yr x (20" ... 50") per year, by position
        assert _sep_arcsec(out['mean'][i], out['apparent'][i]) < 45.0
    assert max(_sep_arcsec(out['mean'][i], out['apparent'][i]) for i in range(3)) > 5.0
    assert abs(NP.linalg.norm(FR.aberration_beta(jd)) / FR.ARCSEC - FR.KAPPA_ABERRATION_ARCSEC) < 0.4       # |v|/c within the orbit's eccentricity
    with pytest.raises(ValueError):
        FR.snapshot_frame('radec', 0.0, 0.0, jd=jd, epoch='J2000', model='nonsense')
    with pytest.raises(ValueError):
        FR.snapshot_frame('galactic', 0.0, 0.0)
