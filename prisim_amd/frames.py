"""Catalogue frame -> local East-North-Up frame of one snapshot, as ONE rotation and ONE aberration vector:

    s_enu = normalise( R . (u_cat + beta) )                                   (prisim_snapshot.cel2enu / aberr_beta, include/prisim_hip.h)

This is what the reference gets from astropy at every observe() (prisim/interferometry.py:6174-6180:
``SkyCoord(ra, dec, frame='fk5', equinox=skymodel.epoch).transform_to(FK5(equinox=timeobj)).transform_to(AltAz(obstime=timeobj, location))``;
scripts/run_prisim.py:1690-1692 precesses the sky model to the first timestamp before that).  astropy / ERFA are not available here
and nothing of them can be executed to pin this module (PARITY UNPINNED at astropy's last digits).  The published
closed forms are restated instead and checked against published worked examples (tests/test_frames.py: Meeus, "Astronomical
Algorithms" 2nd ed., examples 12.a, 21.b, 22.a, 23.a):

  modelled     precession          Capitaine et al. (2003) zeta_A, z_A, theta_A -- the IAU 2006 equatorial precession angles, the polynomial
                                   astropy's FK5 equinox change is built on (astropy.coordinates.earth_orientation.precession_matrix_Capitaine)
               nutation            IAU 1980 series truncated to its 31 largest terms (neglected terms sum to < 0.03 arcsec), mean obliquity IAU 1980
               annual aberration   first order, s + v/c with the Earth's velocity from a Kepler orbit (Meeus ch. 23/25: constant of
                                   aberration 20.49552 arcsec, eccentricity terms kept; error < 0.1 arcsec against the Ron-Vondrak series)
               Earth rotation      by the LOCAL APPARENT sidereal time the caller gives (the reference: timeobj.sidereal_time('apparent'), :6113)
               latitude tilt       equatorial (x to hour angle 0, y East, z pole) -> East, North, Up
  not modelled FK5 <-> ICRS frame bias (0.02 arcsec), light deflection by the Sun (< 0.01 arcsec beyond 30 deg elongation), diurnal
               aberration (<= 0.32 arcsec), polar motion (<= 0.5 arcsec), TT - UTC in the precession / nutation / aberration epoch
               (1e-4 arcsec), refraction (off in the reference: pressure = 0).
  A caller that HAS astropy fills R and beta exactly (INTEGRATION.md 2b shows the six lines) and sets InterferometerArray.frame_provider.

Conventions: every rotation below is a FRAME rotation (ERFA rx/ry/rz, astropy rotation_matrix): r3(a) takes the coordinates of a
fixed vector to a frame turned anticlockwise by a about z.  Angles in radians unless the name says _deg.
"""
import math

import numpy as NP

ARCSEC = math.pi / 648000.0
JD_J2000 = 2451545.0
KAPPA_ABERRATION_ARCSEC = 20.49552

MODELS = ('apparent', 'mean', 'date')


def r1(a):
    c, s = math.cos(a), math.sin(a)
    return NP.array([[1.0, 0.0, 0.0], [0.0, c, s], [0.0, -s, c]])


def r2(a):
    c, s = math.cos(a), math.sin(a)
    return NP.array([[c, 0.0, -s], [0.0, 1.0, 0.0], [s, 0.0, c]])


def r3(a):
    c, s = math.cos(a), math.sin(a)
    return NP.array([[c, s, 0.0], [-s, c, 0.0], [0.0, 0.0, 1.0]])


def jyear(epoch):
    """Julian year of 'J2000', 'J2000.0', '2000', 2000.0, or of an object with .jyear (astropy Time); jd -> jyear with jyear_of_jd."""
    if hasattr(epoch, 'jyear'):
        return float(epoch.jyear)
    if isinstance(epoch, (bytes, bytearray)):
        epoch = epoch.decode()
    if isinstance(epoch, str):
        s = epoch.strip()
        if s[:1] in ('J', 'j'):
            s = s[1:]
        elif s[:1] in ('B', 'b'):
            # Besselian epoch -> Julian year (Lieske 1979): JD = 2415020.31352 + (B - 1900) * 365.242198781
            return jyear_of_jd(2415020.31352 + (float(s[1:]) - 1900.0) * 365.242198781)
        return float(s)
    return float(epoch)


def jyear_of_jd(jd):
    return 2000.0 + (float(jd) - JD_J2000) / 365.25


def _poly(coefs, t):
    """coefs highest power first (numpy.polyval order)."""
    out = 0.0
    for c in coefs:
        out = out * t + c
    return out


def precession_from_j2000(epoch_jyear):
    """Rotation from the mean equator and equinox of J2000.0 to those of `epoch_jyear`: r3(-z) r2(theta) r3(-zeta) with the
    Capitaine et al. (2003) angles (arcsec; USNO Circular 179 eqs. 5.7-5.9 / IAU 2006)."""
    t = (float(epoch_jyear) - 2000.0) / 100.0
    zeta = _poly((-0.0000003173, -0.000005971, 0.01801828, 0.2988499, 2306.083227, 2.650545), t) * ARCSEC
    z = _poly((-0.0000002904, -0.000028596, 0.01826837, 1.0927348, 2306.077181, -2.650545), t) * ARCSEC
    theta = _poly((-0.0000001274, -0.000007089, -0.04182264, -0.4294934, 2004.191903, 0.0), t) * ARCSEC
    return r3(-z).dot(r2(theta)).dot(r3(-zeta))


def precession_matrix(from_jyear, to_jyear):
    """Mean place of equinox `from_jyear` -> mean place of equinox `to_jyear`, through J2000.0 (as astropy composes it)."""
    return precession_from_j2000(to_jyear).dot(precession_from_j2000(from_jyear).T)


# IAU 1980 nutation, the 31 largest terms.  Multiples of (D, M, M', F, Omega); longitude: (psi0 + psi1 T) sin(arg),
# obliquity: (eps0 + eps1 T) cos(arg); units 1e-4 arcsec.
_NUT_TERMS = (
    (0, 0, 0, 0, 1, -171996.0, -174.2, 92025.0, 8.9),
    (-2, 0, 0, 2, 2, -13187.0, -1.6, 5736.0, -3.1),
    (0, 0, 0, 2, 2, -2274.0, -0.2, 977.0, -0.5),
    (0, 0, 0, 0, 2, 2062.0, 0.2, -895.0, 0.5),
    (0, 1, 0, 0, 0, 1426.0, -3.4, 54.0, -0.1),
    (0, 0, 1, 0, 0, 712.0, 0.1, -7.0, 0.0),
    (-2, 1, 0, 2, 2, -517.0, 1.2, 224.0, -0.6),
    (0, 0, 0, 2, 1, -386.0, -0.4, 200.0, 0.0),
    (0, 0, 1, 2, 2, -301.0, 0.0, 129.0, -0.1),
    (-2, -1, 0, 2, 2, 217.0, -0.5, -95.0, 0.3),
    (-2, 0, 1, 0, 0, -158.0, 0.0, 0.0, 0.0),
    (-2, 0, 0, 2, 1, 129.0, 0.1, -70.0, 0.0),
    (0, 0, -1, 2, 2, 123.0, 0.0, -53.0, 0.0),
    (2, 0, 0, 0, 0, 63.0, 0.0, 0.0, 0.0),
    (0, 0, 1, 0, 1, 63.0, 0.1, -33.0, 0.0),
    (2, 0, -1, 2, 2, -59.0, 0.0, 26.0, 0.0),
    (0, 0, -1, 0, 1, -58.0, -0.1, 32.0, 0.0),
    (0, 0, 1, 2, 1, -51.0, 0.0, 27.0, 0.0),
    (-2, 0, 2, 0, 0, 48.0, 0.0, 0.0, 0.0),
    (0, 0, -2, 2, 1, 46.0, 0.0, -24.0, 0.0),
    (2, 0, 0, 2, 2, -38.0, 0.0, 16.0, 0.0),
    (0, 0, 2, 2, 2, -31.0, 0.0, 13.0, 0.0),
    (0, 0, 2, 0, 0, 29.0, 0.0, 0.0, 0.0),
    (-2, 0, 1, 2, 2, 29.0, 0.0, -12.0, 0.0),
    (0, 0, 0, 2, 0, 26.0, 0.0, 0.0, 0.0),
    (-2, 0, 0, 2, 0, -22.0, 0.0, 0.0, 0.0),
    (0, 0, -1, 2, 1, 21.0, 0.0, -10.0, 0.0),
    (0, 2, 0, 0, 0, 17.0, -0.1, 0.0, 0.0),
    (2, 0, -1, 0, 1, 16.0, 0.0, -8.0, 0.0),
    (-2, 2, 0, 2, 2, -16.0, 0.1, 7.0, 0.0),
    (0, 1, 0, 0, 1, -15.0, 0.0, 9.0, 0.0),
)


_NUT = NP.array(_NUT_TERMS, dtype=NP.float64)
_NUT_MULT, _NUT_P0, _NUT_P1, _NUT_E0, _NUT_E1 = _NUT[:, :5].copy(), _NUT[:, 5].copy(), _NUT[:, 6].copy(), _NUT[:, 7].copy(), _NUT[:, 8].copy()


def nutation_angles(jd):
    """(dpsi, deps, eps0) in radians at Julian date jd: nutation in longitude and obliquity (truncated IAU 1980) and the mean obliquity
    of the ecliptic (IAU 1980: 23 26 21.448 - 46.8150 T - 0.00059 T^2 + 0.001813 T^3)."""
    t = (float(jd) - JD_J2000) / 36525.0
    fund = NP.radians(NP.array([297.85036 + 445267.111480 * t - 0.0019142 * t * t + t ** 3 / 189474.0,            # D
                                357.52772 + 35999.050340 * t - 0.0001603 * t * t - t ** 3 / 300000.0,             # M
                                134.96298 + 477198.867398 * t + 0.0086972 * t * t + t ** 3 / 56250.0,             # M'
                                93.27191 + 483202.017538 * t - 0.0036825 * t * t + t ** 3 / 327270.0,             # F
                                125.04452 - 1934.136261 * t + 0.0020708 * t * t + t ** 3 / 450000.0]))            # Omega
    arg = _NUT_MULT.dot(fund)
    dpsi = float(NP.dot(_NUT_P0 + _NUT_P1 * t, NP.sin(arg)))
    deps = float(NP.dot(_NUT_E0 + _NUT_E1 * t, NP.cos(arg)))
    eps0 = (84381.448 - 46.8150 * t - 0.00059 * t * t + 0.001813 * t ** 3) * ARCSEC
    return dpsi * 1e-4 * ARCSEC, deps * 1e-4 * ARCSEC, eps0


def nutation_matrix(jd, angles=None):
    """Mean equator and equinox of date -> true equator and equinox of date: r1(-(eps0 + deps)) r3(-dpsi) r1(eps0)."""
    dpsi, deps, eps0 = nutation_angles(jd) if angles is None else angles
    return r1(-(eps0 + deps)).dot(r3(-dpsi)).dot(r1(eps0))


def sun_longitude(jd):
    """(true geometric longitude of the Sun, eccentricity of the Earth's orbit, longitude of its perihelion), radians / 1 / radians, referred to
    the mean equinox of date (Meeus ch. 25 low-accuracy theory, 0.01 deg; ch. 23 for the perihelion)."""
    t = (float(jd) - JD_J2000) / 36525.0
    l0 = 280.46646 + 36000.76983 * t + 0.0003032 * t * t
    m = math.radians(357.52911 + 35999.05029 * t - 0.0001537 * t * t)
    e = 0.016708634 - 0.000042037 * t - 0.0000001267 * t * t
    c = ((1.914602 - 0.004817 * t - 0.000014 * t * t) * math.sin(m) + (0.019993 - 0.000101 * t) * math.sin(2.0 * m)
         + 0.000289 * math.sin(3.0 * m))
    peri = 102.93735 + 1.71946 * t + 0.00046 * t * t
    return math.radians((l0 + c) % 360.0), e, math.radians(peri)


def aberration_beta(jd, eps0=None):
    """Earth's barycentric velocity / c in the MEAN equatorial frame of date, from the velocity of a Kepler orbit:
    kappa (sin L - e sin w, -cos L + e cos w, 0) in the ecliptic frame (L: longitude of the Sun, w: longitude of the perihelion),
    turned about x by the mean obliquity.  First-order annual aberration is then s' = normalise(s + beta)."""
    lon, e, peri = sun_longitude(jd)
    k = KAPPA_ABERRATION_ARCSEC * ARCSEC
    bx = k * (math.sin(lon) - e * math.sin(peri))
    by = k * (-math.cos(lon) + e * math.cos(peri))
    if eps0 is None:
        eps0 = nutation_angles(jd)[2]
    return NP.array([bx, by * math.cos(eps0), by * math.sin(eps0)])


def gmst_deg(jd_ut1):
    """Greenwich mean sidereal time (degrees) at Julian date jd (UT1), IAU 1982 (Meeus eq. 12.4)."""
    t = (float(jd_ut1) - JD_J2000) / 36525.0
    return (280.46061837 + 360.98564736629 * (float(jd_ut1) - JD_J2000) + 0.000387933 * t * t - t ** 3 / 38710000.0) % 360.0


def apparent_lst_deg(jd_ut1, longitude_deg):
    """Local apparent sidereal time (degrees): GMST + dpsi cos(eps) + East longitude -- what timeobj.sidereal_time('apparent') is in the
    reference (interferometry.py:6113), for callers without astropy."""
    dpsi, deps, eps0 = nutation_angles(jd_ut1)
    return (gmst_deg(jd_ut1) + math.degrees(dpsi * math.cos(eps0 + deps)) + float(longitude_deg)) % 360.0


def equatorial_to_enu(lst_deg, latitude_deg):
    """Equatorial frame of date (x to the equinox, z to the pole) -> East, North, Up at local sidereal time lst and the given latitude:
    hour-angle frame first (x to the meridian, y to the EAST: -sin of the hour angle), then the tilt by the co-latitude --
    tilt . r3(lst) with tilt = [[0, 1, 0], [-sin lat, 0, cos lat], [cos lat, 0, sin lat]], written out (the entries libprisim_hip.so's
    fall-back frame computes, catalog.cpp fallback_frame)."""
    lat, a = math.radians(latitude_deg), math.radians(lst_deg)
    sl, cl, s, c = math.sin(lat), math.cos(lat), math.sin(a), math.cos(a)
    return NP.array([[-s, c, 0.0], [-sl * c, -sl * s, cl], [cl * c, cl * s, sl]])


def hadec_to_enu(latitude_deg):
    """(HA, Dec) unit vectors (cos d cos H, cos d sin H, sin d) -> East, North, Up (geometry.hadec2altaz + altaz2dircos as one matrix)."""
    sl, cl = math.sin(math.radians(latitude_deg)), math.cos(math.radians(latitude_deg))
    return NP.array([[0.0, -1.0, 0.0], [-sl, 0.0, cl], [cl, 0.0, sl]])


def snapshot_frame(coords, lst_deg, latitude_deg, jd=None, epoch=None, model='apparent'):
    """(R (3, 3), beta (3,)) of one snapshot for a catalogue in `coords`:

    'radec'   model 'date'      R = tilt(lat) r3(LST): the catalogue is taken to be in the true equator and equinox of date (HA = LST - RA,
                                what this package did before ABI 0.5); beta = 0.  Also whenever epoch is None / 'date' (a sky model that
                                SAYS its coordinates are of date -- the synthetic skies of prisim_amd/driver.py) or jd is None
              model 'mean'      ... times the precession matrix epoch -> jd
              model 'apparent'  ... times nutation, and beta = annual aberration turned back into the catalogue frame
    'hadec'   R = hadec_to_enu(lat), beta = 0     (interferometry.py:6176-6177: GEOM.hadec2altaz, no astropy)
    'altaz'   R = identity, beta = 0              (the catalogue is local already)
    """
    if coords == 'altaz':
        return NP.eye(3), NP.zeros(3)
    if coords == 'hadec':
        return hadec_to_enu(latitude_deg), NP.zeros(3)
    if coords != 'radec':
        raise ValueError('coords must be "radec", "hadec" or "altaz"')
    if model not in MODELS:
        raise ValueError('frame model must be one of {0}'.format(MODELS))
    local = equatorial_to_enu(lst_deg, latitude_deg)
    if model == 'date' or jd is None or epoch is None or (isinstance(epoch, str) and epoch.strip().lower() == 'date'):
        return local, NP.zeros(3)             # a catalogue given in the coordinates of date (SkyModel.epoch None / 'date'): nothing to add
    prec = precession_from_j2000(jyear_of_jd(jd)).dot(_from_epoch_to_j2000(epoch))
    if model == 'mean':
        return local.dot(prec), NP.zeros(3)
    ang = nutation_angles(jd)
    return local.dot(nutation_matrix(jd, ang)).dot(prec), prec.T.dot(aberration_beta(jd, ang[2]))


_EPOCH_CACHE = {}


def _from_epoch_to_j2000(epoch):
    """Transpose of precession_from_j2000(epoch): the catalogue's equinox does not change over a run, one matrix per epoch is kept."""
    key = epoch if isinstance(epoch, (str, float, int)) else jyear(epoch)
    m = _EPOCH_CACHE.get(key)
    if m is None:
        if len(_EPOCH_CACHE) > 64:
            _EPOCH_CACHE.clear()
        m = _EPOCH_CACHE[key] = precession_from_j2000(jyear(epoch)).T.copy()
    return m


def precess_radec(radec_deg, from_epoch, to_epoch):
    """(RA, Dec) degrees of equinox from_epoch -> equinox to_epoch (scripts/run_prisim.py:1690-1691: the sky model is precessed once to
    the first timestamp of the run).  Epochs as jyear() takes them."""
    radec = NP.asarray(radec_deg, dtype=NP.float64).reshape(-1, 2)
    ra, dec = NP.radians(radec[:, 0]), NP.radians(radec[:, 1])
    u = NP.stack((NP.cos(dec) * NP.cos(ra), NP.cos(dec) * NP.sin(ra), NP.sin(dec)), axis=0)
    v = precession_matrix(jyear(from_epoch), jyear(to_epoch)).dot(u)
    out = NP.stack((NP.degrees(NP.arctan2(v[1], v[0])) % 360.0, NP.degrees(NP.arcsin(NP.clip(v[2], -1.0, 1.0)))), axis=1)
    return out
