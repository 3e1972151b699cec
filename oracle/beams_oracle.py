"""numpy restatement of the analytic primary beams on the hot path (TEST INFRASTRUCTURE).

Follows prisim/primary_beams.py:
  * gaussian_beam        :629-730   (sigma_aprtr :717, sigma_dircos :721, pattern :724, blank :725, power :727-728)
  * airy_disk_pattern    :517-625   (k :609, small-angle clamp :611-612, pattern :614, blank :616, peak-normalise :618-623)
  * primary_beam_generator dispatch for telescope shape 'gaussian' / 'dish' / 'delta' (:354-416) and
    id 'hera' / 'hirax' (:239-247): power = |field|^2.
Pinned by tests/golden/golden_beams.npz (the reference functions executed on seeded inputs) for
zenith pointing.  Off-zenith pointing uses the angular distance to the pointing centre (GEOM.sphdist
in the reference, astroutils: PARITY UNPINNED).
"""
import numpy as NP
import scipy.special as SPS

C_LIGHT = 299792458.0


def _angles(skypos_altaz, pointing_altaz):
    """Angle x (radians) from the pointing centre, and the blanking mask (:584 / :607 / :691 / :714)."""
    skypos = NP.asarray(skypos_altaz, dtype=NP.float64).reshape(-1, 2)
    if pointing_altaz is None:
        x = NP.radians(90.0 - skypos[:, 0])                               # :579 / :686
        zero_ind = x >= NP.pi / 2                                         # :584 / :691
    else:
        pc = NP.asarray(pointing_altaz, dtype=NP.float64).ravel()
        a1, z1 = NP.radians(skypos[:, 0]), NP.radians(skypos[:, 1])
        a0, z0 = NP.radians(pc[0]), NP.radians(pc[1])
        cosx = NP.sin(a1) * NP.sin(a0) + NP.cos(a1) * NP.cos(a0) * NP.cos(z1 - z0)
        x = NP.arccos(NP.clip(cosx, -1.0, 1.0))                           # GEOM.sphdist, :605 / :712
        zero_ind = NP.logical_or(x >= NP.pi / 2, skypos[:, 0] <= 0.0)     # :607 / :714
    return x, zero_ind


def gaussian_beam(diameter, skypos_altaz, frequency_hz, pointing_altaz=None, power=True):
    frequency = NP.asarray(frequency_hz, dtype=NP.float64).ravel()
    x, zero_ind = _angles(skypos_altaz, pointing_altaz)
    x = x.reshape(-1, 1)
    sigma_aprtr = diameter / (2.0 * NP.sqrt(2.0 * NP.log(2.0))) / (C_LIGHT / frequency)   # :717
    sigma_dircos = (1.0 / (2 * NP.pi * sigma_aprtr)).reshape(1, -1)                         # :721-722
    pattern = NP.exp(-0.5 * (NP.sin(x) / sigma_dircos) ** 2)                                # :723-724
    pattern[zero_ind, :] = 0.0                                                              # :725
    if power:
        pattern = NP.abs(pattern) ** 2                                                      # :727-728
    return pattern


def airy_disk_pattern(diameter, skypos_altaz, frequency_hz, pointing_altaz=None, peak=1.0, small_angle_tol=1e-10, power=True):
    frequency = NP.asarray(frequency_hz, dtype=NP.float64).ravel()
    x, zero_ind = _angles(skypos_altaz, pointing_altaz)
    k = (2 * NP.pi * frequency / C_LIGHT).reshape(1, -1)                                    # :609-610
    x = NP.where(x < small_angle_tol, small_angle_tol, x).reshape(-1, 1)                    # :611-613
    arg = k * 0.5 * diameter * NP.sin(x)
    pattern = 2 * SPS.j1(arg) / arg                                                         # :614
    pattern[zero_ind, :] = 0.0                                                              # :616
    arg0 = k * 0.5 * diameter * NP.sin(small_angle_tol)
    maxval = 2 * SPS.j1(arg0) / arg0                                                        # :618
    if power:
        pattern = NP.abs(pattern) ** 2                                                      # :620
        maxval = maxval ** 2                                                                # :621
    return pattern * (peak / maxval)                                                        # :623


def primary_beam_generator(skypos_altaz, frequency_hz, telescope, pointing_altaz=None):
    """Power pattern (nsrc, nchan) for the dispatch branches the BASELINE configs use (:224-247, :354-416)."""
    if not isinstance(telescope, dict):
        raise TypeError('telescope must be specified as a dictionary')
    if 'id' in telescope and telescope['id'] in ('hera', 'hirax'):
        dia = 14.0 if telescope['id'] == 'hera' else 6.0                                    # :240-243
        return airy_disk_pattern(dia, skypos_altaz, frequency_hz, pointing_altaz=pointing_altaz, power=True)   # :244-247
    shape = telescope.get('shape', 'delta')
    nsrc = NP.asarray(skypos_altaz).reshape(-1, 2).shape[0]
    nchan = NP.asarray(frequency_hz).size
    if shape == 'delta':
        return NP.ones((nsrc, nchan))                                                       # :357-359, :416
    if shape == 'dish':
        ep = airy_disk_pattern(telescope['size'], skypos_altaz, frequency_hz, pointing_altaz=pointing_altaz, power=False)  # :370
    elif shape == 'gaussian':
        ep = gaussian_beam(telescope['size'], skypos_altaz, frequency_hz, pointing_altaz=pointing_altaz, power=False)      # :375
    else:
        raise ValueError('Value in key "shape" of telescope dictionary invalid.')
    return NP.abs(ep) ** 2                                                                  # :416


# ---- dipole, isotropic-radiator array factor, ground plane (direction-cosine inputs) --------------------

def dipole_field_pattern(length, dircos, wavelength, dipole_dircos=(1.0, 0.0, 0.0), short_dipole_approx=False,
                         half_wave_dipole_approx=False):
    """Field pattern of a dipole (primary_beams.py:1203-1235), skypos and orientation as direction cosines."""
    dircos = NP.asarray(dircos, dtype=NP.float64).reshape(-1, 3)
    wavelength = NP.asarray(wavelength, dtype=NP.float64).ravel()
    k = 2 * NP.pi / wavelength.reshape(1, -1)                                               # :1203
    h = 0.5 * length                                                                         # :1204
    dot_product = NP.dot(NP.asarray(dipole_dircos, dtype=NP.float64).reshape(1, 3), dircos.T).reshape(-1, 1)   # :1205
    angles = NP.arccos(NP.clip(dot_product, -1.0, 1.0))                                      # :1206
    zero = (NP.abs(NP.abs(dot_product) - 1.0) < 1.0e-10).ravel()                             # :1209
    max_pattern = 1.0
    with NP.errstate(divide='ignore', invalid='ignore'):
        if short_dipole_approx:
            field = NP.repeat(NP.sin(angles).reshape(-1, 1), wavelength.size, axis=1)       # :1215-1216
        else:
            if half_wave_dipole_approx:
                field = NP.cos(0.5 * NP.pi * NP.cos(angles)) / NP.sin(angles)               # :1219
                field = NP.repeat(field.reshape(-1, 1), wavelength.size, axis=1)
            else:
                max_pattern = 1.0 - NP.cos(k * h)                                           # :1222
                field = (NP.cos(k * h * NP.cos(angles)) - NP.cos(k * h)) / NP.sin(angles)   # :1223
            if NP.any(zero):
                field[zero, :] = k * h * NP.sin(k * h * NP.cos(angles[zero])) * NP.tan(angles[zero])   # :1226
    return field / max_pattern                                                              # :1232


def isotropic_radiators_array_field_pattern(nax1, nax2, sep1, sep2, dircos, wavelength, east2ax1=0.0,
                                            pointing_dircos=(0.0, 0.0, 1.0)):
    """Array factor of an nax1 x nax2 grid of isotropic radiators (primary_beams.py:1436-1475).
    The reference uses nax1 in both terms (:1469-1471, harmless for 4 x 4); nax2 is used here."""
    dircos = NP.asarray(dircos, dtype=NP.float64).reshape(-1, 3)
    wavelength = NP.asarray(wavelength, dtype=NP.float64).ravel()
    angle = NP.radians(east2ax1)
    rot = NP.asarray([[NP.cos(angle), NP.sin(angle), 0.0], [-NP.sin(angle), NP.cos(angle), 0.0], [0.0, 0.0, 1.0]])   # :1443-1445
    rel = NP.dot(dircos, rot.T) - NP.dot(NP.asarray(pointing_dircos, dtype=NP.float64), rot.T).reshape(1, -1)      # :1446-1449
    phi = 2 * NP.pi * sep1 * rel[:, [0]] / wavelength.reshape(1, -1)                        # :1458
    psi = 2 * NP.pi * sep2 * rel[:, [1]] / wavelength.reshape(1, -1)                        # :1459
    with NP.errstate(divide='ignore', invalid='ignore'):
        t1 = NP.sin(0.5 * nax1 * phi) / NP.sin(0.5 * phi) / nax1                            # :1465
        t1 = NP.where(NP.abs(phi) < 1e-10, NP.cos(0.5 * nax1 * phi) / NP.cos(0.5 * phi), t1)   # :1466-1467
        t2 = NP.sin(0.5 * nax2 * psi) / NP.sin(0.5 * psi) / nax2                            # :1469
        t2 = NP.where(NP.abs(psi) < 1e-10, NP.cos(0.5 * nax2 * psi) / NP.cos(0.5 * psi), t2)   # :1470-1471
    return t1 * t2                                                                          # :1473


def ground_plane_field_pattern(height, dircos, wavelength, modifier=None):
    """Field pattern of a ground plane at `height` (primary_beams.py:950-966).  sin(alt) = n.
    PARITY UNPINNED (the reference reaches alt through GEOM.dircos2altaz, astroutils)."""
    dircos = NP.asarray(dircos, dtype=NP.float64).reshape(-1, 3)
    k = 2 * NP.pi / NP.asarray(wavelength, dtype=NP.float64).reshape(1, -1)                 # :950
    gp = 2 * NP.sin(k * height * dircos[:, [2]])                                            # :953
    if isinstance(modifier, dict):                                                          # :955-963
        with NP.errstate(divide='ignore'):
            val = 1.0 / NP.sqrt(NP.abs(dircos[:, 2]))
        if 'scale' in modifier:
            val = val * modifier['scale']
        if 'max' in modifier:
            val = NP.clip(val, 0.0, modifier['max'])
        gp = gp * val[:, NP.newaxis]
    return gp / (2 * NP.sin(k * height))                                                    # :965-966


def polynomial_beam(coef, zenith_angle_deg, frequency_hz):
    """VLA / GMRT polynomial power beam (primary_beams.py:503-509 / :796-801): x = (angle [deg] * 60 * f [GHz])^2,
    1 + c0 x/1e3 + c1 x^2/1e7 + c2 x^3/1e10 (+ c3 x^4/1e13)."""
    th = NP.asarray(zenith_angle_deg, dtype=NP.float64).reshape(-1, 1)
    f = NP.asarray(frequency_hz, dtype=NP.float64).reshape(1, -1) / 1e9
    x = (th * 60.0 * f) ** 2
    c = list(coef) + [0.0] * (4 - len(coef))
    return 1.0 + c[0] * x / 1e3 + c[1] * (x ** 2) / 1e7 + c[2] * (x ** 3) / 1e10 + c[3] * (x ** 4) / 1e13


def beamformer_settings(antpos, pointing_info):
    """Element delays [nelem, nrand] (seconds) and gains [nelem, nrand] of the phased-array beamformer from a reference
    ``pointing_info`` dictionary (primary_beams.py:1595-1668): explicit 'delays', or delay compensation towards
    'pointing_center' (dircos; :1632), optional 'gains', and Gaussian jitter 'delayerr' (seconds) / 'gainerr' (dB) drawn from
    numpy's GLOBAL generator in the reference's order (delays first, then gains; :1655, :1665) -- seed it to reproduce a draw."""
    antpos = NP.asarray(antpos, dtype=NP.float64)
    nel = antpos.shape[0]
    if pointing_info is None:
        return NP.zeros((nel, 1)), NP.ones((nel, 1))
    nrand = pointing_info.get('nrand', 1) or 1
    if pointing_info.get('delays', None) is not None:
        delays = NP.asarray(pointing_info['delays'], dtype=NP.float64).ravel()
    elif 'pointing_center' in pointing_info and 'delays' not in pointing_info:
        pc = NP.asarray(pointing_info['pointing_center'], dtype=NP.float64).reshape(1, -1)
        delays = (NP.dot(antpos, pc.T) / C_LIGHT).ravel()                                # :1632
    else:
        delays = NP.zeros(nel)
    gains = pointing_info.get('gains', None)
    gains = NP.ones(nel) if gains is None else NP.asarray(gains, dtype=NP.float64).ravel()
    if pointing_info.get('delayerr', None) is not None:
        delays = delays.reshape(nel, 1) + pointing_info['delayerr'] * NP.random.standard_normal((nel, nrand))      # :1655
    if pointing_info.get('gainerr', None) is not None:
        gains = gains.reshape(nel, 1) * 10 ** (pointing_info['gainerr'] / 10.0 * NP.random.standard_normal((nel, nrand)))   # :1664-1665
    return NP.broadcast_to(delays.reshape(nel, -1), (nel, nrand)).copy(), NP.broadcast_to(gains.reshape(nel, -1), (nel, nrand)).copy()


def array_field_pattern(antpos, dircos, wavelength, delays=None, gains=None, power=False, single=True):
    """Field of an array of isotropic radiators with beamformer delays and gains (primary_beams.py:1670-1754):
    F[s, f, r] = (1/N) sum_i g[i, r] exp(2 pi i c / lambda_f (-antpos_i . s / c + delay[i, r])).
    single=True reproduces the reference's float32 / complex64 arithmetic statement by statement (:1670-1671, :1726, :1733-1746),
    which is what tests/golden/golden_beamformer.npz pins; single=False is the same sum in float64 (what the device evaluates)."""
    antpos = NP.asarray(antpos, dtype=NP.float64)
    nel = antpos.shape[0]
    delays = NP.zeros((nel, 1)) if delays is None else NP.asarray(delays, dtype=NP.float64).reshape(nel, -1)
    gains = NP.ones((nel, 1)) if gains is None else NP.asarray(gains, dtype=NP.float64).reshape(nel, -1)
    nrand = max(delays.shape[1], gains.shape[1])
    delays = NP.broadcast_to(delays, (nel, nrand))
    gains = NP.broadcast_to(gains, (nel, nrand))
    dircos = NP.asarray(dircos, dtype=NP.float64).reshape(-1, 3)
    wl = NP.asarray(wavelength, dtype=NP.float64).ravel()
    if not single:
        geo = -NP.dot(antpos, dircos.T) / C_LIGHT                                          # [nel, nsrc]
        ph = (geo[:, :, None, None] + delays[:, None, None, :]) * (C_LIGHT / wl)[None, None, :, None]
        field = NP.sum(gains[:, None, None, :] * NP.exp(2j * NP.pi * ph), axis=0) / nel
    else:
        g32 = gains.astype(NP.float32)                                                     # :1670
        d32 = delays.astype(NP.float32)                                                    # :1671
        sky32 = dircos.astype(NP.float32)                                                  # :1713
        wl32 = wl.astype(NP.float32)                                                       # :1726
        geo = -NP.dot(antpos, sky32.T) / C_LIGHT                                           # :1728
        geo = geo[:, :, NP.newaxis, NP.newaxis].astype(NP.float32)                         # :1729
        gc = g32.reshape(nel, 1, 1, nrand).astype(NP.complex64)                            # :1731
        dd = d32.reshape(nel, 1, 1, nrand)                                                 # :1732
        wlr = wl32.reshape(1, 1, -1, 1)                                                    # :1733
        ret = (geo + dd).astype(NP.complex64)                                              # :1735-1736
        ret = NP.exp(1j * 2 * NP.pi * C_LIGHT / wlr * ret).astype(NP.complex64)            # :1740
        ret *= gc / nel                                                                    # :1741
        field = NP.sum(ret.astype(NP.complex64), axis=0)                                   # :1742
    return NP.abs(field) ** 2 if power else field


def composite_power_beam(dircos, frequency_hz, element='delta', size=0.0, element_dircos=(0.0, 0.0, 1.0), dipole_mode='general',
                         array=None, ground=None, beamformer=None):
    """power = |element_field x array_factor|^2 x ground_field^2 (primary_beams.py:317, 349, 416, 439).
    element: 'delta' | 'gaussian' | 'dish' | 'dipole'.  array: dict(nax1, nax2, sep1, sep2, east2ax1, pointing_dircos).
    ground: dict(height, modifier).  beamformer: dict(positions [n,3], delays [n,nrand], gains [n,nrand], single) -- then the
    power is the mean over realisations of |element field x beamformed field|^2 (:317, :416)."""
    dircos = NP.asarray(dircos, dtype=NP.float64).reshape(-1, 3)
    f = NP.asarray(frequency_hz, dtype=NP.float64).ravel()
    wl = C_LIGHT / f
    if element == 'delta':
        ep = NP.ones((dircos.shape[0], f.size))
    elif element == 'dipole':
        ep = dipole_field_pattern(size, dircos, wl, dipole_dircos=element_dircos, short_dipole_approx=(dipole_mode == 'short'),
                                  half_wave_dipole_approx=(dipole_mode == 'halfwave'))
    else:
        alt = NP.degrees(NP.arcsin(NP.clip(dircos[:, 2], -1, 1)))
        az = NP.degrees(NP.arctan2(dircos[:, 0], dircos[:, 1])) % 360.0
        e = NP.asarray(element_dircos, dtype=NP.float64)
        pc = [NP.degrees(NP.arcsin(NP.clip(e[2], -1, 1))), NP.degrees(NP.arctan2(e[0], e[1])) % 360.0]
        fn = gaussian_beam if element == 'gaussian' else airy_disk_pattern
        ep = fn(size, NP.stack((alt, az), axis=1), f, pointing_altaz=pc, power=False)
    af = 1.0
    if array is not None:
        af = isotropic_radiators_array_field_pattern(array['nax1'], array['nax2'], array['sep1'], array['sep2'], dircos, wl,
                                                     east2ax1=array.get('east2ax1', 0.0),
                                                     pointing_dircos=array.get('pointing_dircos', (0.0, 0.0, 1.0)))
    if beamformer is not None:
        irap = array_field_pattern(beamformer['positions'], dircos, wl, delays=beamformer.get('delays'), gains=beamformer.get('gains'),
                                   power=False, single=beamformer.get('single', False))
        pb = NP.mean(NP.abs(NP.asarray(ep)[:, :, None] * irap) ** 2, axis=2) if NP.ndim(ep) == 2 else NP.mean(NP.abs(ep * irap) ** 2, axis=2)
    else:
        pb = NP.abs(ep * af) ** 2
    if ground is not None:
        pb = pb * ground_plane_field_pattern(ground['height'], dircos, wl, modifier=ground.get('modifier', None)) ** 2
    return pb
