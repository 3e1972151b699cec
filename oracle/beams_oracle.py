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
