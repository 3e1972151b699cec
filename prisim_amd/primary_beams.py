"""Host mirror of the reference's primary-beam dispatcher for the leaves on the accelerated path.

``primary_beam_generator`` keeps the reference's name and argument meaning
(prisim/primary_beams.py:9-441); the patterns themselves are evaluated ON THE GPU by the fused
beam kernel (prisim_amd/csrc/aux_kernels.hip, prisim_hip_set_sky_analytic):
    telescope id 'hera' / 'hirax'        -> Airy power pattern, D = 14 m / 6 m     (:239-247)
    telescope shape 'dish'               -> Airy power pattern of diameter 'size'  (:369-373, :416)
    telescope shape 'gaussian'           -> Gaussian power pattern                 (:374-377, :416)
    telescope shape 'delta' / no shape   -> 1                                      (:355-359, :416)
    telescope id 'mwa'                   -> dipole (0.74 m) x 4x4 array factor     (:248-317, analytic path)
    telescope id 'mwa_dipole' / 'paper'  -> dipole 0.74 m / 2.0 m                   (:320-349)
    telescope shape 'dipole'             -> dipole of length 'size'                 (:360-368)
    'groundplane' (+ 'ground_modify')    -> ground-plane factor                     (:418-439, :950-966)
The phased-array beamformer (pointing_info: delays / gains / pointing centre / delay and gain jitter, array_field_pattern
:1482-1754) is evaluated on the device too; its settings (and the random draws of the jitter) are formed here on the host.
The VLA / GMRT polynomial beams (:445-513, :734-808) are a fifth device beam kind.  rect/square apertures raise
NotImplementedError here (the reference code for them is broken) -- there is no CPU stand-in.
"""
import numpy as NP

from . import _abi
from . import geometry as GEOM


def _pointing_dircos(pointing_center, pointing_coords):
    if pointing_center is None:
        return NP.array([0.0, 0.0, 1.0])
    pc = NP.asarray(pointing_center, dtype=NP.float64).ravel()
    if pointing_coords in (None, 'altaz'):
        if pc.size != 2:
            raise IndexError('Pointing center in Alt-Az coordinates must contain exactly two elements.')
        return GEOM.altaz2dircos(pc, 'degrees').ravel()
    if pointing_coords == 'dircos':
        if pc.size != 3:
            raise IndexError('Pointing center in direction cosine coordinates must contain exactly three elements.')
        return pc
    raise ValueError('pointing coordinates must be "altaz" or "dircos"')


def _ground_ext(telescope):
    """'groundplane' / 'ground_modify' keys (primary_beams.py:418-439): applied unless the shape is 'dish'."""
    if telescope.get('groundplane', None) is None:
        return None
    if telescope.get('shape', None) == 'dish':
        return None
    return {'height': float(telescope['groundplane']), 'modifier': telescope.get('ground_modify', None)}


def _dipole_axis(telescope):
    """Dipole axis from 'orientation'/'ocoords' with the reference's defaults (:250-265, :326-341): East."""
    if ('orientation' in telescope) and ('ocoords' in telescope):
        return _pointing_dircos(NP.asarray(telescope['orientation']), telescope['ocoords'])
    if ('orientation' not in telescope) and ('ocoords' in telescope):
        if telescope['ocoords'] not in ('altaz', 'dircos'):
            raise ValueError('key "ocoords" in telescope dictionary contains invalid value')
        return NP.array([1.0, 0.0, 0.0])                       # altaz [0, 90] == dircos [1, 0, 0]
    if ('orientation' in telescope) and ('ocoords' not in telescope):
        raise KeyError('key "ocoords" in telescope dictionary not specified.')
    return NP.array([1.0, 0.0, 0.0])


def mwa_tile_element_locs():
    """The 4 x 4 dipole grid of an MWA tile at 1.1 m spacing, ENU metres, row-major from the north-west corner (:289-292)."""
    x, y = NP.meshgrid(1.1 * NP.linspace(-1.5, 1.5, 4), 1.1 * NP.linspace(1.5, -1.5, 4))
    return NP.stack((x.ravel(), y.ravel(), NP.zeros(x.size)), axis=1)


def beamformer_settings(element_locs, pointing_info):
    """Per-element compensation delays (seconds) and gains, shape (nelem, nrand), from a reference pointing_info dictionary
    (array_field_pattern, primary_beams.py:1595-1668).  Keys: 'delays' (nelem,) | 'pointing_center' + 'pointing_coords'
    ('altaz' degrees or 'dircos'; delay = element . pointing / c, :1632), 'gains' (nelem,), 'delayerr' (s), 'gainerr' (dB),
    'nrand'.  Jitter is drawn from numpy's global generator, delays before gains, like the reference (:1655, :1665)."""
    pos = NP.asarray(element_locs, dtype=NP.float64)
    if pos.ndim != 2 or pos.shape[1] != 3:
        raise ValueError('element_locs must be an Nx3 array (ENU metres)')
    nel = pos.shape[0]
    if pointing_info is None:
        return NP.zeros((nel, 1)), NP.ones((nel, 1))
    if not isinstance(pointing_info, dict):
        raise TypeError('pointing_info must be a dictionary')
    nrand = pointing_info.get('nrand', 1)
    if nrand is None:
        nrand = 1
    elif not isinstance(nrand, (int, NP.integer)):
        raise TypeError('nrand must be an integer')
    elif nrand < 1:
        raise ValueError('nrand must be positive')
    if 'delays' in pointing_info:
        delays = pointing_info['delays']
        if delays is None:
            delays = NP.zeros(nel)
        elif not isinstance(delays, NP.ndarray):
            raise TypeError('delays must be a numpy array')
        elif delays.size != nel:
            raise ValueError('size of delays must be equal to the number of antennas')
        delays = NP.asarray(delays, dtype=NP.float64).ravel()
    elif 'pointing_center' in pointing_info:
        if 'pointing_coords' not in pointing_info:
            raise KeyError('pointing_coords not specified.')
        pc = NP.asarray(pointing_info['pointing_center'], dtype=NP.float64)
        if pointing_info['pointing_coords'] == 'altaz':
            pc = GEOM.altaz2dircos(pc.reshape(1, -1), 'degrees')
        elif pointing_info['pointing_coords'] == 'dircos':
            if NP.sum(pc ** 2) > 1.0 + 1e-9:
                raise ValueError('Invalid direction cosines specified in pointing_center')
            pc = pc.reshape(1, -1)
        else:
            raise ValueError('pointing_coords must be set to "dircos" or "altaz"')
        delays = (NP.dot(pos, pc.T) / 299792458.0).ravel()      # delay compensation: opposite sign to the geometric delay
    else:
        delays = NP.zeros(nel)
    gains = pointing_info.get('gains', None)
    if gains is None:
        gains = NP.ones(nel)
    elif not isinstance(gains, NP.ndarray):
        raise TypeError('gains must be a numpy array')
    elif gains.size != nel:
        raise ValueError('size of gains must be equal to the number of antennas')
    gains = NP.asarray(gains, dtype=NP.float64).ravel()
    delayerr = pointing_info.get('delayerr', None)
    if delayerr is not None:
        if not isinstance(delayerr, (int, float)):
            raise TypeError('delayerr must be an integer or float')
        if delayerr < 0.0:
            raise ValueError('delayerr must be non-negative')
        delays = delays.reshape(nel, 1) + delayerr * NP.random.standard_normal((nel, nrand))
    gainerr = pointing_info.get('gainerr', None)
    if gainerr is not None:
        if not isinstance(gainerr, (int, float)):
            raise TypeError('gainerr must be an integer or float')
        if gainerr < 0.0:
            raise ValueError('gainerr must be non-negative')
        gains = gains.reshape(nel, 1) * 10 ** (gainerr / 10.0 * NP.random.standard_normal((nel, nrand)))
    delays = NP.broadcast_to(delays.reshape(nel, -1), (nel, nrand)).copy()
    gains = NP.broadcast_to(gains.reshape(nel, -1), (nel, nrand)).copy()
    return delays, gains


# Polynomial power beams 1 + c0 x/1e3 + c1 x^2/1e7 + c2 x^3/1e10 [+ c3 x^4/1e13], x = (angle [arcmin] * f [GHz])^2:
# reference bands (GHz) and coefficients of the VLA (:491-499) and of the GMRT / upgraded GMRT (:781-790)
_POLY_BANDS = {
    'vla': ((0.0738, 0.3275, 1.465, 4.885, 8.435, 14.965, 22.485, 43.315),
            ((-0.897, 2.71, -0.242), (-0.935, 3.23, -0.378), (-1.343, 6.579, -1.186), (-1.372, 6.940, -1.309),
             (-1.306, 6.253, -1.100), (-1.305, 6.155, -1.030), (-1.417, 7.332, -1.352), (-1.321, 6.185, -0.983))),
    'gmrt': ((0.235, 0.325, 0.610, 1.420),
             ((-3.366, 46.159, -29.963, 7.529), (-3.397, 47.192, -30.931, 7.803), (-3.486, 47.749, -35.203, 10.399),
              (-2.27961, 21.4611, -9.7929, 1.80153))),
    'ugmrt': ((0.235, 0.325, 0.610, 1.420),
              ((NP.nan, NP.nan, NP.nan, NP.nan), (-2.939, 33.312, -16.659, 3.006), (-3.190, 38.642, -20.471, 3.964),
               (-2.608, 27.357, -13.091, 2.365))),
}


def poly_beam_coefficients(telescope_id, first_frequency_hz):
    """Coefficients of the band nearest to the FIRST frequency of the run (:500, :791), padded to 4."""
    key = 'vla' if telescope_id == 'vla' else telescope_id
    if key not in _POLY_BANDS:
        raise KeyError('no polynomial beam for instrument {0!r}'.format(telescope_id))     # the reference's parms_ref[instrument] lookup
    bands, coef = _POLY_BANDS[key]
    idx = int(NP.argmin(NP.abs(NP.asarray(bands) - first_frequency_hz / 1e9)))
    c = NP.zeros(4)
    c[:len(coef[idx])] = coef[idx]
    return c


def device_beam_spec(telescope, pointing_info=None, pointing_center=None, east2ax1=0.0, short_dipole_approx=False,
                     half_wave_dipole_approx=False, first_frequency_hz=None):
    """Map a reference ``telescope`` dictionary onto (beam_kind, size_m, element pointing dircos, ext) of the fused device
    beam kernel (ext: dipole axis / array factor / ground plane, see _abi.make_beam_ext).
    pointing_center: alt-az degrees (observe() passes pc_altaz, interferometry.py:6252)."""
    if (telescope is None) or (not isinstance(telescope, dict)):
        raise TypeError('telescope must be specified as a dictionary')
    if short_dipole_approx and half_wave_dipole_approx:
        raise ValueError('Both short dipole and half-wave dipole approximations cannot be made at the same time')
    dmode = _abi.PRISIM_DIPOLE_SHORT if short_dipole_approx else (_abi.PRISIM_DIPOLE_HALFWAVE if half_wave_dipole_approx
                                                                    else _abi.PRISIM_DIPOLE_GENERAL)
    zen = NP.array([0.0, 0.0, 1.0])
    ground = _ground_ext(telescope)
    tid = telescope.get('id', None)
    if tid is not None and tid not in ('custom',):
        if tid == 'vla' or 'gmrt' in tid:                                            # :225-238: angle from the zenith, no pointing
            if first_frequency_hz is None:
                raise ValueError('the polynomial beams select their band from the first frequency: first_frequency_hz is needed')
            ext = {'poly': poly_beam_coefficients(tid, first_frequency_hz)}
            if ground is not None:
                ext['ground'] = ground
            return _abi.PRISIM_BEAM_POLY, 0.0, zen, ext
        if tid in ('hera', 'hirax'):
            dia = 14.0 if tid == 'hera' else 6.0                                     # :240-243
            if 'orientation' in telescope:                                           # :245-246
                bpc = _pointing_dircos(NP.asarray(telescope['orientation']), telescope.get('ocoords', 'altaz'))
            else:
                bpc = zen
            ext = {'ground': ground} if ground is not None else None
            return _abi.PRISIM_BEAM_AIRY, dia, bpc, ext
        if tid == 'mwa':                                                             # :248-317
            if pointing_info is not None:                                            # :288-316: beamformer over the tile's dipoles
                locs = NP.asarray(telescope['element_locs'], dtype=NP.float64) if 'element_locs' in telescope else mwa_tile_element_locs()
                delays, gains = beamformer_settings(locs, pointing_info)
                ext = {'dipole_dircos': _dipole_axis(telescope), 'dipole_mode': dmode,
                       'beamformer': {'positions': locs, 'delays': delays, 'gains': gains}}
            else:
                ext = {'dipole_dircos': _dipole_axis(telescope), 'dipole_mode': dmode,
                       'array': {'nax1': 4, 'nax2': 4, 'sep1': 1.1, 'sep2': 1.1, 'east2ax1': east2ax1, 'pointing_dircos': zen}}   # :282-285
            if ground is not None:
                ext['ground'] = ground
            return _abi.PRISIM_BEAM_DIPOLE, 0.74, zen, ext                           # :267
        if tid in ('mwa_dipole', 'paper'):                                           # :320-349
            ext = {'dipole_dircos': _dipole_axis(telescope), 'dipole_mode': dmode}
            if ground is not None:
                ext['ground'] = ground
            return _abi.PRISIM_BEAM_DIPOLE, 0.74 if tid == 'mwa_dipole' else 2.0, zen, ext
        raise NotImplementedError('telescope id {0!r}: beam preset not on the accelerated path (SURVEY.md 8(f) N1)'.format(tid))
    shape = telescope.get('shape', 'delta')
    bpc = _pointing_dircos(pointing_center, 'altaz')
    ext = {'ground': ground} if ground is not None else None
    if pointing_info is not None and 'element_locs' in telescope:                    # :385-410 (no element_locs: factor 1, :387-389)
        locs = NP.asarray(telescope['element_locs'], dtype=NP.float64)
        delays, gains = beamformer_settings(locs, pointing_info)
        ext = dict(ext or {})
        ext['beamformer'] = {'positions': locs, 'delays': delays, 'gains': gains}
    if shape == 'delta':
        return _abi.PRISIM_BEAM_DELTA, 0.0, bpc, ext
    if shape == 'dish':
        return _abi.PRISIM_BEAM_AIRY, float(telescope['size']), bpc, ext
    if shape == 'gaussian':
        return _abi.PRISIM_BEAM_GAUSSIAN, float(telescope['size']), bpc, ext
    if shape == 'dipole':                                                            # :360-368
        ext = dict(ext or {})
        ext.update({'dipole_dircos': _pointing_dircos(NP.asarray(telescope['orientation']), telescope['ocoords']), 'dipole_mode': dmode})
        return _abi.PRISIM_BEAM_DIPOLE, float(telescope['size']), bpc, ext
    if shape in ('rect', 'square'):
        raise NotImplementedError('telescope shape {0!r} is not on the accelerated path (the reference code for it is broken, '
                                  'SURVEY.md Q17)'.format(shape))
    raise ValueError('Value in key "shape" of telescope dictionary invalid.')


def primary_beam_generator(skypos, frequency, telescope, freq_scale='GHz', skyunits='degrees', east2ax1=0.0,
                           pointing_info=None, pointing_center=None, short_dipole_approx=False,
                           half_wave_dipole_approx=False, device=0):
    """Power pattern (nsrc, nchan) at the given sky positions, evaluated on the GPU (:9-441).

    skyunits: 'altaz' (degrees) or 'dircos'.  frequency is scaled by freq_scale exactly like :212-217."""
    frequency = NP.asarray(frequency, dtype=NP.float64).ravel()
    if freq_scale in ('ghz', 'GHz'):                                                 # :212-217
        frequency = frequency * 1.0e9
    elif freq_scale in ('mhz', 'MHz'):
        frequency = frequency * 1.0e6
    elif freq_scale in ('khz', 'kHz'):
        frequency = frequency * 1.0e3
    skypos = NP.asarray(skypos, dtype=NP.float64)
    if skyunits == 'altaz':
        dircos = GEOM.altaz2dircos(skypos, 'degrees')
    elif skyunits == 'dircos':
        dircos = skypos.reshape(-1, 3)
    else:
        raise ValueError('skyunits must be "altaz" or "dircos" on the accelerated path')
    kind, dia, bpc, ext = device_beam_spec(telescope, pointing_info=pointing_info, pointing_center=pointing_center,
                                           east2ax1=east2ax1, short_dipole_approx=short_dipole_approx,
                                           half_wave_dipole_approx=half_wave_dipole_approx, first_frequency_hz=float(frequency[0]))
    nsrc = dircos.shape[0]
    with _abi.Context(device) as ctx:
        ctx.set_array(NP.zeros((1, 3)), frequency, nt_max=1)
        ctx.set_sky_analytic(dircos, NP.ones(nsrc), NP.zeros(nsrc), 1.0, kind, dia, bpc, NP.array([0.0, 0.0, 1.0]), ext=ext)
        return ctx.get_pbflux()


def spectral_interp_matrix(beam_freqs_hz, chans_hz, kind='cubic', chromatic=True, select_freq=None):
    """(nchan, nfreq) linear operator that resamples an external beam's frequency axis onto the channel grid, to be applied
    on the device to log10(beam) (prisim_hip_set_external_beam).  chromatic: scipy interp1d of the given kind applied to unit
    vectors (the interpolation is linear in the data; scripts/run_prisim.py:2094); achromatic: the single tabulated
    frequency nearest to select_freq is used for every channel (:2096-2097)."""
    bf = NP.asarray(beam_freqs_hz, dtype=NP.float64).ravel()
    ch = NP.asarray(chans_hz, dtype=NP.float64).ravel()
    if not chromatic:
        if select_freq is None:
            raise ValueError('select_freq is needed for an achromatic external beam')
        m = NP.zeros((ch.size, bf.size))
        m[:, int(NP.argmin(NP.abs(bf - select_freq)))] = 1.0
        return m
    if bf.size == 1:
        return NP.ones((ch.size, 1))
    if NP.any(NP.diff(bf) <= 0):
        raise ValueError('external beam frequencies must be strictly increasing')
    need = {'linear': 2, 'quadratic': 3, 'cubic': 4}.get(kind, 2)
    if bf.size < need:
        kind = 'linear'
    from scipy.interpolate import interp1d
    return interp1d(bf, NP.eye(bf.size), kind=kind, axis=0, bounds_error=False, fill_value='extrapolate', assume_sorted=True)(ch)
