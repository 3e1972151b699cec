"""Host mirror of the reference's primary-beam dispatcher for the leaves on the accelerated path.

``primary_beam_generator`` keeps the reference's name and argument meaning
(prisim/primary_beams.py:9-441); the patterns themselves are evaluated ON THE GPU by the fused
beam kernel (prisim_amd/csrc/aux_kernels.hip, prisim_hip_set_sky_analytic):
    telescope id 'hera' / 'hirax'        -> Airy power pattern, D = 14 m / 6 m     (:239-247)
    telescope shape 'dish'               -> Airy power pattern of diameter 'size'  (:369-373, :416)
    telescope shape 'gaussian'           -> Gaussian power pattern                 (:374-377, :416)
    telescope shape 'delta' / no shape   -> 1                                      (:355-359, :416)
Other presets (vla, gmrt, mwa, dipole, phased arrays, rect/square apertures, ground planes) are
later rows of SURVEY.md 8(f) and raise NotImplementedError here -- there is no CPU stand-in.
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


def device_beam_spec(telescope, pointing_info=None, pointing_center=None):
    """Map a reference ``telescope`` dictionary onto (beam_kind, diameter_m, beam pointing dircos) of the
    fused device beam kernel.  pointing_center: alt-az degrees (observe() passes pc_altaz, :6252)."""
    if (telescope is None) or (not isinstance(telescope, dict)):
        raise TypeError('telescope must be specified as a dictionary')
    if telescope.get('groundplane', None) is not None and telescope.get('shape', None) != 'dish':
        raise NotImplementedError('ground-plane patterns (primary_beams.py:812-971) are not on the accelerated path yet')
    if 'id' in telescope:
        tid = telescope['id']
        if tid in ('hera', 'hirax'):
            dia = 14.0 if tid == 'hera' else 6.0                                     # :240-243
            if 'orientation' in telescope:                                           # :245-246
                bpc = _pointing_dircos(NP.asarray(telescope['orientation']), telescope.get('ocoords', 'altaz'))
            else:
                bpc = NP.array([0.0, 0.0, 1.0])
            return _abi.PRISIM_BEAM_AIRY, dia, bpc
        if tid in ('custom', None) and 'shape' in telescope:
            pass
        else:
            raise NotImplementedError('telescope id {0!r}: beam preset not on the accelerated path (SURVEY.md 8(f) N1)'.format(tid))
    if pointing_info is not None:
        raise NotImplementedError('phased-array beamformer (pointing_info) is not on the accelerated path yet')
    shape = telescope.get('shape', 'delta')
    bpc = _pointing_dircos(pointing_center, 'altaz')
    if shape == 'delta':
        return _abi.PRISIM_BEAM_DELTA, 0.0, bpc
    if shape == 'dish':
        return _abi.PRISIM_BEAM_AIRY, float(telescope['size']), bpc
    if shape == 'gaussian':
        return _abi.PRISIM_BEAM_GAUSSIAN, float(telescope['size']), bpc
    if shape in ('dipole', 'rect', 'square'):
        raise NotImplementedError('telescope shape {0!r} is not on the accelerated path yet'.format(shape))
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
    kind, dia, bpc = device_beam_spec(telescope, pointing_info=pointing_info, pointing_center=pointing_center)
    nsrc = dircos.shape[0]
    with _abi.Context(device) as ctx:
        ctx.set_array(NP.zeros((1, 3)), frequency, nt_max=1)
        ctx.set_sky_analytic(dircos, NP.ones(nsrc), NP.zeros(nsrc), 1.0, kind, dia, bpc, NP.array([0.0, 0.0, 1.0]))
        return ctx.get_pbflux()
