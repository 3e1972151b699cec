#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by EXECUTING THE REFERENCE'S OWN STATEMENTS.

Runs only in the build container, where /root/reference is mounted read-only; the GPU box never
sees the reference, only the small .npz fixtures this script writes (inputs + expected outputs).

The reference package cannot be imported (Python-2 syntax elsewhere in the files, un-vendored
astroutils/astropy -- SURVEY.md 8(c)), but the statements of the hot path are plain numpy and
run unchanged under Python 3.  This script reads the cited line ranges from the reference
source AT GENERATION TIME, dedents them and exec()s them on seeded inputs.  No reference
source text is stored in this repository.

  golden_skyvis.npz, golden_skyvis_long.npz (the same statements on 2.5 km baselines)
                       interferometry.py:6332,6340 (fp64), 6335 (+taper), 6338/6343 (gradient),
                       6323,6326,6327,6330 (fp32 "memsave"), taper 6259-6262,6265-6270,6281-6283,
                       baseline_delay_horizon.py:133-241 (function geometric_delay, dircos path),
                       phase_centering :7871-7872, 7877
  golden_beams.npz     primary_beams.py:517-625 (airy_disk_pattern), 629-730 (gaussian_beam),
                       9-441 (primary_beam_generator dispatch, shapes 'gaussian' / 'dish' / 'delta')
  golden_beams_ext.npz primary_beams.py:975-1235 (dipole_field_pattern), 1239-1478 (isotropic_radiators_array_field_pattern),
                       9-441 (dispatch for id 'mwa' / 'mwa_dipole' / 'paper' and shape 'dipole'), direction-cosine inputs
  golden_polybeams.npz primary_beams.py:445-513 (VLA_primary_beam_PBCOR), 734-808 (GMRT_primary_beam), 9-441 (id 'vla' / 'gmrt' / 'ugmrt')
  golden_beamformer.npz primary_beams.py:1482-1754 (array_field_pattern: beamformer delays / gains / pointing centre / seeded
                       delay and gain jitter, complex64 arithmetic), 9-441 (id 'mwa' and shape 'dipole' with pointing_info)
  golden_apply_gradients.npz  interferometry.py:6726-6819 (method apply_gradients, called on a stand-in object that holds the
                       attributes it reads: gradient, gradient_mode, channels, labels, lst)
  golden_aux.npz       the last executable statements next to the path:
                       scripts/run_prisim.py:2099-2103 (log-beam normalisation after the HEALPix interpolation) + interferometry.py:4466
                       (the supplied beam stored as float32); interferometry.py:7980-7985 (uvw rotation matrix and projection, on given
                       equatorial baselines and (HA, Dec) in radians); :8035-8045 (conjugate: flipped baselines, conjugated cubes, labels);
                       :6676-6691 (vis_rms_freq, flux_unit 'JY' and 'K'; CNST.Jy of the un-vendored astroutils = 1e-26 W m^-2 Hz^-1)
"""
import os
import sys
import textwrap
import types

import numpy as NP
import scipy.constants as FCNST
import scipy.special as SPS

REF = '/root/reference/prisim'
HERE = os.path.dirname(os.path.abspath(__file__))


def _pick(fname, ranges):
    with open(os.path.join(REF, fname)) as f:
        lines = f.readlines()
    picked = []
    for lo, hi in ranges:
        picked += lines[lo - 1:hi]
    return picked


def ref_block(fname, ranges):
    """Source text of a contiguous block (a whole function definition), dedented as a unit."""
    return textwrap.dedent(''.join(_pick(fname, ranges)))


def ref_stmts(fname, ranges):
    """Source text of single-line statements picked from different nesting levels: each line is
    stripped of its own indentation (every picked line is one complete statement)."""
    return ''.join(line.lstrip() for line in _pick(fname, ranges))


def altaz2dircos(altaz):
    alt = NP.radians(altaz[:, 0])
    az = NP.radians(altaz[:, 1])
    return NP.stack((NP.cos(alt) * NP.sin(az), NP.cos(alt) * NP.cos(az), NP.sin(alt)), axis=1)


def make_skyvis(fname='golden_skyvis.npz', seed=20261003, nsrc=37, nbl=9, nchan=24, maxbl=150.0, f0=150e6, df=390625.0):
    rng = NP.random.default_rng(seed)
    baselines = rng.uniform(-maxbl, maxbl, size=(nbl, 3))
    baselines[:, 2] *= 0.02
    channels = f0 + (NP.arange(nchan) - nchan // 2) * df
    alt = NP.degrees(NP.arcsin(rng.uniform(0.1, 1.0, nsrc)))
    az = rng.uniform(0.0, 360.0, nsrc)
    skypos_dircos_roi = altaz2dircos(NP.stack((alt, az), axis=1))
    pbfluxes = rng.uniform(0.1, 10.0, size=(nsrc, 1)) * (channels / 150e6).reshape(1, -1) ** -0.83 \
        * rng.uniform(0.3, 1.0, size=(nsrc, nchan))
    pc_dircos = altaz2dircos(NP.array([[78.0, 40.0]]))
    src_shape = NP.stack((rng.uniform(0.05, 1.0, nsrc), rng.uniform(0.05, 1.0, nsrc), NP.zeros(nsrc)), axis=1)
    src_shape[::6, :2] = 0.0        # zero-size sources: sigma = inf, w = 1 (divide-by-zero warning in the reference)

    # geometric_delay(): the reference function itself (dircos path uses only NP and FCNST)
    ns = {'NP': NP, 'FCNST': FCNST, 'GEOM': None}
    exec(ref_block('baseline_delay_horizon.py', [(133, 241)]), ns)
    geometric_delays = ns['geometric_delay'](baselines, skypos_dircos_roi, altaz=False, hadec=False, dircos=True)
    pc_delay_offsets = ns['geometric_delay'](baselines, pc_dircos, altaz=False, hadec=False, dircos=True)

    self_ = types.SimpleNamespace(geometric_delays=[geometric_delays], channels=channels,
                                  baseline_lengths=NP.sqrt(NP.sum(baselines ** 2, axis=1)),   # :5684
                                  baselines=baselines)
    skymodel = types.SimpleNamespace(src_shape=src_shape)
    m2 = NP.arange(nsrc)
    env = {'NP': NP, 'FCNST': FCNST, 'self': self_, 'skymodel': skymodel, 'm2': m2,
           'geometric_delays': geometric_delays, 'pc_delay_offsets': pc_delay_offsets, 'pbfluxes': pbfluxes,
           'skypos_dircos_roi': skypos_dircos_roi}

    out = {}
    # ---- fp64, no taper: :6332 then :6340, gradient :6343
    e = dict(env)
    exec(ref_stmts('interferometry.py', [(6332, 6332), (6340, 6340), (6343, 6343)]), e)
    out['skyvis_f64'] = e['skyvis']
    out['grad_f64'] = e['skyvis_gradient']
    # ---- taper weights :6259-6262, 6265, 6267-6268, 6270, 6281, 6283
    with NP.errstate(divide='ignore', invalid='ignore'):
        e = dict(env)
        exec(ref_stmts('interferometry.py', [(6259, 6262), (6265, 6265), (6267, 6268), (6270, 6270), (6281, 6281), (6283, 6283)]), e)
    vis_wts = e['vis_wts']
    out['vis_wts'] = vis_wts
    # ---- fp64 with taper :6332, 6335, 6338
    e = dict(env, vis_wts=vis_wts)
    exec(ref_stmts('interferometry.py', [(6332, 6332), (6335, 6335), (6338, 6338)]), e)
    out['skyvis_f64_taper'] = e['skyvis']
    out['grad_f64_taper'] = e['skyvis_gradient']
    # ---- fp32 ("memsave") :6286 cast, :6287 delays cast, :6323, 6327, 6330 ; with taper also :6289, 6326
    e = dict(env)
    e['pbfluxes'] = pbfluxes.astype(NP.float32)                                   # :6286
    e['self'] = types.SimpleNamespace(geometric_delays=[geometric_delays.astype(NP.float32)], channels=channels)  # :6287
    e['pc_delay_offsets'] = pc_delay_offsets.astype(NP.float32)                   # :6167
    exec(ref_stmts('interferometry.py', [(6323, 6323), (6327, 6327), (6330, 6330)]), e)
    out['skyvis_f32'] = e['skyvis']
    out['grad_f32'] = e['skyvis_gradient']
    e = dict(env)
    e['pbfluxes'] = pbfluxes.astype(NP.float32)
    e['self'] = types.SimpleNamespace(geometric_delays=[geometric_delays.astype(NP.float32)], channels=channels)
    e['pc_delay_offsets'] = pc_delay_offsets.astype(NP.float32)
    e['vis_wts'] = vis_wts.astype(NP.float32)                                     # :6289
    exec(ref_stmts('interferometry.py', [(6323, 6323), (6326, 6327)]), e)
    out['skyvis_f32_taper'] = e['skyvis']

    # ---- phase centering (rotate visibilities to a new phase centre): :7871-7872, :7877
    n_t = 3
    cube = (rng.normal(size=(nbl, nchan, n_t)) + 1j * rng.normal(size=(nbl, nchan, n_t)))
    pc_cur = altaz2dircos(NP.stack((rng.uniform(60, 90, n_t), rng.uniform(0, 360, n_t)), axis=1))
    pc_new = altaz2dircos(NP.stack((rng.uniform(60, 90, n_t), rng.uniform(0, 360, n_t)), axis=1))
    e = {'NP': NP, 'FCNST': FCNST, 'phase_center_current_temp': pc_cur, 'phase_center_new': pc_new,
         'self': types.SimpleNamespace(baselines=baselines, channels=channels, skyvis_freq=cube.copy())}
    exec(ref_stmts('interferometry.py', [(7871, 7872), (7877, 7877)]), e)
    out['phase_cube_in'] = cube
    out['phase_pc_cur'] = pc_cur
    out['phase_pc_new'] = pc_new
    out['phase_cube_out'] = e['self'].skyvis_freq

    NP.savez_compressed(os.path.join(HERE, fname),
                        baselines=baselines, channels=channels, dircos=skypos_dircos_roi, pbfluxes=pbfluxes,
                        pc_dircos=pc_dircos.ravel(), src_shape=src_shape, geometric_delays=geometric_delays,
                        pc_delay_offsets=pc_delay_offsets, **out)
    print(fname + ':', {k: v.shape for k, v in out.items()})


def make_beams():
    rng = NP.random.default_rng(77)
    nsrc, nchan = 41, 12
    alt = NP.concatenate(([90.0, 89.999999, 0.0, -5.0], NP.degrees(NP.arcsin(rng.uniform(0.0, 1.0, nsrc - 4)))))
    az = rng.uniform(0.0, 360.0, nsrc)
    skypos = NP.stack((alt, az), axis=1)
    freq_hz = 150e6 + (NP.arange(nchan) - nchan // 2) * 4e6
    ns = {'NP': NP, 'FCNST': FCNST, 'SPS': SPS, 'GEOM': None}
    exec(ref_block('primary_beams.py', [(517, 625)]), ns)      # airy_disk_pattern
    exec(ref_block('primary_beams.py', [(629, 730)]), ns)      # gaussian_beam
    exec(ref_block('primary_beams.py', [(9, 441)]), ns)        # primary_beam_generator (dispatch)
    out = {}
    with NP.errstate(divide='ignore', invalid='ignore'):
        out['airy_power_d14'] = ns['airy_disk_pattern'](14.0, skypos, freq_hz, skyunits='altaz', peak=1.0,
                                                        pointing_center=None, power=True, small_angle_tol=1e-10)
        out['gauss_power_d14'] = ns['gaussian_beam'](14.0, skypos, freq_hz, skyunits='altaz', pointing_center=None, power=True)
        # dispatcher: frequency passed in GHz with freq_scale='GHz' exactly like observe() does (:6252)
        out['pbg_gaussian_d14'] = ns['primary_beam_generator'](skypos, freq_hz / 1e9, {'shape': 'gaussian', 'size': 14.0},
                                                               freq_scale='GHz', skyunits='altaz')
        out['pbg_dish_d14'] = ns['primary_beam_generator'](skypos, freq_hz / 1e9, {'shape': 'dish', 'size': 14.0},
                                                           freq_scale='GHz', skyunits='altaz')
        out['pbg_delta'] = ns['primary_beam_generator'](skypos, freq_hz / 1e9, {'shape': 'delta'},
                                                        freq_scale='GHz', skyunits='altaz')
    NP.savez_compressed(os.path.join(HERE, 'golden_beams.npz'), skypos_altaz=skypos, freq_hz=freq_hz, **out)
    print('golden_beams.npz:', {k: v.shape for k, v in out.items()})

    # ---- dipole, 4x4 array factor, MWA / PAPER presets: direction-cosine inputs (those code paths need no GEOM)
    rng = NP.random.default_rng(78)
    n = 37
    alt = NP.concatenate(([90.0, 0.0], NP.degrees(NP.arcsin(rng.uniform(0.02, 1.0, n - 2)))))
    az = NP.concatenate(([0.0, 90.0], rng.uniform(0.0, 360.0, n - 2)))
    dircos = altaz2dircos(NP.stack((alt, az), axis=1))
    dircos[1] = [1.0, 0.0, 0.0]                       # exactly along an east-pointing dipole: L'Hospital branch (:1226)
    wl = FCNST.c / freq_hz
    exec(ref_block('primary_beams.py', [(975, 1235)]), ns)     # dipole_field_pattern
    exec(ref_block('primary_beams.py', [(1239, 1478)]), ns)    # isotropic_radiators_array_field_pattern
    out2 = {}
    east = NP.asarray([1.0, 0.0, 0.0]).reshape(1, -1)
    tilt = altaz2dircos(NP.array([[20.0, 35.0]]))
    with NP.errstate(divide='ignore', invalid='ignore'):
        for name, kw in (('general', {'short_dipole_approx': False, 'half_wave_dipole_approx': False}),
                         ('short', {'short_dipole_approx': True, 'half_wave_dipole_approx': False}),
                         ('halfwave', {'short_dipole_approx': False, 'half_wave_dipole_approx': True})):
            out2['dipole_field_' + name] = ns['dipole_field_pattern'](0.74, NP.copy(dircos), dipole_coords='dircos',
                                                                       dipole_orientation=NP.copy(east), skycoords='dircos',
                                                                       wavelength=wl, power=False, **kw)
        out2['dipole_field_tilted_2m'] = ns['dipole_field_pattern'](2.0, NP.copy(dircos), dipole_coords='dircos', dipole_orientation=NP.copy(tilt),
                                                                    skycoords='dircos', wavelength=wl, short_dipole_approx=False, half_wave_dipole_approx=False, power=False)
        out2['irap_4x4_zenith'] = ns['isotropic_radiators_array_field_pattern'](4, 4, 1.1, 1.1, NP.copy(dircos), wl, east2ax1=0.0,
                                                                                 pointing_center=NP.asarray([0.0, 0.0, 1.0]), skycoords='dircos', power=False)
        pc = altaz2dircos(NP.array([[70.0, 120.0]])).ravel()
        out2['irap_4x4_rot30_pointed'] = ns['isotropic_radiators_array_field_pattern'](4, 4, 1.1, 1.1, NP.copy(dircos), wl, east2ax1=30.0,
                                                                                        pointing_center=NP.copy(pc), skycoords='dircos', power=False)
        out2['pbg_mwa'] = ns['primary_beam_generator'](NP.copy(dircos), freq_hz / 1e9, {'id': 'mwa'}, freq_scale='GHz', skyunits='dircos', east2ax1=0.0)
        out2['pbg_mwa_dipole'] = ns['primary_beam_generator'](NP.copy(dircos), freq_hz / 1e9, {'id': 'mwa_dipole'}, freq_scale='GHz', skyunits='dircos')
        out2['pbg_paper'] = ns['primary_beam_generator'](NP.copy(dircos), freq_hz / 1e9, {'id': 'paper'}, freq_scale='GHz', skyunits='dircos')
        out2['pbg_shape_dipole'] = ns['primary_beam_generator'](NP.copy(dircos), freq_hz / 1e9,
                                                               {'shape': 'dipole', 'size': 1.5, 'ocoords': 'dircos', 'orientation': NP.copy(tilt)},
                                                               freq_scale='GHz', skyunits='dircos')
    NP.savez_compressed(os.path.join(HERE, 'golden_beams_ext.npz'), dircos=dircos, freq_hz=freq_hz, tilt=tilt.ravel(), array_pc=pc, **out2)
    print('golden_beams_ext.npz:', {k: v.shape for k, v in out2.items()})


def make_beamformer():
    """Phased-array beamformer (primary_beams.py:1482-1754) and its callers in primary_beam_generator (:288-317, :385-416)."""
    rng = NP.random.default_rng(79)
    n, nchan = 29, 7
    alt = NP.concatenate(([90.0], NP.degrees(NP.arcsin(rng.uniform(0.05, 1.0, n - 1)))))
    az = NP.concatenate(([0.0], rng.uniform(0.0, 360.0, n - 1)))
    dircos = altaz2dircos(NP.stack((alt, az), axis=1))
    freq_hz = 185e6 + (NP.arange(nchan) - nchan // 2) * 2.56e6
    wl = FCNST.c / freq_hz
    ns = {'NP': NP, 'FCNST': FCNST, 'SPS': SPS, 'GEOM': None}
    exec(ref_block('primary_beams.py', [(975, 1235)]), ns)     # dipole_field_pattern
    exec(ref_block('primary_beams.py', [(1239, 1478)]), ns)    # isotropic_radiators_array_field_pattern
    exec(ref_block('primary_beams.py', [(1482, 1754)]), ns)    # array_field_pattern
    exec(ref_block('primary_beams.py', [(9, 441)]), ns)        # primary_beam_generator
    afp = ns['array_field_pattern']
    xl, yl = NP.meshgrid(1.1 * NP.linspace(-1.5, 1.5, 4), 1.1 * NP.linspace(1.5, -1.5, 4))           # the MWA tile of :290-291
    tile = NP.hstack((xl.reshape(-1, 1), yl.reshape(-1, 1), NP.zeros(xl.size).reshape(-1, 1)))
    irregular = NP.hstack((rng.uniform(-3.0, 3.0, (11, 2)), rng.uniform(-0.1, 0.1, (11, 1))))
    delays = rng.uniform(-4e-9, 4e-9, 16)
    gains = rng.uniform(0.7, 1.2, 16)
    pc = altaz2dircos(NP.array([[72.0, 140.0]])).ravel()
    out = {}
    with NP.errstate(divide='ignore', invalid='ignore'):
        out['field_delays_gains'] = afp(NP.copy(tile), NP.copy(dircos), skycoords='dircos', pointing_info={'delays': NP.copy(delays), 'gains': NP.copy(gains)},
                                        wavelength=NP.copy(wl), power=False)
        out['power_pointed'] = afp(NP.copy(tile), NP.copy(dircos), skycoords='dircos',
                                   pointing_info={'pointing_center': NP.copy(pc), 'pointing_coords': 'dircos'}, wavelength=NP.copy(wl), power=True)
        out['field_irregular_none'] = afp(NP.copy(irregular), NP.copy(dircos), skycoords='dircos', pointing_info=None, wavelength=NP.copy(wl), power=False)
        NP.random.seed(5)
        out['field_jitter_seed5'] = afp(NP.copy(tile), NP.copy(dircos), skycoords='dircos',
                                        pointing_info={'pointing_center': NP.copy(pc), 'pointing_coords': 'dircos', 'delayerr': 0.3e-9, 'gainerr': 0.5, 'nrand': 3},
                                        wavelength=NP.copy(wl), power=False)
        out['pbg_mwa_delays'] = ns['primary_beam_generator'](NP.copy(dircos), freq_hz / 1e9, {'id': 'mwa'}, freq_scale='GHz', skyunits='dircos',
                                                             pointing_info={'delays': NP.copy(delays), 'gains': NP.copy(gains)})
        NP.random.seed(6)
        out['pbg_mwa_jitter_seed6'] = ns['primary_beam_generator'](NP.copy(dircos), freq_hz / 1e9, {'id': 'mwa'}, freq_scale='GHz', skyunits='dircos',
                                                                   pointing_info={'pointing_center': NP.copy(pc), 'pointing_coords': 'dircos', 'delayerr': 0.2e-9,
                                                                                  'gainerr': 0.3, 'nrand': 4})
        tilt = altaz2dircos(NP.array([[20.0, 35.0]]))
        out['pbg_dipole_elements_pointed'] = ns['primary_beam_generator'](
            NP.copy(dircos), freq_hz / 1e9, {'shape': 'dipole', 'size': 1.5, 'ocoords': 'dircos', 'orientation': NP.copy(tilt), 'element_locs': NP.copy(irregular)},
            freq_scale='GHz', skyunits='dircos', pointing_info={'pointing_center': NP.copy(pc), 'pointing_coords': 'dircos'})
    NP.savez_compressed(os.path.join(HERE, 'golden_beamformer.npz'), dircos=dircos, freq_hz=freq_hz, tile=tile, irregular=irregular, delays=delays,
                        gains=gains, pc=pc, tilt=tilt.ravel(), **out)
    print('golden_beamformer.npz:', {k: (v.shape, v.dtype) for k, v in out.items()})


def make_polybeams():
    """Polynomial (PBCOR-style) dish beams of the VLA and the GMRT (primary_beams.py:445-513, 734-808) and their dispatch (:225-238)."""
    rng = NP.random.default_rng(80)
    ns = {'NP': NP, 'FCNST': FCNST, 'SPS': SPS, 'GEOM': None}
    exec(ref_block('primary_beams.py', [(445, 513)]), ns)
    exec(ref_block('primary_beams.py', [(734, 808)]), ns)
    exec(ref_block('primary_beams.py', [(9, 441)]), ns)
    out = {}
    n = 33
    for name, tel, f0 in (('vla_L', {'id': 'vla'}, 1.4e9), ('vla_P', {'id': 'vla'}, 0.33e9), ('gmrt_610', {'id': 'gmrt'}, 0.6e9), ('ugmrt_325', {'id': 'ugmrt'}, 0.32e9)):
        freq_hz = f0 + NP.arange(5) * 2e6
        # zenith angles where the polynomial is a beam (first-null radius ~ 45' * 1.4 GHz / f for a 25 m dish, 1.8x that for 45 m)
        theta_max = (0.55 if 'vla' in name else 0.3) * 1.4e9 / f0
        alt = 90.0 - NP.concatenate(([0.0], rng.uniform(0.0, theta_max, n - 1)))
        skypos = NP.stack((alt, rng.uniform(0.0, 360.0, n)), axis=1)
        out['pbg_' + name] = ns['primary_beam_generator'](NP.copy(skypos), freq_hz / 1e9, dict(tel), freq_scale='GHz', skyunits='altaz')
        out['altaz_' + name] = skypos
        out['freq_' + name] = freq_hz
    NP.savez_compressed(os.path.join(HERE, 'golden_polybeams.npz'), **out)
    print('golden_polybeams.npz:', {k: v.shape for k, v in out.items() if k.startswith('pbg')}, {k: (float(v.min()), float(v.max())) for k, v in out.items() if k.startswith('pbg')})


def make_apply_gradients():
    import warnings
    rng = NP.random.default_rng(20261004)
    nbl, nchan, nt = 7, 12, 3
    channels = 150e6 + (NP.arange(nchan) - nchan // 2) * 390625.0
    gradient = rng.normal(size=(3, nbl, nchan, nt)) + 1j * rng.normal(size=(3, nbl, nchan, nt))
    ns = {'NP': NP, 'FCNST': FCNST, 'warnings': warnings}
    exec(ref_block('interferometry.py', [(6726, 6819)]), ns)
    self_ = types.SimpleNamespace(gradient_mode='baseline', gradient={'baseline': gradient}, channels=channels,
                                  labels=NP.arange(nbl), lst=NP.arange(nt))
    out = {'gradient': gradient, 'channels': channels}
    cases = {'seeds3': rng.normal(scale=0.05, size=(4, 3, nbl)),         # nseed x 3 x nbl
             'plain2d': rng.normal(scale=0.05, size=(3, nbl)),           # one realisation, no seed axis
             'grid5d': rng.normal(scale=0.05, size=(2, 3, 3, nbl)),      # (n1, n2, 3, nbl)
             'xy_only': rng.normal(scale=0.05, size=(2, 2, nbl)),        # z perturbation missing -> zero, with a warning
             'x_only': rng.normal(scale=0.05, size=(2, 1, nbl)),
             'four_axes': rng.normal(scale=0.05, size=(2, 4, nbl))}      # fourth axis dropped, with a warning
    for name, pert in cases.items():
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            res = ns['apply_gradients'](self_, gradient_mode='baseline', perturbations={'baseline': pert.copy()})
        out['pert_' + name] = pert
        out['delta_' + name] = res
    NP.savez_compressed(os.path.join(HERE, 'golden_apply_gradients.npz'), **out)
    print('golden_apply_gradients.npz:', {k: v.shape for k, v in out.items()})


def make_aux():
    rng = NP.random.default_rng(20261006)
    out = {}
    # ---- run_prisim.py:2098-2102 + interferometry.py:4466: normalisation of the interpolated log-beam, stored as float32 ----
    nsrc, nchan = 23, 9
    interp_logbeam = rng.uniform(-6.0, 0.4, size=(nsrc, nchan))
    interp_logbeam[:, 3] -= 1.0                       # a channel whose maximum is below 0: the clamp of :2100 keeps it un-normalised
    interp_logbeam[5, 6] = NP.nan                     # nanmax ignores it (:2099); the NaN itself propagates
    ns = {'NP': NP, 'interp_logbeam': interp_logbeam.copy(), 'roiinfo': {}}
    exec(ref_stmts('../scripts/run_prisim.py', [(2099, 2103)]), ns)
    out['logbeam_in'] = interp_logbeam
    out['pbeam'] = ns['roiinfo']['pbeam']
    roi_info = {'pbeam': ns['roiinfo']['pbeam']}
    holder = types.SimpleNamespace(info={'pbeam': []})
    exec(ref_stmts('interferometry.py', [(4466, 4466)]), {'NP': NP, 'self': holder, 'roi_info': roi_info})
    out['pbeam_f32'] = holder.info['pbeam'][0]
    # ---- interferometry.py:7980-7985: the uvw rotation matrix and the projection ----
    nbl, nt = 6, 4
    eq_baselines = rng.uniform(-300.0, 300.0, size=(nbl, 3))
    ha = NP.radians(rng.uniform(-80.0, 80.0, nt))
    dec = NP.radians(rng.uniform(-70.0, 40.0, nt))
    holder = types.SimpleNamespace()
    exec(ref_block('interferometry.py', [(7980, 7985)]), {'NP': NP, 'self': holder, 'ha': ha, 'dec': dec, 'eq_baselines': eq_baselines})
    out.update(proj_eq_baselines=eq_baselines, proj_ha=ha, proj_dec=dec, projected_baselines=holder.projected_baselines)
    exec(ref_block('interferometry.py', [(7980, 7985)]), {'NP': NP, 'self': holder, 'ha': ha[:1], 'dec': dec[:1], 'eq_baselines': eq_baselines})
    out['projected_baselines_one'] = holder.projected_baselines
    # ---- interferometry.py:8035-8045: conjugate ----
    nchan2 = 5
    def cube():
        return rng.normal(size=(nbl, nchan2, nt)) + 1j * rng.normal(size=(nbl, nchan2, nt))
    holder = types.SimpleNamespace(labels=[(i, 100 + i) for i in range(nbl)], baselines=rng.uniform(-100, 100, size=(nbl, 3)),
                                   vis_freq=cube(), skyvis_freq=cube(), vis_noise_freq=cube(), projected_baselines=rng.normal(size=(nbl, 3, nt)))
    ind = NP.asarray([1, 4, 5])
    out.update(conj_ind=ind, conj_baselines_in=holder.baselines.copy(), conj_vis_in=holder.vis_freq.copy(), conj_skyvis_in=holder.skyvis_freq.copy(),
               conj_noise_in=holder.vis_noise_freq.copy(), conj_proj_in=holder.projected_baselines.copy())
    exec(ref_block('interferometry.py', [(8035, 8045)]), {'NP': NP, 'self': holder, 'ind': ind, 'xrange': range})
    out.update(conj_baselines=holder.baselines, conj_orientations=holder.baseline_orientations, conj_vis=holder.vis_freq,
               conj_skyvis=holder.skyvis_freq, conj_noise=holder.vis_noise_freq, conj_proj=holder.projected_baselines,
               conj_labels=NP.array([list(l) for l in holder.labels]))
    # ---- interferometry.py:6676-6691: vis_rms_freq ----
    CNST = types.SimpleNamespace(Jy=1.0e-26)
    nchan3 = 7
    for unit in ('JY', 'K'):
        holder = types.SimpleNamespace(eff_Q=rng.uniform(0.8, 0.95, size=(nbl, nchan3)), A_eff=rng.uniform(100.0, 200.0, size=(nbl, nchan3)),
                                       t_acc=[10.0, 10.0, 12.5, 8.0], flux_unit=unit, freq_resolution=97656.25,
                                       Tsys=rng.uniform(80.0, 400.0, size=(nbl, nchan3, nt)))
        exec(ref_block('interferometry.py', [(6676, 6691)]), {'NP': NP, 'FCNST': FCNST, 'CNST': CNST, 'self': holder})
        out.update({'rms_effQ_' + unit: holder.eff_Q, 'rms_Aeff_' + unit: holder.A_eff, 'rms_tacc_' + unit: NP.asarray(holder.t_acc),
                    'rms_Tsys_' + unit: holder.Tsys, 'rms_df': NP.asarray(holder.freq_resolution), 'rms_out_' + unit: holder.vis_rms_freq})
    NP.savez_compressed(os.path.join(HERE, 'golden_aux.npz'), **out)
    print('golden_aux.npz:', {k: v.shape for k, v in out.items()})


if __name__ == '__main__':
    if not os.path.isdir(REF):
        sys.exit('reference tree not available: golden vectors can only be regenerated in the build container')
    make_skyvis()
    # the same statements on an MWA-like case: baselines to 2.5 km (delays to ~8 us, ~1600 cycles of phase), 40 kHz channels at 185 MHz
    make_skyvis('golden_skyvis_long.npz', seed=20261005, nsrc=101, nbl=21, nchan=32, maxbl=2500.0, f0=185e6, df=40e3)
    make_beams()
    make_beamformer()
    make_polybeams()
    make_apply_gradients()
    make_aux()
