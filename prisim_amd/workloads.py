"""Synthetic workloads of BASELINE.json's configs -- host side, numpy only.

Every workload is defined in the local ENU frame (the frame external beams use,
scripts/run_prisim.py:1898), so no astropy/healpy is needed:
  * channel grid  f_k = f0 + (k - nchan/2) df        scripts/run_prisim.py:900
  * point sources S = S0 (f/f_ref)^alpha             defaultparms.yaml skyparm.spindex = -0.83
  * diffuse sky   S = T 2k (f_ref/c)^2 Omega_pix/Jy   scripts/run_prisim.py:1220-1223, alpha = beta + 2
  * pixel "source shape" FWHM = nside2resol           scripts/run_prisim.py:1230-1246  (taper ON)
The returned dict holds only small per-source vectors (direction cosines, reference flux, spectral
index, FWHM); the (nsrc, nchan) beam x flux array is built on the GPU by
Context.set_sky_analytic (prisim_hip_set_sky_analytic).
"""
import numpy as NP

from . import geometry as GEOM
from . import layouts as LAY

K_BOLTZMANN = 1.380649e-23
C_LIGHT = 299792458.0
JY = 1.0e-26


def channel_grid(f0, df, nchan):
    """chans = f0 + (arange(nchan) - 0.5*nchan) * df   (scripts/run_prisim.py:900)."""
    return f0 + (NP.arange(nchan) - 0.5 * nchan) * df


def point_source_sky(nsrc, seed, alt_min_deg=10.0, f_ref=150e6, spindex=-0.83):
    """Random point sources above alt_min, uniform in solid angle; S0 ~ U(1,10) Jy at f_ref."""
    rng = NP.random.default_rng(seed)
    sin_alt = rng.uniform(NP.sin(NP.radians(alt_min_deg)), 1.0, nsrc)
    alt = NP.degrees(NP.arcsin(sin_alt))
    az = rng.uniform(0.0, 360.0, nsrc)
    flux = rng.uniform(1.0, 10.0, nsrc)
    return {
        'dircos': GEOM.altaz2dircos(NP.stack((alt, az), axis=1)),
        'altaz': NP.stack((alt, az), axis=1),
        'flux_ref': flux,
        'spindex': NP.full(nsrc, spindex),
        'ref_freq': f_ref,
        'fwhm_deg': NP.zeros(nsrc),      # MAJAX = MINAX = 0: src_shape is set, taper evaluates to 1
    }


def diffuse_sky(nside, seed, f_ref=150e6, spindex=-0.55, t_mean=300.0, t_sigma=0.5):
    """HEALPix RING pixels strictly above the horizon in the local frame (theta = zenith angle, phi = az)."""
    theta, phi = GEOM.healpix_pix2ang_ring(nside)
    rng = NP.random.default_rng(seed)
    temp = rng.lognormal(mean=NP.log(t_mean), sigma=t_sigma, size=theta.size)
    keep = theta < NP.pi / 2
    alt = 90.0 - NP.degrees(theta[keep])
    az = NP.degrees(phi[keep])
    omega = 4 * NP.pi / theta.size
    flux = temp[keep] * 2.0 * K_BOLTZMANN * (f_ref / C_LIGHT) ** 2 * omega / JY       # run_prisim.py:1220
    n = int(keep.sum())
    return {
        'dircos': GEOM.altaz2dircos(NP.stack((alt, az), axis=1)),
        'altaz': NP.stack((alt, az), axis=1),
        'flux_ref': flux,
        'spindex': NP.full(n, spindex),
        'ref_freq': f_ref,
        'fwhm_deg': NP.full(n, NP.degrees(GEOM.nside2resol(nside))),
    }


def concat_skies(*skies):
    out = {}
    for key in ('dircos', 'altaz', 'flux_ref', 'spindex', 'fwhm_deg'):
        out[key] = NP.concatenate([s[key] for s in skies], axis=0)
    out['ref_freq'] = skies[0]['ref_freq']
    if any(s['ref_freq'] != out['ref_freq'] for s in skies):
        raise ValueError('skies must share the reference frequency')
    return out


def config1():
    """3 baselines, 64 channels, 100 point sources, 1 snapshot, Gaussian D=14 m beam."""
    bl = NP.array([[14.6, 0.0, 0.0], [7.3, 12.644, 0.0], [29.2, 0.0, 0.0]])
    return {'name': 'cfg1: 3 bl x 64 ch x 100 src, Gaussian 14 m', 'baselines': bl,
            'channels': channel_grid(150e6, 390625.0, 64), 'sky': point_source_sky(100, 1),
            'beam': 'gaussian', 'diameter': 14.0, 'taper': True, 'precision': 'fp64'}


def config2():
    """HERA-19 (171 bl), 256 channels, nside=16 diffuse sky above the horizon, Airy 14 m, fp64."""
    bl, _ = LAY.layout_baselines('HERA-19')
    return {'name': 'cfg2: HERA-19 (171 bl) x 256 ch x nside16 diffuse, Airy 14 m', 'baselines': bl,
            'channels': channel_grid(150e6, 390625.0, 256), 'sky': diffuse_sky(16, 2),
            'beam': 'airy', 'diameter': 14.0, 'taper': True, 'precision': 'fp64'}


def config3(nsrc=10000, with_diffuse=False):
    """HERA-350 (61 075 bl), 1024 channels, 1e4 point sources (+ optional nside=128 diffuse), Airy 14 m, fp32."""
    bl, _ = LAY.layout_baselines('HERA-350')
    sky = point_source_sky(nsrc, 3)
    if with_diffuse:
        sky = concat_skies(sky, diffuse_sky(128, 33))
    return {'name': 'cfg3: HERA-350 (61075 bl) x 1024 ch x %d src, Airy 14 m' % sky['dircos'].shape[0], 'baselines': bl,
            'channels': channel_grid(150e6, 97656.25, 1024), 'sky': sky,
            'beam': 'airy', 'diameter': 14.0, 'taper': bool(with_diffuse), 'precision': 'fp32'}


def subsample(cfg, bl_stride=1, ch_count=None, src_stride=1):
    """Sub-sample a config (used for the bounded CPU baseline and for oracle-sized parity cases)."""
    out = dict(cfg)
    out['baselines'] = cfg['baselines'][::bl_stride]
    if ch_count is not None:
        out['channels'] = cfg['channels'][:ch_count]
    sky = cfg['sky']
    out['sky'] = {k: (v[::src_stride] if isinstance(v, NP.ndarray) else v) for k, v in sky.items()}
    return out


def mwa128_layout(seed=4, sigma_m=400.0, n_tiles=128):
    """Synthetic MWA-128T: the real tile file (prisim/data/array_layouts/MWA-I-128T_tile_coordinates.txt,
    interferometry.py:1801) is not in the reference tree; 128 tiles drawn from a 2-D Gaussian."""
    rng = NP.random.default_rng(seed)
    xy = rng.normal(0.0, sigma_m, size=(n_tiles, 2))
    return NP.hstack((xy, NP.zeros((n_tiles, 1))))


def synthetic_healpix_beam(nside, freqs_hz, tile=True):
    """External power beam [npix, nfreq] in the local (zenith angle, azimuth) frame: cos^2(theta) envelope times a 4x4
    array factor of 1.1 m spacing (MWA-tile-like), strictly positive (log10 is interpolated, run_prisim.py:2094)."""
    theta, phi = GEOM.healpix_pix2ang_ring(nside)
    freqs_hz = NP.asarray(freqs_hz, dtype=float)
    ct = NP.cos(NP.clip(theta, 0.0, NP.pi / 2))
    env = (ct ** 2)[:, None]
    if tile:
        l = NP.sin(theta) * NP.sin(phi)
        m = NP.sin(theta) * NP.cos(phi)
        lam = C_LIGHT / freqs_hz
        with NP.errstate(divide='ignore', invalid='ignore'):
            def af(x):
                ph = 2 * NP.pi * 1.1 * x[:, None] / lam[None, :]
                out = NP.sin(2.0 * ph) / NP.sin(0.5 * ph) / 4.0
                return NP.where(NP.abs(ph) < 1e-10, 1.0, out)
            env = env * (af(l) * af(m)) ** 2
    return env + 1e-6


def config4(n_acc=32):
    """MWA-128T (8128 bl), 768 channels, nside=64 diffuse sky (taper ON), external HEALPix beam nside=32, drift scan."""
    pos = mwa128_layout()
    bl, _ = LAY.baseline_generator(pos)
    bl = LAY.fold_and_sort_baselines(bl)
    ch = channel_grid(185e6, 40e3, 768)
    beam_freqs = NP.linspace(165e6, 205e6, 21)
    sky = diffuse_sky(64, 44, f_ref=185e6)
    return {'name': 'cfg4: MWA-128T (8128 bl) x 768 ch x nside64 diffuse x %d acc, external HEALPix beam nside32' % n_acc,
            'baselines': bl, 'channels': ch, 'sky': sky, 'beam': 'external', 'beam_table': synthetic_healpix_beam(32, beam_freqs),
            'beam_freqs': beam_freqs, 'taper': True, 'precision': 'fp32', 'n_acc': n_acc, 't_acc': 112.0, 'latitude': -26.701}


def config5(n_acc=120, nside=256):
    """HERA-350, 1024 channels, nside=256 diffuse sky (taper ON), Airy 14 m, 120 LSTs + delay transform."""
    bl, _ = LAY.layout_baselines('HERA-350')
    return {'name': 'cfg5: HERA-350 (61075 bl) x 1024 ch x nside%d diffuse x %d LST, Airy 14 m' % (nside, n_acc), 'baselines': bl,
            'channels': channel_grid(150e6, 97656.25, 1024), 'sky': diffuse_sky(nside, 55), 'beam': 'airy', 'diameter': 14.0,
            'taper': True, 'precision': 'fp32', 'n_acc': n_acc, 't_acc': 10.7, 'latitude': -30.7224}


def drift_snapshot_directions(sky, latitude, lst_offset_deg):
    """Directions of a sky that is fixed in (HA, Dec) at lst offset 0 after the sidereal sphere has turned by lst_offset_deg:
    returns (dircos, altaz, keep) with keep = above-horizon mask (the ROI selection of interferometry.py:6215-6216)."""
    hadec = GEOM.altaz2hadec(sky['altaz'], latitude, units='degrees')
    hadec = NP.stack((hadec[:, 0] + lst_offset_deg, hadec[:, 1]), axis=1)
    altaz = GEOM.hadec2altaz(hadec, latitude, units='degrees')
    keep = altaz[:, 0] > 0.0
    return GEOM.altaz2dircos(altaz[keep], 'degrees'), altaz[keep], keep
