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


CONFIGS = {
    # name: (description, builder)
}


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
            'beam': 'airy', 'diameter': 14.0, 'taper': False, 'precision': 'fp32'}


def subsample(cfg, bl_stride=1, ch_count=None, src_stride=1):
    """Sub-sample a config (used for the bounded CPU baseline and for oracle-sized parity cases)."""
    out = dict(cfg)
    out['baselines'] = cfg['baselines'][::bl_stride]
    if ch_count is not None:
        out['channels'] = cfg['channels'][:ch_count]
    sky = cfg['sky']
    out['sky'] = {k: (v[::src_stride] if isinstance(v, NP.ndarray) else v) for k, v in sky.items()}
    return out
