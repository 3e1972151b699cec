"""Diagnostic: config 4 snapshot 0 in fp32 through the catalogue path and through the per-snapshot upload path, each against the C oracle."""
import os, sys
import numpy as NP
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from prisim_amd import interferometry as RI, workloads as W, skymodel as SM, geometry as GEOM
from oracle import c_oracle as CO, healpix_oracle as H
from conftest import body_class_sample
cfg = W.config4(n_acc=1)
bl, ch, sky, lat = cfg['baselines'], cfg['channels'], cfg['sky'], cfg['latitude']
kap = float(NP.log(2.0) * (2.0 * NP.sin(0.5 * NP.radians(NP.max(sky['fwhm_deg'])))) ** 2)
sel, _ = body_class_sample(bl, ch, sky['dircos'], NP.array([0.0, 0.0, 1.0]), f32=True, kappa=kap)
lst0 = 40.0
hadec = GEOM.altaz2hadec(sky['altaz'], lat, units='degrees')
radec = NP.stack(((lst0 - hadec[:, 0]) % 360.0, hadec[:, 1]), axis=1)
n = radec.shape[0]
skymod = SM.SkyModel(location=radec, flux_ref=sky['flux_ref'], spindex=sky['spindex'], ref_freq=sky['ref_freq'],
                     src_shape=NP.stack((sky['fwhm_deg'], sky['fwhm_deg'], NP.zeros(n)), axis=1), epoch=None)
zen = NP.array([0.0, 0.0, 1.0])
dc, altaz, keep = W.drift_snapshot_directions(sky, lat, 0.0)
flux = sky['flux_ref'][keep, None] * (ch[None, :] / sky['ref_freq']) ** sky['spindex'][keep, None]
beam = H.external_beam(cfg['beam_table'], cfg['beam_freqs'], NP.pi / 2 - NP.radians(altaz[:, 0]), NP.radians(altaz[:, 1]), ch)
pb = beam.astype(NP.float32).astype(NP.float64) * flux
ref = CO.skyvis(bl[sel], ch, dc, pb, zen, fwhm_deg=sky['fwhm_deg'][keep])
scale = NP.sum(NP.abs(pb), axis=0)[None, :]
for mode in ('catalog', 'upload', 'catalog_f64'):
    os.environ['PRISIM_CATALOG'] = '0' if mode == 'upload' else '1'
    ia = RI.InterferometerArray(['b%d' % i for i in range(bl.shape[0])], bl, ch, telescope={'id': 'mwa'}, latitude=lat, skycoords='radec', pointing_coords='hadec')
    ia.set_external_beam(cfg['beam_table'], cfg['beam_freqs'], spec_interp='cubic')
    ia.observe((2457000.5, lst0), {'Tnet': 100.0}, NP.ones(ch.size), [0.0, lat], skymod, 112.0, memsave=(mode != 'catalog_f64'))
    v = ia.skyvis_freq[:, :, 0]
    e = NP.abs(v[sel] - ref) / scale
    idx = NP.asarray(ia.obs_catalog_indices[0])
    print(mode, 'err max %.3e' % e.max(), 'per baseline', NP.round(e.max(axis=1) * 1e6, 2), 'nroi', idx.size, NP.array_equal(idx, NP.flatnonzero(keep)),
          ia._ctx.timing()['last_culled_fraction'])
    pbd = ia._ctx.get_pbflux()
    print('   pbflux nsrc', pbd.shape, 'sum|pb| rel diff', float(abs(NP.abs(pbd).sum() - NP.abs(pb).sum()) / NP.abs(pb).sum()))
    if mode == 'upload':
        from prisim_amd import frames as FR
        rot, beta = FR.snapshot_frame('radec', lst0, lat, model='date')
        dcn = GEOM.frame_dircos(GEOM.catalog_unitvec(radec, 'radec'), rot, beta)[keep]
        order = NP.argsort(-dcn[:, 2], kind='stable')
        rel = NP.abs(pbd - pb[order]) / NP.max(NP.abs(pb))
        worst = NP.argsort(-rel.max(axis=1))[:8]
        print('   worst sources (sorted pos, alt, az, rel err):')
        for w in worst:
            print('     ', int(w), NP.round(altaz[order][w], 6), '%.3e' % rel[w].max(), 'dircos diff new-old %.2e' % NP.max(NP.abs(dcn[order][w] - dc[order][w])))
        print('   count rel > 1e-7:', int((rel.max(axis=1) > 1e-7).sum()))
