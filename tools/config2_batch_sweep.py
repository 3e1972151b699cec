"""Development helper: the batched small-array launch (k_skyvis_taper_f64_wave_batch) on config 2 x K snapshots over channel-tile widths
and source splits (set_tuning), and the library's own choice (tile 0 / split 0).   python tools/config2_batch_sweep.py [K]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as NP

from prisim_amd import _abi, geometry as GEOM, workloads as W

cfg = W.config2()
lat, lst0 = -30.7224, 30.0
sky = cfg['sky']
hadec = GEOM.altaz2hadec(sky['altaz'], lat, units='degrees')
radec = NP.stack(((lst0 - hadec[:, 0]) % 360.0, hadec[:, 1]), axis=1)
zen = NP.array([0., 0., 1.])
for k in ([int(sys.argv[1])] if len(sys.argv) > 1 else [16, 64, 256]):
    lsts = lst0 + 0.25 * NP.arange(k) * 64.0 / k
    with _abi.Context(0) as ctx:
        ctx.set_array(cfg['baselines'], cfg['channels'], nt_max=k)
        ctx.set_catalog(radec, 'radec', flux_ref=sky['flux_ref'], spindex=sky['spindex'], ref_freq_hz=sky['ref_freq'], fwhm_deg=sky['fwhm_deg'])
        obs = ctx.make_obs(lat, beam_kind=_abi.PRISIM_BEAM_AIRY, diameter_m=14.0)
        for ct, splits in ((0, (0,)), (32, (1, 2, 3, 4, 5, 6, 8, 12, 16, 24, 32)), (16, (2, 4, 8))):
            for ns in splits:
                ctx.set_tuning(ct, 0, ns)
                best, comp = 1e9, 1e9
                for rep in range(5):
                    ctx.sync()
                    ctx.timing(reset=True)
                    counts = ctx.observe_catalog(obs, lsts, zen, precision=_abi.PRISIM_FP64)
                    ctx.sync()
                    tm = ctx.timing()
                    best, comp = min(best, tm['sum_kernel_ms']), min(comp, tm['last_compute_ms'])
                terms = 171 * 256 * float(counts.sum())
                print('K %d tile %d split %d (ran %d): kernel %.3f ms = %.3f of the fp64 contract, compute %.3f ms = %.3f' % (
                    k, ct, ns, tm['last_nsplit'], best, terms * 10 / (best * 1e-3) / 78.6e12, comp, terms * 10 / (comp * 1e-3) / 78.6e12), flush=True)
