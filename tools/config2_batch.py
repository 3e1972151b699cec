"""BASELINE config 2 (HERA-19 x 256 ch x nside-16 diffuse, fp64, taper) as a RUN: K LSTs through prisim_hip_observe_catalog -- the
batched launch (one sky-sum launch + one reduction for all K) against one launch per snapshot (PRISIM_HIP_WAVE_BATCH=0) -- whole
device time of the call by wall clock with the queue drained at both ends, kernel time by hipEvents, roofline against the fp64
10-flop contract.  With `grad`: visibilities + baseline gradients (the batched MFMA kernel, 16-flop contract; batched mode only).
usage: python tools/config2_batch.py [K ...] [grad]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as NP

from prisim_amd import _abi, geometry as GEOM, workloads as W

PEAK_F64 = 78.6e12


def main():
    grad = 'grad' in sys.argv[1:]
    flop = 16.0 if grad else 10.0
    ks = [int(x) for x in sys.argv[1:] if x != 'grad'] or [64]
    cfg = W.config2()
    lat, lst0 = -30.7224, 30.0
    sky = cfg['sky']
    hadec = GEOM.altaz2hadec(sky['altaz'], lat, units='degrees')
    radec = NP.stack(((lst0 - hadec[:, 0]) % 360.0, hadec[:, 1]), axis=1)
    zen = NP.array([0.0, 0.0, 1.0])
    for k in ks:
        lsts = lst0 + 0.25 * NP.arange(k)
        for mode in (('batch',) if grad else ('batch', 'single')):
            os.environ['PRISIM_HIP_WAVE_BATCH'] = '1' if mode == 'batch' else '0'
            with _abi.Context(0) as ctx:
                ctx.set_array(cfg['baselines'], cfg['channels'], nt_max=k)
                ctx.set_catalog(radec, 'radec', flux_ref=sky['flux_ref'], spindex=sky['spindex'], ref_freq_hz=sky['ref_freq'], fwhm_deg=sky['fwhm_deg'])
                obs = ctx.make_obs(lat, beam_kind=_abi.PRISIM_BEAM_AIRY, diameter_m=14.0)
                best = None
                for rep in range(6):
                    ctx.sync()
                    ctx.timing(reset=True)
                    t0 = time.perf_counter()
                    counts = ctx.observe_catalog(obs, lsts, zen, precision=_abi.PRISIM_FP64, want_grad=grad)
                    ctx.sync()
                    wall = time.perf_counter() - t0
                    tm = ctx.timing()
                    terms = float(cfg['baselines'].shape[0]) * cfg['channels'].size * float(NP.sum(counts))
                    nsum = float(NP.sum(counts))
                    alg = nsum * cfg['channels'].size * 8 + 32 * nsum + 24 * cfg['baselines'].shape[0] + 8 * cfg['channels'].size + \
                        k * 16.0 * (4 if grad else 1) * cfg['baselines'].shape[0] * cfg['channels'].size
                    rec = {'K': k, 'mode': mode + ('_grad' if grad else ''), 'flop_per_term': flop, 'wall_ms': 1e3 * wall, 'wall_us_per_snapshot': 1e6 * wall / k, 'kernel_ms_total': tm['sum_kernel_ms'],
                           'launches': tm['n_kernel'], 'terms': terms, 'chan_tile': tm['last_chan_tile'], 'nsplit': tm['last_nsplit'],
                           'roofline_whole_call': terms * flop / wall / PEAK_F64, 'roofline_kernel_only': terms * flop / (tm['sum_kernel_ms'] * 1e-3) / PEAK_F64,
                           # (the keys tools/summarize_pmc.py reads: per launch of the dominant kernel)
                           'roofline': {'terms_per_launch': terms / max(tm['n_kernel'], 1), 'avg_kernel_ms': tm['sum_kernel_ms'] / max(tm['n_kernel'], 1)},
                           'roofline_hbm': {'algorithmic_bytes_per_launch': alg / max(tm['n_kernel'], 1)}}
                    if rep > 0 and (best is None or rec['wall_ms'] < best['wall_ms']):
                        best = rec
                print(json.dumps(best), flush=True)


if __name__ == '__main__':
    main()
