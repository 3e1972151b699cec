"""config 2 through the class, per snapshot: observe() and observe_batch() (64 and 256 snapshots per call), second pass of a resident
instance; and the C-ABI batch call at K = 64 / 256 (kernel, compute, whole call).  One JSON line each."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as NP

import bench
from prisim_amd import _abi, geometry as GEOM, workloads as W

for mode, n in (('observe', 64), ('observe', 256), ('batch', 64), ('batch', 256)):
    r = bench.product_loop_case(2, 1, n, False, mode, True, device=0, reps=1, passes=2)
    print(json.dumps({'mode': mode, 'n': n, 'us_per_snapshot': 1e3 * r['wall_ms_per_snapshot_resident'], 'fresh': 1e3 * r['wall_ms_per_snapshot'],
                      'host_us': 1e3 * r['host_ms_per_snapshot'], 'per_launch': r['snapshots_per_launch']}), flush=True)
cfg = W.config2()
sky = cfg['sky']
zen = NP.array([0.0, 0.0, 1.0])
lat, lst0 = -30.7224, 30.0
hadec = GEOM.altaz2hadec(sky['altaz'], lat, units='degrees')
radec = NP.stack(((lst0 - hadec[:, 0]) % 360.0, hadec[:, 1]), axis=1)
for nb in (64, 256, 1024):
    lsts = lst0 + 0.05 * NP.arange(nb)
    with _abi.Context(0) as c2:
        c2.set_array(cfg['baselines'], cfg['channels'], nt_max=nb)
        c2.set_catalog(radec, 'radec', flux_ref=sky['flux_ref'], spindex=sky['spindex'], ref_freq_hz=sky['ref_freq'], fwhm_deg=sky['fwhm_deg'])
        obs = c2.make_obs(lat, beam_kind=_abi.PRISIM_BEAM_AIRY, diameter_m=cfg['diameter'])
        best = None
        for rep in range(4):
            c2.sync(); c2.timing(reset=True)
            t0 = time.perf_counter()
            counts = c2.observe_catalog(obs, lsts, zen, precision=_abi.PRISIM_FP64)
            c2.sync()
            wall = time.perf_counter() - t0
            tm = c2.timing()
            terms = 171.0 * 256 * float(NP.sum(counts))
            f = lambda ms: terms * 10.0 / (ms * 1e-3) / 1e12 / bench.PEAK_TFLOPS['f64']
            rec = {'K': nb, 'per_launch': tm['last_batch_snapshots'], 'nsplit': tm['last_nsplit'], 'launches': tm['n_kernel'],
                   'kernel_us': 1e3 * tm['sum_kernel_ms'] / nb, 'call_us': 1e6 * wall / nb, 'frac_kernel': f(tm['sum_kernel_ms']), 'frac_call': f(wall * 1e3)}
            if rep and (best is None or rec['call_us'] < best['call_us']):
                best = rec
        print(json.dumps(best), flush=True)
