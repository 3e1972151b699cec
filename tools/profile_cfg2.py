"""Profiling target for the small-problem regime (tools/profile_round.sh with PROFILE_CMD): BASELINE config 2 -- HERA-19 (171 bl) x 256 ch
x nside-16 diffuse sky, Airy beam, taper on, fp64 (or fp32 with argv[1] = fp32) -- computed 200 times back to back; prints one JSON line with
the hipEvent averages so that tools/summarize_pmc.py can put the counters beside them."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as NP

from prisim_amd import _abi, workloads as W

prec = _abi.PRISIM_FP32 if (len(sys.argv) > 1 and sys.argv[1] == 'fp32') else _abi.PRISIM_FP64
cfg = W.config2()
bl, ch, sky = cfg['baselines'], cfg['channels'], cfg['sky']
zen = NP.array([0.0, 0.0, 1.0])
ctx = _abi.Context(0)
ctx.set_array(bl, ch)
ctx.set_sky_analytic(sky['dircos'], sky['flux_ref'], sky['spindex'], sky['ref_freq'], _abi.PRISIM_BEAM_AIRY, 14.0, zen, zen, fwhm_deg=sky['fwhm_deg'])
for i in range(10):
    ctx.compute(precision=prec)
ctx.sync()
ctx.timing(reset=True)
n = 200
comp = []
for i in range(n):
    ctx.compute(precision=prec)
    if i % 16 == 15:                       # the timing ring holds 16 entries: harvest before it wraps
        ctx.sync()
        comp.append(ctx.timing()['last_compute_ms'])
ctx.sync()
tm = ctx.timing()
nsrc = sky['dircos'].shape[0]
terms = float(bl.shape[0]) * ch.size * nsrc
wp = 4 if prec == _abi.PRISIM_FP32 else 8
alg = nsrc * ch.size * wp + 32 * nsrc + 24 * bl.shape[0] + 8 * ch.size + 16 * bl.shape[0] * ch.size
kern_ms = tm['sum_kernel_ms'] / max(1, tm['n_kernel'])
peak = 157.3 if prec == _abi.PRISIM_FP32 else 78.6
print(json.dumps({'workload': cfg['name'], 'precision': 'fp32' if prec == _abi.PRISIM_FP32 else 'fp64', 'taper': True, 'launches': n,
                  'chan_tile': tm['last_chan_tile'], 'nsplit': tm['last_nsplit'], 'compute_ms_median (prep + pack + sum + reduce)': float(NP.median(comp)),
                  'roofline': {'terms_per_launch': terms, 'avg_kernel_ms': kern_ms, 'frac_no_taper_contract': terms * 10.0 / (kern_ms * 1e-3) / 1e12 / peak},
                  'roofline_hbm': {'algorithmic_bytes_per_launch': alg}}))
ctx.close()
