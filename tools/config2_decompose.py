"""Where config 2's sky-sum kernel time goes: fixed cost per launch against cost per source.

HERA-19 (171 bl) x 256 ch, fp64, taper on, channel tile 16.  The nside-16 sky (1504 sources above the horizon) is repeated 1, 2, 4, 8 times
(directions jittered by 1e-3 so no two coincide) and timed
  (a) at a FIXED split count of 24: every wavefront's chain of sources grows, the grid does not -> slope = cost per wave-source when a
      wavefront has its SIMD to itself, intercept = launch + prologue + flush;
  (b) with the split count grown in step (24, 48, 96, 192): 64 sources per wavefront throughout, 1.1 -> 9 wavefronts per SIMD -> the same
      work per wavefront with neighbours to hide its latencies.
Device time by hipEvents (ctx.timing()); prints one JSON line per case."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as NP

from prisim_amd import _abi, workloads as W

zen = NP.array([0.0, 0.0, 1.0])
cfg = W.config2()
sky0 = cfg['sky']


def repeated(k):
    rng = NP.random.default_rng(5)
    d, f, fw, sp = [], [], [], []
    for r in range(k):
        dc = sky0['dircos'] + (1e-3 * rng.standard_normal(sky0['dircos'].shape) if r else 0.0)
        dc[:, 2] = NP.abs(dc[:, 2])
        dc /= NP.linalg.norm(dc, axis=1, keepdims=True)
        d.append(dc)
        f.append(sky0['flux_ref'])
        fw.append(NP.broadcast_to(sky0['fwhm_deg'], (dc.shape[0],)))
        sp.append(NP.broadcast_to(sky0['spindex'], (dc.shape[0],)))
    return NP.concatenate(d), NP.concatenate(f), NP.concatenate(fw), NP.concatenate(sp)


ctx = _abi.Context(0)
ctx.set_array(cfg['baselines'], cfg['channels'])
for k in (1, 2, 4, 8):
    d, f, fw, sp = repeated(k)
    ctx.set_sky_analytic(d, f, sp, sky0['ref_freq'], _abi.PRISIM_BEAM_AIRY, 14.0, zen, zen, fwhm_deg=fw)
    for what, ns in (('fixed_grid', 24), ('fixed_chain', 24 * k)):
        if k == 1 and what == 'fixed_chain':
            continue
        ctx.set_tuning(16, 0, ns)
        km, cm = [], []
        for r in range(12):
            ctx.compute(precision=_abi.PRISIM_FP64)
            ctx.sync()
            t = ctx.timing()
            km.append(t['last_kernel_ms'] * 1e3)
            cm.append(t['last_compute_ms'] * 1e3)
        t = ctx.timing()
        print(json.dumps({'case': what, 'sky_repeats': k, 'nsrc': int(d.shape[0]), 'ct': t['last_chan_tile'], 'nsplit': t['last_nsplit'],
                          'sources_per_wave': d.shape[0] / t['last_nsplit'], 'kernel_us_min': min(km[2:]), 'kernel_us_median': float(NP.median(km[2:])),
                          'compute_us_min': min(cm[2:])}), flush=True)
ctx.close()
