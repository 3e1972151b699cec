"""Development helper: time kernel variants on the headline workload (cfg3) on the GPU."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as NP
from prisim_amd import _abi, workloads as W

cfg = W.config3()
bl, ch, sky = cfg['baselines'], cfg['channels'], cfg['sky']
print(cfg['name'], bl.shape, ch.shape, sky['dircos'].shape)
ctx = _abi.Context(0)
ctx.set_array(bl, ch, nt_max=1)
zen = NP.array([0.0, 0.0, 1.0])
variants = sys.argv[1:] or ['f32:32:64', 'f32:64:64', 'f32:16:64', 'f32:32:128', 'f64:16:64', 'f64:32:64', 'f64:8:64']
for taper in (False, True):
    ctx.set_sky_analytic(sky['dircos'], sky['flux_ref'], sky['spindex'], sky['ref_freq'], _abi.PRISIM_BEAM_AIRY, 14.0, zen, zen,
                         fwhm_deg=(NP.full(sky['dircos'].shape[0], 0.46) if taper else None))
    for v in variants:
        prec, ct, chunk = v.split(':')
        p = _abi.PRISIM_FP32 if prec == 'f32' else _abi.PRISIM_FP64
        ctx.set_tuning(int(ct), int(chunk), 0)
        best = 1e9
        for rep in range(2):
            ctx.compute(precision=p); ctx.sync()
            t = ctx.timing()
            best = min(best, t['last_kernel_ms'])
        terms = t['last_terms']
        print('taper=%d %s kern_ms=%.2f compute_ms=%.2f  %.3e terms/s  nsplit=%d' % (taper, v, best, t['last_compute_ms'], terms / best * 1e3, t['last_nsplit']), flush=True)
