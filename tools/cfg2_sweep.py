"""Development helper: kernel+reduce time of BASELINE config 2 (171 bl x 256 ch x 1504 src, fp64, taper) against tile width and split."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as NP
from prisim_amd import _abi, workloads as W
cfg = W.config2(); bl, ch, sky = cfg['baselines'], cfg['channels'], cfg['sky']
zen = NP.array([0.0, 0.0, 1.0])
ctx = _abi.Context(0); ctx.set_array(bl, ch)
ctx.set_sky_analytic(sky['dircos'], sky['flux_ref'], sky['spindex'], sky['ref_freq'], _abi.PRISIM_BEAM_AIRY, 14.0, zen, zen, fwhm_deg=sky['fwhm_deg'])
cands = [(0, 0)] + [(ct, ns) for ct in (8, 16, 32) for ns in (8, 16, 24, 47, 94)]
acc = {c: [] for c in cands}
for rnd in range(8):
    for c in cands:
        ctx.set_tuning(c[0], 0, c[1])
        ctx.sync(); t0 = time.perf_counter(); ctx.compute(precision=_abi.PRISIM_FP64); ctx.sync()
        acc[c].append((time.perf_counter() - t0) * 1e6)
for c in cands:
    print('ct=%d nsplit=%d  median %.1f us' % (c[0], c[1], NP.median(acc[c][2:])), flush=True)
