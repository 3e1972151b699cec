"""Development helper: step time of the full bench workload against the source-split factor."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as NP
from prisim_amd import _abi, workloads as W
cfg = W.config3(); bl, ch, sky = cfg['baselines'], cfg['channels'], cfg['sky']
zen = NP.array([0.0, 0.0, 1.0])
ctx = _abi.Context(0); ctx.set_array(bl, ch, nt_max=1)
for taper in (False, True):
    ctx.set_sky_analytic(sky['dircos'], sky['flux_ref'], sky['spindex'], sky['ref_freq'], _abi.PRISIM_BEAM_AIRY, 14.0, zen, zen,
                         fwhm_deg=(NP.full(sky['dircos'].shape[0], 0.46) if taper else None))
    row = []
    for ns in (1, 2, 3, 4, 6, 8):
        ctx.set_tuning(0, 0, ns)
        best = 1e9
        for rep in range(3):
            ctx.sync(); t0 = time.perf_counter(); ctx.compute(precision=_abi.PRISIM_FP32); ctx.sync(); best = min(best, (time.perf_counter() - t0) * 1e3)
        row.append('%d:%.2f' % (ns, best))
    print('taper=%d full cfg3 step ms by nsplit  ' % taper + '  '.join(row), flush=True)
