"""Development helper: step time of the full bench workload against the source-split factor, alternating the candidates so that
clock / temperature drift of the box cancels."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as NP
from prisim_amd import _abi, workloads as W
cfg = W.config3(); bl, ch, sky = cfg['baselines'], cfg['channels'], cfg['sky']
zen = NP.array([0.0, 0.0, 1.0])
ctx = _abi.Context(0); ctx.set_array(bl, ch, nt_max=1)
ctx.set_sky_analytic(sky['dircos'], sky['flux_ref'], sky['spindex'], sky['ref_freq'], _abi.PRISIM_BEAM_AIRY, 14.0, zen, zen)
cands = [int(a) for a in sys.argv[1:]] or [1, 2, 4]
acc = {c: [] for c in cands}
for rnd in range(6):
    for c in cands:
        ctx.set_tuning(0, 0, c)
        ctx.sync(); t0 = time.perf_counter(); ctx.compute(precision=_abi.PRISIM_FP32); ctx.sync()
        acc[c].append((time.perf_counter() - t0) * 1e3)
for c in cands:
    v = NP.array(acc[c][1:])
    print('nsplit=%d  step ms: median %.2f  min %.2f  max %.2f' % (c, NP.median(v), v.min(), v.max()), flush=True)
