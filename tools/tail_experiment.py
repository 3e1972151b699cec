"""Development helper: does the partly filled last round of blocks cost what the arithmetic says?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as NP
from prisim_amd import _abi, workloads as W
cfg = W.config3()
ch, sky = cfg['channels'], cfg['sky']
zen = NP.array([0.0, 0.0, 1.0])
ctx = _abi.Context(0)
for nbl in (61075, 57344, 65536, 61184, 32768, 36864):
    bl = NP.resize(cfg['baselines'], (nbl, 3))
    ctx.set_array(bl, ch, nt_max=1)
    ctx.set_sky_analytic(sky['dircos'], sky['flux_ref'], sky['spindex'], sky['ref_freq'], _abi.PRISIM_BEAM_AIRY, 14.0, zen, zen)
    best = 1e9
    for rep in range(3):
        ctx.compute(precision=_abi.PRISIM_FP32); ctx.sync()
        t = ctx.timing(); best = min(best, t['last_kernel_ms'])
    blocks = ((nbl + 255) // 256) * 16
    print('nbl=%d blocks=%d rounds=%.3f kern_ms=%.2f terms/s=%.3e' % (nbl, blocks, blocks / 512.0, best, t['last_terms'] / best * 1e3), flush=True)
