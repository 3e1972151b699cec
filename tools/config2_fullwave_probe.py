"""Probe for the wave-granular item idea (config 2, fp64, taper, ct 16): the same sky on 256 baselines (HERA-19's 171 + 85 copies scaled by
1.01) so that all four wavefronts of a block are full -- what a mapping that keeps every SIMD busy could reach -- against the 171 of the
real array, over source splits and chunk sizes.  `fp32`: the packed fp32 kernel on 32-channel tiles instead."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as NP
from prisim_amd import _abi, workloads as W

zen = NP.array([0.0, 0.0, 1.0])
PREC = _abi.PRISIM_FP32 if 'fp32' in sys.argv[1:] else _abi.PRISIM_FP64
CT = 32 if PREC == _abi.PRISIM_FP32 else 16
cfg = W.config2()
sky = cfg['sky']
bl171 = cfg['baselines']
bl256 = NP.concatenate([bl171, 1.01 * bl171[:85]])
for name, bl in (('171', bl171), ('256', bl256)):
    ctx = _abi.Context(0)
    ctx.set_array(bl, cfg['channels'])
    ctx.set_sky_analytic(sky['dircos'], sky['flux_ref'], sky['spindex'], sky['ref_freq'], _abi.PRISIM_BEAM_AIRY, 14.0, zen, zen, fwhm_deg=sky['fwhm_deg'])
    for chunk, ns in ((0, 0), (64, 24), (32, 24), (32, 32), (32, 47), (16, 32), (16, 40), (16, 47), (16, 64), (16, 94)):
        try:
            ctx.set_tuning(0 if (chunk, ns) == (0, 0) else CT, chunk, ns)
        except ValueError as e:
            print(json.dumps({'bl': name, 'chunk': chunk, 'ns': ns, 'error': str(e)})); continue
        km, cm = [], []
        for r in range(12):
            ctx.compute(precision=PREC); ctx.sync()
            t = ctx.timing(); km.append(t['last_kernel_ms'] * 1e3); cm.append(t['last_compute_ms'] * 1e3)
        t = ctx.timing()
        print(json.dumps({'precision': 'fp32' if PREC == _abi.PRISIM_FP32 else 'fp64', 'bl': name, 'chunk_asked': chunk, 'ns_asked': ns, 'ct': t['last_chan_tile'], 'nsplit': t['last_nsplit'],
                          'kernel_us_min': min(km[2:]), 'kernel_us_median': float(NP.median(km[2:])), 'compute_us_min': min(cm[2:])}), flush=True)
    ctx.close()
