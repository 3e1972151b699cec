"""Cost of the baseline gradient (interferometry.py:6330-6343) on the headline array: plain pass vs fused MFMA pass vs four passes."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as NP
from prisim_amd import _abi, workloads as W
cfg = W.config3()
bl, ch, sky = cfg['baselines'], cfg['channels'], cfg['sky']
zen = NP.array([0.0, 0.0, 1.0])
ctx = _abi.Context(0)
ctx.set_array(bl, ch, nt_max=1)
out = {}
for taper in (False, True):
    ctx.set_sky_analytic(sky['dircos'], sky['flux_ref'], sky['spindex'], sky['ref_freq'], _abi.PRISIM_BEAM_AIRY, 14.0, zen, zen,
                         fwhm_deg=(NP.full(sky['dircos'].shape[0], 0.46) if taper else None))
    for name, prec, grad, env in (('plain_fp64', _abi.PRISIM_FP64, False, None), ('plain_fp32', _abi.PRISIM_FP32, False, None),
                                  ('fused_grad_fp64', _abi.PRISIM_FP64, True, None), ('fused_grad_fp32', _abi.PRISIM_FP32, True, None),
                                  ('four_pass_fp64', _abi.PRISIM_FP64, True, '0'),
                                  ('four_pass_fp32', _abi.PRISIM_FP32, True, '0')):
        if env is not None:
            os.environ['PRISIM_HIP_FUSED_GRAD'] = env
        else:
            os.environ.pop('PRISIM_HIP_FUSED_GRAD', None)
        best = 1e9
        for rep in range(2):
            ctx.compute(precision=prec, want_grad=grad)
            ctx.sync()
            best = min(best, ctx.timing()['last_compute_ms'])
        out['%s taper=%d' % (name, taper)] = best
print(json.dumps(out, indent=1))
