import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as NP
from prisim_amd import _abi, workloads as W
from oracle import c_oracle as CO, beams_oracle as BO
cfg = W.config3()
bl, ch, sky = cfg['baselines'], cfg['channels'], cfg['sky']
zen = NP.array([0.0, 0.0, 1.0])
ctx = _abi.Context(0)
KEYS = ('PRISIM_HIP_BALANCED', 'PRISIM_HIP_BALANCED_BLOCKS', 'PRISIM_HIP_BALANCED_STAGGER', 'PRISIM_HIP_BALANCED_COST_NOLIFT', 'PRISIM_HIP_BALANCED_OLD_SHARE')
def run(tag, **env):
    for k in KEYS:
        os.environ.pop(k, None)
    os.environ.update({k: str(v) for k, v in env.items()})
    ts = []
    for rep in range(3):
        ctx.compute(precision=_abi.PRISIM_FP32); ctx.sync(); ts.append(round(ctx.timing()['last_kernel_ms'], 2))
    print(tag, ts, flush=True)
ctx.set_array(bl, ch, nt_max=1)
ctx.set_sky_analytic(sky['dircos'], sky['flux_ref'], sky['spindex'], sky['ref_freq'], _abi.PRISIM_BEAM_AIRY, 14.0, zen, zen)
run('legacy', PRISIM_HIP_BALANCED=0)
for share in (512, 640, 700, 740, 768, 800, 840, 880):
    run('balanced 512 cost 82 old_share %d' % share, PRISIM_HIP_BALANCED_COST_NOLIFT=82, PRISIM_HIP_BALANCED_OLD_SHARE=share)
run('legacy', PRISIM_HIP_BALANCED=0)
# parity at the best-looking share
pb = ctx.get_pbflux()
sel = NP.unique(NP.linspace(0, bl.shape[0] - 1, 12).astype(int))
ref = CO.skyvis(bl[sel], ch, sky['dircos'], pb, zen)
for share in (512, 768):
    os.environ['PRISIM_HIP_BALANCED_OLD_SHARE'] = str(share)
    ctx.compute(precision=_abi.PRISIM_FP32); v = ctx.get_vis()[sel]
    print('share', share, 'parity', float(NP.max(NP.abs(v - ref) / NP.sum(NP.abs(pb), axis=0)[None, :])))
