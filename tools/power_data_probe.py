"""Development helper: the headline kernel on the config-3 sky and on a sky whose 10 000 sources all sit at the phase centre (every phasor
is 1, every rotation the identity: the same instruction stream on operands that never change).  Under the package power limit the second
runs faster -- the gap is data-dependent switching power, not work."""
import os, sys, time, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as NP
from prisim_amd import _abi, workloads as W

cfg = W.config3(); bl, ch, sky = cfg['baselines'], cfg['channels'], cfg['sky']
zen = NP.array([0.0, 0.0, 1.0])
ctx = _abi.Context(0); ctx.set_array(bl, ch, nt_max=1)


def clock():
    try:
        out = subprocess.run(['rocm-smi', '--showclocks', '--showpower', '-d', '0'], capture_output=True, text=True, timeout=20).stdout
        return ' | '.join(l.split(':', 1)[1].strip() if ':' in l else l for l in out.splitlines() if 'sclk' in l or 'Package Power' in l)
    except Exception as exc:
        return repr(exc)


for name, dc in (('config-3 sky', sky['dircos']), ('all sources at the phase centre', NP.repeat(zen[None, :], sky['dircos'].shape[0], axis=0))):
    for prec, pname in ((_abi.PRISIM_FP32, 'fp32'), (_abi.PRISIM_FP64, 'fp64')):
        ctx.set_sky_analytic(dc, sky['flux_ref'], sky['spindex'], sky['ref_freq'], _abi.PRISIM_BEAM_AIRY, 14.0, zen, zen)
        for i in range(30):
            ctx.compute(precision=prec)
        ctx.sync(); ctx.timing(reset=True)
        n = 60 if prec == _abi.PRISIM_FP32 else 30
        for i in range(n):
            ctx.compute(precision=prec)
            if i == n // 2:
                smi = clock()
        ctx.sync()
        tm = ctx.timing()
        print('%-34s %s  kernel avg %.2f ms over %d launches   [%s]' % (name, pname, tm['sum_kernel_ms'] / tm['n_kernel'], tm['n_kernel'], smi), flush=True)
