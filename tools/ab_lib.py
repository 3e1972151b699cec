"""Development helper: A/B two builds of libprisim_hip.so on the headline workload in alternating processes (same box).
usage: python tools/ab_lib.py <lib_a.so> <lib_b.so>"""
import subprocess, sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = '''
import sys, os
sys.path.insert(0, %r)
import numpy as NP
from prisim_amd import _abi
_abi.LIB_PATH = sys.argv[1]
from prisim_amd import workloads as W
cfg = W.config3(); bl, ch, sky = cfg['baselines'], cfg['channels'], cfg['sky']
ctx = _abi.Context(0); ctx.set_array(bl, ch, nt_max=1); zen = NP.array([0.0, 0.0, 1.0])
for taper in (False, True):
    ctx.set_sky_analytic(sky['dircos'], sky['flux_ref'], sky['spindex'], sky['ref_freq'], _abi.PRISIM_BEAM_AIRY, 14.0, zen, zen,
                         fwhm_deg=(NP.full(sky['dircos'].shape[0], 0.46) if taper else None))
    best = 1e9
    for rep in range(3):
        ctx.compute(precision=(_abi.PRISIM_FP64 if os.environ.get('AB_PREC') == 'fp64' else _abi.PRISIM_FP32)); ctx.sync(); best = min(best, ctx.timing()['last_kernel_ms'])
    print(os.path.basename(sys.argv[1]), 'taper=%%d kern_ms=%%.2f' %% (taper, best), flush=True)
''' % root
for rnd in range(2):
    for lib in sys.argv[1:]:
        subprocess.run([sys.executable, '-c', code, os.path.abspath(lib)], check=True)
