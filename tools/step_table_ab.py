"""Round-4 seed experiment on the headline kernel: the lifting groups' step phasor (sin alpha, tan alpha/2) of every
(source, baseline) pair from a table written by a pre-pass (k_step_table, 4.9 GB at config 3, built inside the timed region) instead of
two polynomials per (source, baseline, channel tile).  Alternating launches in one process; results must be bit-identical.
  python tools/step_table_ab.py [rounds]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as NP
from prisim_amd import _abi, workloads as W

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
cfg = W.config3()
bl, ch, sky = cfg['baselines'], cfg['channels'], cfg['sky']
zen = NP.array([0.0, 0.0, 1.0])
ctx = _abi.Context(0)
ctx.set_array(bl, ch, nt_max=1)
ctx.set_sky_analytic(sky['dircos'], sky['flux_ref'], sky['spindex'], sky['ref_freq'], _abi.PRISIM_BEAM_AIRY, 14.0, zen, zen)
times = {'0': [], '1': []}
vis = {}
for rnd in range(rounds + 1):
    for mode in ('0', '1'):
        os.environ['PRISIM_HIP_STEP_TABLE'] = mode
        ctx.compute(precision=_abi.PRISIM_FP32)
        ctx.sync()
        if rnd:
            times[mode].append(ctx.timing()['last_kernel_ms'])
        elif mode not in vis:
            vis[mode] = ctx.get_vis()[::97]
os.environ.pop('PRISIM_HIP_STEP_TABLE', None)
print(json.dumps({'workload': cfg['name'], 'kernel_ms_polynomials (k_skyvis_rec_f32pk<64,false>)': [round(t, 3) for t in times['0']],
                  'kernel_ms_step_table (k_step_table + k_skyvis_rec_f32pk_stab<64>)': [round(t, 3) for t in times['1']],
                  'min_polynomials': min(times['0']), 'min_step_table': min(times['1']),
                  'median_polynomials': float(NP.median(times['0'])), 'median_step_table': float(NP.median(times['1'])),
                  'table_GB': 8.0 * 10240 * 61184 / 1e9, 'bit_identical': bool(NP.array_equal(vis['0'], vis['1']))}))
