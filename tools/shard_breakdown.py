"""Where a baseline shard's step goes (one rank's work of an N-GPU run of the headline workload, measured on one GPU): kernel time and
whole-compute time (prep + pack + kernel + partial reduce) by hipEvents against the source-split factor, candidates alternating so that
clock drift cancels; beside the ideal T1 / N of the unsharded step measured in the same process.

  python tools/shard_breakdown.py [taper] [fp64]      -> one JSON line per N in (2, 4, 8)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as NP

import bench
from prisim_amd import _abi, workloads as W

taper = 'taper' in sys.argv[1:]
prec = _abi.PRISIM_FP64 if 'fp64' in sys.argv[1:] else _abi.PRISIM_FP32
cfg = W.config3(with_diffuse=taper)
bl, ch, sky = cfg['baselines'], cfg['channels'], cfg['sky']
zen = NP.array([0.0, 0.0, 1.0])


def measure(ctx, cands, reps):
    acc = {c: ([], []) for c in cands}
    plan = {}
    for rnd in range(reps + 1):
        for c in cands:
            ctx.set_tuning(0, 0, c)
            ctx.compute(precision=prec)
            ctx.sync()
            t = ctx.timing()
            if rnd:
                acc[c][0].append(t['last_kernel_ms'])
                acc[c][1].append(t['last_compute_ms'])
            plan[c] = t['last_nsplit']
    return {str(c): {'nsplit': plan[c], 'kernel_ms': round(min(acc[c][0]), 4), 'compute_ms': round(min(acc[c][1]), 4),
                     'compute_ms_median': round(float(NP.median(acc[c][1])), 4)} for c in cands}


ctx = _abi.Context(0)


def set_sky():
    ctx.set_sky_analytic(sky['dircos'], sky['flux_ref'], sky['spindex'], sky['ref_freq'], _abi.PRISIM_BEAM_AIRY, 14.0, zen, zen,
                         fwhm_deg=(sky['fwhm_deg'] if taper else None))


ctx.set_array(bl, ch, nt_max=1)
set_sky()
full = measure(ctx, (0,), 3)['0']
print(json.dumps({'n': 1, 'workload': cfg['name'], 'taper': taper, 'precision': 'fp64' if prec == _abi.PRISIM_FP64 else 'fp32', 'planned': full}), flush=True)
for N in (2, 4, 8):
    out = {'n': N, 'ideal_compute_ms': round(full['compute_ms'] / N, 4), 'ranks': {}}
    for r in sorted(set((0, N - 1))):
        mine = bench.shard_baselines(bl, N, r)[0]
        ctx.set_array(mine, ch, nt_max=1)
        set_sky()
        cands = (0, 1, 2, 4, 6, 8, 12, 16) if r == 0 else (0,)
        res = measure(ctx, cands, 3)
        out['ranks'][str(r)] = {'nbl': int(mine.shape[0]), 'by_nsplit_request (0 = planner)': res,
                                'planned_over_ideal': round(res['0']['compute_ms'] / (full['compute_ms'] / N), 4)}
    print(json.dumps(out), flush=True)
