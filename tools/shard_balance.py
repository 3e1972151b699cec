"""Load balance of baseline shards (one rank's work of an N-GPU run, measured on one GPU): the headline array's baselines are sorted by
length, so CONTIGUOUS shards (the reference's chunks, scripts/run_prisim.py:1775-1791) give the last rank all the long baselines -- the
groups that cannot use the lifting rotation (and, with the taper, the re-anchored bodies) -- and the job runs at the slowest rank's pace.
Shards dealt round-robin in groups of 256 baselines give every rank its share.  Prints the step time (wall clock per snapshot of six queued back to back) of
every rank's shard for both schemes.
    python tools/shard_balance.py N [taper | cfg5 | plain] [fp64]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as NP

import bench
from prisim_amd import _abi, workloads as W

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
taper = len(sys.argv) > 2 and sys.argv[2] in ('taper', 'cfg5')
cfg = W.config5(n_acc=1) if (len(sys.argv) > 2 and sys.argv[2] == 'cfg5') else W.config3(with_diffuse=taper)      # cfg5: one run of one pixel size
bl, ch, sky = cfg['baselines'], cfg['channels'], cfg['sky']
zen = NP.array([0.0, 0.0, 1.0])
PREC = _abi.PRISIM_FP64 if 'fp64' in sys.argv[2:] else _abi.PRISIM_FP32
ctx = _abi.Context(0)
out = {'n': N, 'workload': cfg['name'], 'taper': taper, 'precision': 'fp64' if PREC == _abi.PRISIM_FP64 else 'fp32'}
for scheme in ('contiguous', 'interleaved'):
    times = []
    for r in range(N):
        mine = bench.shard_baselines_contiguous(bl, N, r)[0] if scheme == 'contiguous' else bench.shard_baselines(bl, N, r)[0]
        ctx.set_array(mine, ch, nt_max=1)
        ctx.set_sky_analytic(sky['dircos'], sky['flux_ref'], sky['spindex'], sky['ref_freq'], _abi.PRISIM_BEAM_AIRY, 14.0, zen, zen,
                             fwhm_deg=(sky['fwhm_deg'] if taper else None))
        # wall clock per step of a queue of back-to-back snapshots (what a run does; since the partial-cube reduction moved to a side stream
        # the compute stream's own events no longer contain it)
        import time
        best = 1e9
        ctx.compute(precision=PREC)
        ctx.sync()
        for rep in range(3):
            t0 = time.perf_counter()
            for k in range(6):
                ctx.compute(precision=PREC)
            ctx.sync()
            best = min(best, (time.perf_counter() - t0) / 6 * 1e3)
        times.append(best)
    out[scheme] = {'ms_per_rank': times, 'slowest': max(times), 'mean': float(NP.mean(times)), 'slowest_over_mean': max(times) / float(NP.mean(times))}
print(json.dumps(out))
