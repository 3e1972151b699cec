"""Development helper: step time (kernel + reduce) of 1/N baseline shards of the bench workload against the source-split factor;
candidates alternate so that clock / temperature drift of the box cancels."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as NP
from prisim_amd import _abi, workloads as W
from bench import shard_baselines
cfg = W.config3(); bl, ch, sky = cfg['baselines'], cfg['channels'], cfg['sky']
zen = NP.array([0.0, 0.0, 1.0])
cands = (1, 2, 3, 4, 6, 8, 12, 16)
for world in (2, 4, 8):
    mine, _ = shard_baselines(bl, world, 0)
    ctx = _abi.Context(0)
    ctx.set_array(mine, ch, nt_max=1)
    ctx.set_sky_analytic(sky['dircos'], sky['flux_ref'], sky['spindex'], sky['ref_freq'], _abi.PRISIM_BEAM_AIRY, 14.0, zen, zen)
    acc = {c: [] for c in cands}
    for rnd in range(5):
        for c in cands:
            ctx.set_tuning(0, 0, c)
            ctx.sync(); t0 = time.perf_counter(); ctx.compute(precision=_abi.PRISIM_FP32); ctx.sync()
            acc[c].append((time.perf_counter() - t0) * 1e3)
    print('world=%d shard=%d  median step ms by nsplit  ' % (world, mine.shape[0]) + '  '.join('%d:%.2f' % (c, NP.median(acc[c][1:])) for c in cands), flush=True)
    ctx.close()
