"""What the fixed ~0.5 ms per launch of a baseline shard's sky-sum is made of (tools/shard_fixed_cost.py found it: independent of the source
count and of the split count).  Kernel time (hipEvents, average over 8 launches queued back to back -- a launch after an idle gap pays
~1.3 ms of clock ramp on top, first block below) for one rank's share of the headline workload at N = 8 against sources, channels and
baselines; fp32, 8 splits."""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as NP
import bench
from prisim_amd import _abi, workloads as W

zen = NP.array([0.0, 0.0, 1.0])
cfg = W.config3(); sky = cfg['sky']
sh = bench.shard_baselines(cfg['baselines'], 8, 0)[0]


def queued(ctx, n=8):
    for r in range(3):
        ctx.compute(precision=_abi.PRISIM_FP32)
    ctx.sync(); ctx.timing(reset=True)
    for r in range(n):
        ctx.compute(precision=_abi.PRISIM_FP32)
    ctx.sync()
    tm = ctx.timing()
    return tm['sum_kernel_ms'] / tm['n_kernel']


ctx = _abi.Context(0)
ctx.set_array(sh, cfg['channels'])
ctx.set_sky_analytic(sky['dircos'], sky['flux_ref'], sky['spindex'], sky['ref_freq'], _abi.PRISIM_BEAM_AIRY, 14.0, zen, zen)
ctx.set_tuning(64, 0, 8)
single = []
for r in range(5):
    ctx.compute(precision=_abi.PRISIM_FP32); ctx.sync(); single.append(round(ctx.timing()['last_kernel_ms'], 3)); time.sleep(0.02)
print(json.dumps({'what': 'launch after 20 ms of idle against queued launches', 'after_idle_ms': single, 'queued_ms': queued(ctx)}), flush=True)
for label, bl, ch in (('7680 bl x 1024 ch', sh, cfg['channels']), ('7680 bl x 512 ch', sh, cfg['channels'][:512]), ('3840 bl x 1024 ch', sh[::2], cfg['channels']),
                      ('61075 bl x 1024 ch, no split', cfg['baselines'], cfg['channels'])):
    ctx.set_array(bl, ch)
    rows = []
    for nsrc in (640, 1280, 2560, 5120, 10000):
        ctx.set_sky_analytic(sky['dircos'][:nsrc], sky['flux_ref'][:nsrc], sky['spindex'][:nsrc], sky['ref_freq'], _abi.PRISIM_BEAM_AIRY, 14.0, zen, zen)
        ctx.set_tuning(64, 0, 1 if bl.shape[0] > 10000 else 8)
        rows.append((nsrc, round(queued(ctx), 4)))
    x = NP.array([r[0] for r in rows], dtype=float); y = NP.array([r[1] for r in rows])
    slope, icpt = NP.polyfit(x, y, 1)
    print(json.dumps({'array': label, 'kernel_ms_by_nsrc': rows, 'fit_ms_per_1e4_sources': slope * 1e4, 'fit_fixed_ms': icpt}), flush=True)
ctx.close()
