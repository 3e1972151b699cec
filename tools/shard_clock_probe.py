"""Workload of tools/shard_clock_probe.sh: 12 queued sky-sums of one rank's share of the headline workload at N = 8, then 4 of the whole array."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as NP
import bench
from prisim_amd import _abi, workloads as W
zen = NP.array([0.0, 0.0, 1.0])
cfg = W.config3(); sky = cfg['sky']
sh = bench.shard_baselines(cfg['baselines'], 8, 0)[0]
ctx = _abi.Context(0)
for bl, n in ((sh, 12), (cfg['baselines'], 4)):
    ctx.set_array(bl, cfg['channels'])
    ctx.set_sky_analytic(sky['dircos'], sky['flux_ref'], sky['spindex'], sky['ref_freq'], _abi.PRISIM_BEAM_AIRY, 14.0, zen, zen)
    for r in range(n):
        ctx.compute(precision=_abi.PRISIM_FP32)
    ctx.sync()
ctx.close()
