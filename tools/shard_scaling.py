"""Development helper: per-GPU step time of the bench workload's baseline shards (what each rank of an N-GPU strong-scaling run
does, without the all-gather), on one GPU.  ideal = step time of the unsharded run / N."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as NP
from prisim_amd import _abi, workloads as W
from bench import shard_baselines

cfg = W.config3()
bl, ch, sky = cfg['baselines'], cfg['channels'], cfg['sky']
zen = NP.array([0.0, 0.0, 1.0])
K = 5
base = None
for world in (1, 2, 4, 8):
    mine, n_real = shard_baselines(bl, world, 0)
    ctx = _abi.Context(0)
    ctx.set_array(mine, ch, nt_max=K)
    ctx.set_sky_analytic(sky['dircos'], sky['flux_ref'], sky['spindex'], sky['ref_freq'], _abi.PRISIM_BEAM_AIRY, 14.0, zen, zen)
    ctx.compute(precision=_abi.PRISIM_FP32, slot=0); ctx.sync(); ctx.timing(reset=True)
    t0 = time.perf_counter()
    for t in range(K):
        ctx.compute(precision=_abi.PRISIM_FP32, slot=t)
    ctx.sync()
    dt = (time.perf_counter() - t0) / K * 1e3
    tm = ctx.timing()
    if base is None:
        base = dt
    print('world=%d shard=%d bl  step=%.3f ms  kernel=%.3f ms  ct=%d nsplit=%d  ideal=%.3f ms  efficiency=%.3f' %
          (world, mine.shape[0], dt, tm['sum_kernel_ms'] / tm['n_kernel'], tm['last_chan_tile'], tm['last_nsplit'], base / world, base / world / dt), flush=True)
    ctx.close()
