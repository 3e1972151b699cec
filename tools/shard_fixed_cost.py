"""Fixed cost against per-source cost of one rank's share of the headline workload at N = 8 (7680 baselines x 1024 channels, fp32, 8 source
splits): the sky of 1e4 point sources repeated 1, 2, 4, 8 times at a FIXED split count -- kernel time = fixed + slope x sources -- and, for
comparison, the same for the unsharded array (61 075 baselines, no split).  hipEvents around the sky-sum kernel (ctx.timing()), min of 5.
    python tools/shard_fixed_cost.py [splits]      splits: the shard at 2, 4, 8, 16, 32 splits (fixed cost against blocks per launch)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as NP
import bench
from prisim_amd import _abi, workloads as W

zen = NP.array([0.0, 0.0, 1.0])
cfg = W.config3()
sky = cfg['sky']
rng = NP.random.default_rng(9)


def repeated(k):
    d = [sky['dircos']]
    for r in range(1, k):
        dc = sky['dircos'] + 1e-3 * rng.standard_normal(sky['dircos'].shape)
        dc[:, 2] = NP.abs(dc[:, 2]); dc /= NP.linalg.norm(dc, axis=1, keepdims=True)
        d.append(dc)
    return NP.concatenate(d), NP.tile(sky['flux_ref'], k), NP.tile(sky['spindex'], k)


sh = bench.shard_baselines(cfg['baselines'], 8, 0)[0]
cases = [('shard N=8', sh, 8), ('unsharded', cfg['baselines'], 1)]
if 'splits' in sys.argv[1:]:
    cases = [('shard N=8, %d splits' % n, sh, n) for n in (2, 4, 8, 16, 32)]
for name, bl, ns in cases:
    ctx = _abi.Context(0)
    ctx.set_array(bl, cfg['channels'])
    rows = []
    for k in (1, 2, 4, 8):
        d, f, sp = repeated(k)
        ctx.set_sky_analytic(d, f, sp, sky['ref_freq'], _abi.PRISIM_BEAM_AIRY, 14.0, zen, zen)
        ctx.set_tuning(64, 0, ns)
        km = []
        for r in range(6):
            ctx.compute(precision=_abi.PRISIM_FP32); ctx.sync(); km.append(ctx.timing()['last_kernel_ms'])
        rows.append({'nsrc': int(d.shape[0]), 'kernel_ms': min(km[1:]), 'nsplit': ctx.timing()['last_nsplit']})
    x = NP.array([r['nsrc'] for r in rows], dtype=float); y = NP.array([r['kernel_ms'] for r in rows])
    slope, icpt = NP.polyfit(x, y, 1)
    print(json.dumps({'case': name, 'nbl': int(bl.shape[0]), 'rows': rows, 'fit_ms_per_1e4_sources': slope * 1e4, 'fit_fixed_ms': icpt}), flush=True)
    ctx.close()
