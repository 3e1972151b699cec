"""Development helper: worst per-term error of the fp32 kernels on a ONE-source sky (no averaging over sources), as a profile over the
position inside the 64-channel tile, on short and long HERA-350 baselines."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as NP
from prisim_amd import _abi, workloads as W
from oracle import c_oracle as CO, skyvis_oracle as O
cfg = W.config3(); bl = cfg['baselines'][::37][-600:]; ch = cfg['channels']
bl = NP.vstack((bl, cfg['baselines'][-200:]))
ctx = _abi.Context(0); ctx.set_array(bl, ch)
rng = NP.random.default_rng(3)
worst = {}
for trial in range(10):
    alt = rng.uniform(8, 80); az = rng.uniform(0, 360)
    dc = O.altaz2dircos(NP.array([[alt, az]]))
    pb = rng.uniform(0.5, 2.0, size=(1, ch.size))
    pc = NP.array([0.0, 0.0, 1.0])
    for taper in (0, 1):
        fw = NP.array([rng.uniform(0.05, 0.3)]) if taper else None
        ref = CO.skyvis(bl, ch, dc, pb, pc, fwhm_deg=fw)
        ctx.set_sky(dc, pb, pc, fwhm_deg=fw)
        for grp in ((0, 1) if taper else (0,)):
            os.environ['PRISIM_HIP_TAPER_GROUP'] = str(grp)
            ctx.set_tuning(64, 0, 1)
            ctx.compute(precision=_abi.PRISIM_FP32)
            e = NP.abs(ctx.get_vis() - ref) / NP.abs(pb)
            prof = e.reshape(e.shape[0], -1, 64).max(axis=(0, 1))
            worst[(taper, grp)] = NP.maximum(worst.get((taper, grp), 0), prof)
# the split taper form (one source = one run): sources near the zenith take the uncorrected bodies (8- and 16-step groups), sources
# near the horizon the corrected ones; PRISIM_HIP_TAPER_SPLIT=0 is the unsplit kernel on the same skies
os.environ.pop('PRISIM_HIP_TAPER_GROUP', None)
for trial in range(10):
    for where, alt in (('zenith', rng.uniform(80, 89.5)), ('horizon', rng.uniform(8, 40))):
        dc = O.altaz2dircos(NP.array([[alt, rng.uniform(0, 360)]]))
        pb = rng.uniform(0.5, 2.0, size=(1, ch.size))
        pc = NP.array([0.0, 0.0, 1.0])
        fw = NP.array([rng.choice([0.229, 0.458])])
        ref = CO.skyvis(bl, ch, dc, pb, pc, fwhm_deg=fw)
        ctx.set_sky(dc, pb, pc, fwhm_deg=fw)
        for sp in (1, 0):
            os.environ['PRISIM_HIP_TAPER_SPLIT'] = str(sp)
            ctx.set_tuning(64, 0, 1)
            ctx.compute(precision=_abi.PRISIM_FP32)
            tm = ctx.timing()
            e = NP.abs(ctx.get_vis() - ref) / NP.abs(pb)
            prof = e.reshape(e.shape[0], -1, 64).max(axis=(0, 1))
            key = ('split=%d %s (runs %d, uncorrected groups %d of %d)' % (sp, where, tm['last_taper_split'], tm['last_split_uncorrected_groups'], (bl.shape[0] + 255) // 256),)
            worst[key] = NP.maximum(worst.get(key, 0), prof)
os.environ.pop('PRISIM_HIP_TAPER_SPLIT', None)
for k, v in sorted(worst.items(), key=lambda kv: str(kv[0])):
    if len(k) == 1:
        print('%s worst %.2e  profile over tile position (1e-6):' % (k[0], v.max()), ' '.join('%.1f' % (x * 1e6) for x in v[::4]))
        continue
    print('taper=%d grouped=%d worst %.2e  profile over tile position (1e-6):' % (k[0], k[1], v.max()), ' '.join('%.1f' % (x * 1e6) for x in v[::4]))
