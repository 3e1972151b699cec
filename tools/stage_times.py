"""Development helper: where one config-5 snapshot spends its wall time (host staging vs device)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as NP
from prisim_amd import _abi, workloads as W
t0 = time.perf_counter(); cfg = W.config5(n_acc=3); t1 = time.perf_counter()
print('build config (HEALPix nside 256, layout): %.2f s' % (t1 - t0))
bl, ch, sky = cfg['baselines'], cfg['channels'], cfg['sky']
ctx = _abi.Context(0); ctx.set_array(bl, ch, nt_max=3); zen = NP.array([0.0, 0.0, 1.0])
for j in range(3):
    a = time.perf_counter()
    dc, altaz, keep = W.drift_snapshot_directions(sky, cfg['latitude'], j * cfg['t_acc'] * 360.0 * 1.00273790935 / 86400.0)
    fr, sp, fw = sky['flux_ref'][keep], sky['spindex'][keep], sky['fwhm_deg'][keep]
    b = time.perf_counter()
    ctx.set_sky_analytic(dc, fr, sp, sky['ref_freq'], _abi.PRISIM_BEAM_AIRY, cfg['diameter'], zen, zen, fwhm_deg=fw)
    c = time.perf_counter()
    ctx.compute(precision=_abi.PRISIM_FP32, slot=j); ctx.sync()
    d = time.perf_counter()
    print('snapshot %d: nsrc %d  host directions %.3f s  set_sky_analytic (upload + beam kernel) %.3f s  compute %.3f s (kernel %.3f)' %
          (j, dc.shape[0], b - a, c - b, d - c, ctx.timing()['last_kernel_ms'] * 1e-3), flush=True)
