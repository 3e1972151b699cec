"""Small-problem regime: DEVICE time of the sky-sum kernel and of the whole compute() (prep + pack + sum + reduce,
hipEvent pairs on the stream -- not host wall, which a back-to-back observe loop hides) against channel tile and source split, for
  cfg2      BASELINE config 2: HERA-19 (171 bl) x 256 ch x nside-16 diffuse sky, fp64 and fp32, taper on        6.6e7 terms / snapshot
  cfg4s8    one rank's share of config 4 at N = 8: 1016 of the 8128 MWA-128T baselines x 768 ch x nside-64 sky    1.9e10 terms / snapshot
and the planner's own choice (ct = 0, nsplit = 0).  Prints one JSON line per case; the roofline fraction is terms x 10 flop / kernel time
against 78.6 (fp64) / 157.3 (fp32) TFLOP/s, the no-taper contract figure (the taper's extra work is not in it)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as NP

from prisim_amd import _abi, workloads as W

PEAK = {_abi.PRISIM_FP64: 78.6e12, _abi.PRISIM_FP32: 157.3e12}
zen = NP.array([0.0, 0.0, 1.0])


def run_case(name, bl, ch, sky, beam_kind, precs, cands, reps=7):
    ctx = _abi.Context(0)
    ctx.set_array(bl, ch)
    ctx.set_sky_analytic(sky['dircos'], sky['flux_ref'], sky['spindex'], sky['ref_freq'], beam_kind, 14.0, zen, zen, fwhm_deg=sky['fwhm_deg'])
    terms = float(bl.shape[0]) * ch.size * sky['dircos'].shape[0]
    for prec in precs:
        rows = []
        for ct, ns in cands:
            try:
                ctx.set_tuning(ct, 0, ns)
            except ValueError:
                continue
            km, cm = [], []
            for r in range(reps):
                ctx.compute(precision=prec)
                ctx.sync()
                t = ctx.timing()
                km.append(t['last_kernel_ms'])
                cm.append(t['last_compute_ms'])
            t = ctx.timing()
            k, c = float(NP.median(km[2:])), float(NP.median(cm[2:]))
            rows.append({'ct_asked': ct, 'nsplit_asked': ns, 'ct': t['last_chan_tile'], 'nsplit': t['last_nsplit'], 'kernel_us': k * 1e3, 'compute_us': c * 1e3,
                         'roofline_frac_kernel': terms * 10.0 / (k * 1e-3) / PEAK[prec], 'roofline_frac_compute': terms * 10.0 / (c * 1e-3) / PEAK[prec]})
        best = min(rows, key=lambda r: r['compute_us'])
        if cands[0] == (0, 0) and len(rows) > 1:
            # the planner's choice once more at the end: its first measurement also carries the clock ramp of a cold GPU
            ctx.set_tuning(0, 0, 0)
            km = []
            for r in range(reps):
                ctx.compute(precision=prec)
                ctx.sync()
                km.append(ctx.timing()['last_kernel_ms'])
            rows[0]['kernel_us_remeasured_last'] = float(NP.median(km[2:])) * 1e3
        print(json.dumps({'case': name, 'precision': 'fp64' if prec == _abi.PRISIM_FP64 else 'fp32', 'terms': terms, 'planner': rows[0], 'best': best,
                          'all': rows}), flush=True)
    ctx.close()


which = sys.argv[1:] or ['cfg2', 'cfg4s8']
if 'cfg2' in which:
    cfg = W.config2()
    cands = [(0, 0)] + [(ct, ns) for ct in (8, 16, 32, 64) for ns in (4, 8, 12, 16, 24, 32, 47)]
    run_case('cfg2', cfg['baselines'], cfg['channels'], cfg['sky'], _abi.PRISIM_BEAM_AIRY, (_abi.PRISIM_FP64, _abi.PRISIM_FP32), cands)
if 'cfg4s8' in which:
    cfg = W.config4(n_acc=1)
    bl = cfg['baselines'][:1016]
    cands = [(0, 0)] + [(ct, ns) for ct in (16, 32, 64) for ns in (1, 2, 4, 8, 10, 16, 24)]
    run_case('cfg4s8', bl, cfg['channels'], cfg['sky'], _abi.PRISIM_BEAM_AIRY, (_abi.PRISIM_FP32, _abi.PRISIM_FP64), cands, reps=5)
