"""fp64 + source-shape taper: the grouped kernel (k_skyvis_taper_f64) against the exact second-order form (k_skyvis_rec<double,CT,true>,
PRISIM_HIP_TAPER_F64_GROUP=0) on one box, alternating, plus a parity check of both against the C oracle on a baseline sample.

  python tools/taper_f64_ab.py [--diffuse] [--cfg4] [--reps N]
Workloads: config 3's array with its 1e4 sources given FWHM 0.46 deg (the round-2/3 timing case), optionally config 3 + nside-128 diffuse
(two source runs) and one rank's share of config 4 (long baselines: taper culling)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as NP
from prisim_amd import _abi, workloads as W


def oracle_check(ctx, bl, ch, sky, pb, zen, vis, nsample=24):
    from oracle import c_oracle as CO, skyvis_oracle as O
    idx = NP.unique(NP.linspace(0, bl.shape[0] - 1, nsample).astype(int))
    ref = CO.skyvis(bl[idx], ch, sky['dircos'], pb, zen, fwhm_deg=sky['fwhm_deg'])
    return float(NP.max(NP.abs(vis[idx] - ref) / O.abs_flux_sum(pb)[None, :]))


def run_case(name, bl, ch, sky, reps, check=True):
    zen = NP.array([0.0, 0.0, 1.0])
    out = {'case': name, 'nbl': int(bl.shape[0]), 'nchan': int(ch.size), 'nsrc': int(sky['dircos'].shape[0])}
    with _abi.Context(0) as ctx:
        ctx.set_array(bl, ch, nt_max=1)
        ctx.set_sky_analytic(sky['dircos'], sky['flux_ref'], sky['spindex'], sky['ref_freq'], _abi.PRISIM_BEAM_AIRY, 14.0, zen, zen,
                             fwhm_deg=sky['fwhm_deg'])
        pb = ctx.get_pbflux() if check else None
        times = {'grouped': [], 'exact': []}
        vis = {}
        for rep in range(reps):
            for mode, env in (('grouped', '1'), ('exact', '0')):
                os.environ['PRISIM_HIP_TAPER_F64_GROUP'] = env
                ctx.compute(precision=_abi.PRISIM_FP64)
                ctx.sync()
                t = ctx.timing()
                times[mode].append(t['last_kernel_ms'])
                if rep == 0:
                    out[mode + '_plan'] = {k: t[k] for k in ('last_chan_tile', 'last_nsplit', 'last_culled_fraction') if k in t}
                    if check:
                        vis[mode] = ctx.get_vis()
        os.environ.pop('PRISIM_HIP_TAPER_F64_GROUP', None)
        terms = float(bl.shape[0]) * ch.size * sky['dircos'].shape[0]
        for mode in times:
            best = min(times[mode])
            out[mode + '_ms'] = [round(v, 3) for v in times[mode]]
            out[mode + '_roofline_10flop'] = round(terms * 10.0 / (best * 1e-3) / 78.6e12, 4)
            out[mode + '_roofline_12flop'] = round(terms * 12.0 / (best * 1e-3) / 78.6e12, 4)
        out['speedup'] = round(min(times['exact']) / min(times['grouped']), 4)
        if check:
            from oracle import skyvis_oracle as O
            out['grouped_vs_exact'] = float(NP.max(NP.abs(vis['grouped'] - vis['exact']) / O.abs_flux_sum(pb)[None, :]))
            t0 = time.time()
            out['grouped_vs_oracle'] = oracle_check(ctx, bl, ch, sky, pb, zen, vis['grouped'])
            out['exact_vs_oracle'] = oracle_check(ctx, bl, ch, sky, pb, zen, vis['exact'])
            out['oracle_s'] = round(time.time() - t0, 1)
    print(json.dumps(out), flush=True)


def main():
    reps = int(sys.argv[sys.argv.index('--reps') + 1]) if '--reps' in sys.argv else 3
    cfg = W.config3()
    sky = dict(cfg['sky'])
    sky['fwhm_deg'] = NP.full(sky['dircos'].shape[0], 0.46)
    run_case('cfg3 array x 1e4 sources, FWHM 0.46 deg', cfg['baselines'], cfg['channels'], sky, reps)
    if '--diffuse' in sys.argv:
        cfg = W.config3(with_diffuse=True)
        run_case('cfg3 + nside-128 diffuse', cfg['baselines'], cfg['channels'], cfg['sky'], max(1, reps - 1))
    if '--cfg4' in sys.argv:
        cfg = W.config4()
        sky = cfg['sky']
        order = NP.argsort(-sky['altaz'][:, 0], kind='stable')          # by decreasing altitude, as observe() lists a run (culling)
        sky = {k: (v[order] if isinstance(v, NP.ndarray) else v) for k, v in sky.items()}
        run_case('cfg4 array (MWA-128T, 8128 bl) x 768 ch x nside-64 diffuse, Airy stand-in beam', cfg['baselines'], cfg['channels'], sky, reps)


if __name__ == '__main__':
    main()
