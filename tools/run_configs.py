"""Run the five BASELINE.json configurations through the C-ABI on one MI355X and print one JSON line
per configuration: sizes, terms, kernel time, terms/s, wall time including sky staging, and a parity spot check of a few
baselines against the C oracle.  Config 5 is run for `--lst5` of its 120 LSTs (the full run is 2.95e15 terms)."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as NP

from prisim_amd import _abi, workloads as W, primary_beams as PB
from oracle import c_oracle as CO


def run(cfg, ctx, n_acc, spot=6, delay=False, cpu=False):
    bl, ch, sky = cfg['baselines'], cfg['channels'], cfg['sky']
    prec = _abi.PRISIM_FP32 if cfg['precision'] == 'fp32' else _abi.PRISIM_FP64
    zen = NP.array([0.0, 0.0, 1.0])
    ctx.set_array(bl, ch, nt_max=n_acc)
    ctx.set_tuning(0, 0, 0)
    if cfg['beam'] == 'external':
        ctx.set_external_beam(cfg['beam_table'], PB.spectral_interp_matrix(cfg['beam_freqs'], ch, kind='cubic'))
    kinds = {'gaussian': _abi.PRISIM_BEAM_GAUSSIAN, 'airy': _abi.PRISIM_BEAM_AIRY}
    kern_ms = 0.0
    terms = 0
    ctx.timing(reset=True)
    t0 = time.perf_counter()
    worst = 0.0
    for j in range(n_acc):
        if n_acc > 1:
            dc, altaz, keep = W.drift_snapshot_directions(sky, cfg['latitude'], j * cfg['t_acc'] * 360.0 * 1.00273790935 / 86400.0)
            fr, sp, fw = sky['flux_ref'][keep], sky['spindex'][keep], sky['fwhm_deg'][keep]
        else:
            dc, fr, sp, fw = sky['dircos'], sky['flux_ref'], sky['spindex'], sky['fwhm_deg']
        fwhm = fw if cfg['taper'] else None
        if cfg['taper']:
            # what InterferometerArray.observe() does before it uploads a sky with source sizes: by decreasing altitude, so that the
            # library's taper culling can skip the zenith-most sources of long-baseline groups (their weight underflows)
            order = NP.argsort(-dc[:, 2], kind='stable')
            dc, fr, sp, fwhm = dc[order], fr[order], sp[order], fwhm[order]
        if cfg['beam'] == 'external':
            ctx.set_sky_external_analytic(dc, fr, sp, sky['ref_freq'], zen, fwhm_deg=fwhm)
        else:
            ctx.set_sky_analytic(dc, fr, sp, sky['ref_freq'], kinds[cfg['beam']], cfg['diameter'], zen, zen, fwhm_deg=fwhm)
        ctx.compute(precision=prec, slot=j)
        terms += bl.shape[0] * ch.size * dc.shape[0]
        if j == n_acc - 1 and spot:
            ctx.sync()
            pb = ctx.get_pbflux()
            sel = NP.linspace(0, bl.shape[0] - 1, spot).astype(int)
            ref = CO.skyvis(bl[sel], ch, dc, pb, zen, fwhm_deg=fwhm)
            vis = ctx.get_vis(slot=j)[sel]
            worst = float(NP.max(NP.abs(vis - ref) / NP.sum(NP.abs(pb), axis=0)[None, :]))
    ctx.sync()
    wall = time.perf_counter() - t0
    if spot:
        # same loop again without the parity spot check (oracle + two downloads): what a production run pays per snapshot
        t0 = time.perf_counter()
        for j in range(n_acc):
            if n_acc > 1:
                dc, altaz, keep = W.drift_snapshot_directions(sky, cfg['latitude'], j * cfg['t_acc'] * 360.0 * 1.00273790935 / 86400.0)
                fr, sp, fw = sky['flux_ref'][keep], sky['spindex'][keep], sky['fwhm_deg'][keep]
            else:
                dc, fr, sp, fw = sky['dircos'], sky['flux_ref'], sky['spindex'], sky['fwhm_deg']
            fwhm = fw if cfg['taper'] else None
            if cfg['taper']:
                order = NP.argsort(-dc[:, 2], kind='stable')
                dc, fr, sp, fwhm = dc[order], fr[order], sp[order], fwhm[order]
            if cfg['beam'] == 'external':
                ctx.set_sky_external_analytic(dc, fr, sp, sky['ref_freq'], zen, fwhm_deg=fwhm)
            else:
                ctx.set_sky_analytic(dc, fr, sp, sky['ref_freq'], kinds[cfg['beam']], cfg['diameter'], zen, zen, fwhm_deg=fwhm)
            ctx.compute(precision=prec, slot=j)
        ctx.sync()
        wall_nospot = time.perf_counter() - t0
        ctx.timing(reset=True)
        for j in range(n_acc):
            ctx.compute(precision=prec, slot=j)
        ctx.sync()
    tm = ctx.timing()
    kern_ms = tm['sum_kernel_ms']
    out = {'config': cfg['name'], 'nbl': int(bl.shape[0]), 'nchan': int(ch.size), 'nsrc_catalog': int(sky['dircos'].shape[0]),
           'n_acc': n_acc, 'precision': cfg['precision'], 'taper': bool(cfg['taper']), 'terms': float(terms),
           'kernel_ms_total': kern_ms, 'terms_per_s_kernel': terms / (kern_ms * 1e-3) if kern_ms > 0 else None,
           'wall_s_incl_sky_staging_and_spot_check': wall, 'wall_s_incl_host_geometry_and_sky_staging': wall_nospot if spot else wall, 'chan_tile': tm['last_chan_tile'], 'nsplit': tm['last_nsplit'],
           'parity_spot_max_err_rel_sumflux': worst, 'tolerance': 5e-6 if cfg['precision'] == 'fp32' else 1e-11,
           'taper_split_runs': tm.get('last_taper_split', 0), 'taper_culled_fraction_last_snapshot': tm.get('last_culled_fraction', 0.0)}
    if cpu and n_acc == 1 and terms <= 3e8:
        # the CPU beside it on the same box: the numpy restatement of the reference's statements, one process, and the
        # C/OpenMP port on every host core -- whole configuration, and the GPU result checked against both
        from oracle import skyvis_oracle as O
        pb = ctx.get_pbflux()
        fwhm = sky['fwhm_deg'] if cfg['taper'] else None
        gpu = ctx.get_vis(slot=0)
        scale = NP.sum(NP.abs(pb), axis=0)[None, :]
        t1 = time.perf_counter(); ref_np = O.skyvis(bl, ch, sky['dircos'], pb, zen, fwhm_deg=fwhm); t_np = time.perf_counter() - t1
        CO.use_native_build()
        threads = max(1, min(16, os.cpu_count() or 1, CO.max_threads()))                # the box's CPU share for one GPU is 16 cores
        CO.skyvis(bl[:threads], ch, sky['dircos'], pb, zen, fwhm_deg=fwhm, nthreads=threads)   # load + thread pool warm-up
        t1 = time.perf_counter(); ref_c = CO.skyvis(bl, ch, sky['dircos'], pb, zen, fwhm_deg=fwhm, nthreads=threads); t_c = time.perf_counter() - t1
        out['cpu'] = {'numpy_reference_formulation_terms_per_s': terms / t_np, 'numpy_seconds': t_np, 'c_openmp_terms_per_s': terms / t_c,
                      'c_openmp_seconds': t_c, 'c_openmp_threads': threads,
                      'gpu_vs_numpy_max_err_rel_sumflux': float(NP.max(NP.abs(gpu - ref_np) / scale)),
                      'gpu_vs_c_max_err_rel_sumflux': float(NP.max(NP.abs(gpu - ref_c) / scale)),
                      'gpu_kernel_over_numpy': t_np / (kern_ms * 1e-3), 'gpu_kernel_over_c_openmp': t_c / (kern_ms * 1e-3)}
    if delay:
        w = NP.blackman(ch.size)
        from prisim_amd import delay_spectrum as DSM
        pconst = DSM.power_constants(ch, {'id': 'hera'}, freq_wts=w)      # abs^2 -> K^2 (Mpc/h)^3 (delay_spectrum.py:3659-3663, 3992)
        out['delay_power_scale_K2_Mpc3_per_Jy2Hz2'] = pconst['factor']
        for rep in range(2):
            t1 = time.perf_counter()
            ctx.delay_transform_device(n_acc, bpwts=w, pad=1.0, want_lag=False, want_power=True, power_scale=pconst['factor'])
            ctx.sync()
            out['delay_power_spectrum_wall_s'] = time.perf_counter() - t1
        tmd = ctx.timing()
        out['delay_power_spectrum_device_ms'] = tmd['last_delay_ms']
        out['delay_fused_kernel'] = bool(tmd['last_delay_fused'])
        out['delay_ffts'] = int(n_acc * bl.shape[0])
        out['delay_fft_length_kept'] = int(ch.size)
        nbytes = n_acc * bl.shape[0] * ch.size * (16 + 8)             # read each visibility once, write each power sample once
        out['delay_algorithmic_GBps'] = nbytes / (tmd['last_delay_ms'] * 1e-3) / 1e9
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--lst5', type=int, default=2)
    ap.add_argument('--nside5', type=int, default=256)
    ap.add_argument('--configs', default='1,2,3,4,5')
    ap.add_argument('--cpu', action='store_true', help='configs 1 and 2: also time the numpy restatement and the C/OpenMP port on the host cores')
    args = ap.parse_args()
    ctx = _abi.Context(0)
    which = [int(x) for x in args.configs.split(',')]
    if 1 in which:
        print(json.dumps(run(W.config1(), ctx, 1, cpu=args.cpu)), flush=True)
    if 2 in which:
        print(json.dumps(run(W.config2(), ctx, 1, cpu=args.cpu)), flush=True)
    if 3 in which:
        print(json.dumps(run(W.config3(), ctx, 1)), flush=True)
    if 4 in which:
        print(json.dumps(run(W.config4(), ctx, 32)), flush=True)
    if 5 in which:
        print(json.dumps(run(W.config5(n_acc=args.lst5, nside=args.nside5), ctx, args.lst5, delay=True)), flush=True)
    ctx.close()


if __name__ == '__main__':
    main()
