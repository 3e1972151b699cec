"""Where a snapshot of the PRODUCT loop goes: one rank's share of a BASELINE configuration through
InterferometerArray.observe() / observe_batch() on a (RA, Dec) sky model, as prisim_amd.driver.run drives it -- wall per snapshot with the
queue kept full (one synchronisation at the end) -- beside the kernel-only figure (the same shard's compute() with the sky already
resident, queued back to back), with the device-resident catalogue on and off (PRISIM_CATALOG=0: the sky of every snapshot formed on the
host and uploaded, the path of rounds 1-4).

    python tools/product_loop.py [--config 4|2|3] [--nranks 8] [--n-acc 32] [--precision fp32|fp64]
prints one JSON line per (mode, catalogue on / off)."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as NP

from prisim_amd import _abi, geometry as GEOM, interferometry as RI, sharding as SH, skymodel as SM, workloads as W

SIDEREAL_DEG_PER_SEC = 360.0 * 1.00273790935 / 86400.0


def build(cfgno, nranks, n_acc):
    if cfgno == 4:
        cfg = W.config4(n_acc=n_acc)
        tel = {'id': 'custom', 'shape': 'delta', 'size': 1.0, 'ocoords': 'altaz', 'orientation': NP.array([[90.0, 270.0]]), 'groundplane': None}
    elif cfgno == 2:
        cfg = W.config2()
        cfg.update(latitude=-30.7224, t_acc=60.0)
        tel = {'id': 'hera', 'shape': 'dish', 'size': 14.0, 'ocoords': 'altaz', 'orientation': NP.array([[90.0, 270.0]]), 'groundplane': None}
    else:
        cfg = W.config3(with_diffuse=False)
        cfg.update(latitude=-30.7224, t_acc=10.7)
        tel = {'id': 'hera', 'shape': 'dish', 'size': 14.0, 'ocoords': 'altaz', 'orientation': NP.array([[90.0, 270.0]]), 'groundplane': None}
    lat = cfg['latitude']
    sky = cfg['sky']
    hadec = GEOM.altaz2hadec(sky['altaz'], lat, units='degrees')
    lst0 = 30.0
    radec = NP.stack(((lst0 - hadec[:, 0]) % 360.0, hadec[:, 1]), axis=1)
    n = radec.shape[0]
    skymod = SM.SkyModel(location=radec, flux_ref=sky['flux_ref'], spindex=sky['spindex'], ref_freq=sky['ref_freq'],
                         src_shape=(NP.stack((sky['fwhm_deg'], sky['fwhm_deg'], NP.zeros(n)), axis=1) if cfg['taper'] else None))
    bl, idx, nreal = SH.shard_rows(cfg['baselines'], nranks, 0)
    return cfg, tel, skymod, bl, lat, lst0


def run_case(cfgno, nranks, n_acc, memsave, mode, catalog, reps=2):
    os.environ['PRISIM_CATALOG'] = '1' if catalog else '0'
    cfg, tel, skymod, bl, lat, lst0 = build(cfgno, nranks, n_acc)
    ch = cfg['channels']
    out = None
    for rep in range(reps):            # the second repetition is the one reported (allocations, first-touch, clock ramp are in the first)
        ia = RI.InterferometerArray(['b%d' % i for i in range(bl.shape[0])], bl, ch, telescope=tel, latitude=lat, skycoords='radec',
                                    pointing_coords='hadec')
        ia.reserve(n_acc)
        if cfg['beam'] == 'external':
            ia.set_external_beam(cfg['beam_table'], cfg['beam_freqs'])
        lsts = lst0 + NP.arange(n_acc) * cfg['t_acc'] * SIDEREAL_DEG_PER_SEC
        times = [(2455000.0 + j * cfg['t_acc'] / 86400.0, float(lsts[j])) for j in range(n_acc)]
        tsys = {'Tnet': 100.0}
        bp = NP.ones(ch.size)
        pc = NP.array([0.0, lat])
        ia._ctx.sync()
        ia._ctx.timing(reset=True)
        t0 = time.perf_counter()
        if mode == 'observe':
            for j in range(n_acc):
                ia.observe(times[j], tsys, bp, pc, skymod, cfg['t_acc'], memsave=memsave)
        else:
            ia.observe_batch(times, tsys, bp, pc, skymod, cfg['t_acc'], memsave=memsave)
        t_host = time.perf_counter() - t0          # the host is done queueing
        ia._ctx.sync()
        wall = time.perf_counter() - t0
        tm = ia._ctx.timing()
        nsrc = [int(e.size) for e in ia.obs_catalog_indices]
        # kernel-only: the last snapshot's sky is resident; its compute() queued n_acc times
        prec = _abi.PRISIM_FP32 if memsave else _abi.PRISIM_FP64
        batched = tm.get('last_batch_snapshots', 1)
        if catalog:      # (a batched launch leaves no single current sky: make the last snapshot's sky current again)
            from prisim_amd import geometry as G2
            pcd = G2.altaz2dircos(G2.hadec2altaz(pc, lat, units='degrees'), 'degrees').ravel()
            ia._ctx.set_sky_from_catalog(ia._catalog_obs_cache[1], float(lsts[-1]), pcd)
        ia._ctx.sync()
        ia._ctx.timing(reset=True)
        t1 = time.perf_counter()
        for j in range(n_acc):
            ia._ctx.compute(precision=prec, slot=j)
        ia._ctx.sync()
        wall_k = time.perf_counter() - t1
        tmk = ia._ctx.timing()
        out = {'config': cfgno, 'nranks': nranks, 'shard_baselines': int(bl.shape[0]), 'nchan': int(ch.size), 'n_acc': n_acc,
               'precision': 'fp32' if memsave else 'fp64', 'mode': mode, 'catalog': bool(catalog),
               'nsrc_roi_first_last': [nsrc[0], nsrc[-1]], 'wall_ms_per_snapshot': 1e3 * wall / n_acc,
               'host_ms_per_snapshot': 1e3 * t_host / n_acc, 'kernel_ms_per_snapshot': tm['sum_kernel_ms'] / max(tm['n_kernel'], 1),
               'kernel_only_wall_ms_per_snapshot': 1e3 * wall_k / n_acc,
               'kernel_only_kernel_ms_per_snapshot': tmk['sum_kernel_ms'] / max(tmk['n_kernel'], 1),
               'ratio_wall_over_kernel_only_wall': (wall / n_acc) / (wall_k / n_acc), 'chan_tile': tm['last_chan_tile'], 'nsplit': tm['last_nsplit'],
               'culled_fraction_last': tm['last_culled_fraction'], 'snapshots_per_launch': batched}
        del ia
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', type=int, default=4)
    ap.add_argument('--nranks', type=int, default=8)
    ap.add_argument('--n-acc', type=int, default=32)
    ap.add_argument('--precision', default=None)
    ap.add_argument('--modes', default='observe,batch')
    args = ap.parse_args()
    prec = args.precision or ('fp64' if args.config == 2 else 'fp32')
    for mode in args.modes.split(','):
        for catalog in (True, False):
            if mode == 'batch' and not catalog:
                continue
            print(json.dumps(run_case(args.config, args.nranks, args.n_acc, prec == 'fp32', mode, catalog)), flush=True)


if __name__ == '__main__':
    main()
