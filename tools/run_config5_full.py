"""BASELINE config 5 in full on ONE MI355X through the reference's entry point for the path: HERA-350 (61 075 baselines) x 1024 channels x
nside=256 diffuse sky (392 704 pixels above the horizon, source-shape taper) x 120 LSTs of 10.7 s drift, fp32 ("memsave"), the 120 GB
visibility cube resident in HBM, then the delay power spectra of all 7.3e6 rows (pad = 1) on the device.  One JSON line on stdout;
progress on stderr.  ~7 minutes of kernels.

    python tools/run_config5_full.py [n_lst] [fp64]      fp64: the reference's default precision (memsave=False): complex128 snapshots, the grouped
                                                         fp64 taper kernel, tolerance 1e-11 in the spot check, ~14 minutes of kernels
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as NP

from prisim_amd import workloads as W, geometry as GEOM, interferometry as RI, skymodel as SM
from oracle import c_oracle as CO, beams_oracle as BO          # spot check of the last snapshot only

SIDEREAL_DEG_PER_SEC = 360.0 * 1.00273790935 / 86400.0
n_lst = int(sys.argv[1]) if len(sys.argv) > 1 else 120
fp64 = 'fp64' in sys.argv[2:]
cfg = W.config5(n_acc=n_lst)
bl, ch, sky, lat = cfg['baselines'], cfg['channels'], cfg['sky'], cfg['latitude']
lst0 = 15.0
hadec = GEOM.altaz2hadec(sky['altaz'], lat, units='degrees')
radec = NP.stack(((lst0 - hadec[:, 0]) % 360.0, hadec[:, 1]), axis=1)
n = radec.shape[0]
skymod = SM.SkyModel(location=radec, flux_ref=sky['flux_ref'], spindex=sky['spindex'], ref_freq=sky['ref_freq'],
                     src_shape=NP.stack((sky['fwhm_deg'], sky['fwhm_deg'], NP.zeros(n)), axis=1), epoch=None)
ia = RI.InterferometerArray(['b%d' % i for i in range(bl.shape[0])], bl, ch, telescope={'id': 'hera', 'orientation': [90.0, 270.0], 'ocoords': 'altaz'},
                            latitude=lat, skycoords='radec', pointing_coords='hadec')
ia.reserve(n_lst)
ia._ctx.timing(reset=True)
terms = 0
t0 = time.perf_counter()
for j in range(n_lst):
    ia.observe((2457000.5 + j * cfg['t_acc'] / 86400.0, lst0 + j * cfg['t_acc'] * SIDEREAL_DEG_PER_SEC), {'Tnet': 100.0}, NP.ones(ch.size),
               [0.0, lat], skymod, cfg['t_acc'], memsave=not fp64)
    terms += bl.shape[0] * ch.size * ia.obs_catalog_indices[j].size
    if j % 5 == 4 or j == n_lst - 1:
        ia._ctx.sync()
        sys.stderr.write('LST %3d of %d  wall %.1f s\n' % (j + 1, n_lst, time.perf_counter() - t0))
        sys.stderr.flush()
ia._ctx.sync()
wall = time.perf_counter() - t0
tm = ia._ctx.timing()
assert all(isinstance(s, RI._DeviceSlot) for s in ia._cube)                       # nothing left the device while observing

# parity spot check, last snapshot, 3 baselines
j = n_lst - 1
dc, altaz, keep = W.drift_snapshot_directions(sky, lat, j * cfg['t_acc'] * SIDEREAL_DEG_PER_SEC)
pb = BO.airy_disk_pattern(14.0, altaz, ch, pointing_altaz=[90.0, 270.0]) * (sky['flux_ref'][keep, None] * (ch[None, :] / sky['ref_freq']) ** sky['spindex'][keep, None])
sel = NP.unique(NP.linspace(0, bl.shape[0] - 1, 3).astype(int))
ref = CO.skyvis(bl[sel], ch, dc, pb, NP.array([0.0, 0.0, 1.0]), fwhm_deg=sky['fwhm_deg'][keep])
vis = ia._ctx.get_vis(slot=j)[sel]
err = float(NP.max(NP.abs(vis - ref) / NP.sum(NP.abs(pb), axis=0)[None, :]))

# delay power spectra of every row, on the device
w = NP.blackman(ch.size) + 0.01
from prisim_amd import delay_spectrum as DSM
pconst = DSM.power_constants(ch, {'id': 'hera'}, freq_wts=w)      # abs^2 -> K^2 (Mpc/h)^3 (delay_spectrum.py:3659-3663, 3992)
t1 = time.perf_counter()
ia._ctx.delay_transform_device(n_lst, bpwts=w, pad=1.0, want_lag=False, want_power=True, power_scale=pconst['factor'])
ia._ctx.sync()
dwall = time.perf_counter() - t1
tmd = ia._ctx.timing()
pw = ia._ctx.get_delay_power(n_lst - 1, 1, rows=sel)                               # (1, 3, nlag)
rows = n_lst * bl.shape[0]
print(json.dumps({
    'config': cfg['name'], 'n_lst': n_lst, 'nbl': int(bl.shape[0]), 'nchan': int(ch.size), 'nsrc_catalog': int(n), 'terms': float(terms),
    'precision': 'fp64' if fp64 else 'fp32', 'taper': True, 'wall_s_observe_loop': wall, 'kernel_s_total': tm['sum_kernel_ms'] * 1e-3, 'n_kernel': tm['n_kernel'],
    'terms_per_s_wall': terms / wall, 'terms_per_s_kernel': terms / (tm['sum_kernel_ms'] * 1e-3), 'cube_GB_resident': rows * ch.size * 16 / 1e9,
    'parity_spot_max_err_rel_sumflux_last_lst': err, 'tolerance': 1e-11 if fp64 else 5e-6,
    'delay_ffts': rows, 'delay_device_ms': tmd['last_delay_ms'], 'delay_wall_s': dwall, 'delay_fused_kernel': bool(tmd['last_delay_fused']),
    'delay_algorithmic_GBps': rows * ch.size * 24 / (tmd['last_delay_ms'] * 1e-3) / 1e9, 'delay_power_finite': bool(NP.all(NP.isfinite(pw))),
    'delay_power_scale_K2_Mpc3_per_Jy2Hz2': pconst['factor'], 'delay_power_max_K2_Mpc3': float(NP.max(pw))}))
