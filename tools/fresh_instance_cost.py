"""Development helper: where the first snapshots of a FRESH InterferometerArray go (config 4, rank 0 of 8, fp32): catalogue upload, first
allocations, streams / pinned buffers, against the same calls on the instance once its state is resident.   python tools/fresh_instance_cost.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as NP

import bench
from prisim_amd import interferometry as RI, sharding, workloads as W

n_acc, nranks = 32, int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = W.config4(n_acc=n_acc)
tel = {'id': 'custom', 'shape': 'delta', 'size': 1.0, 'ocoords': 'altaz', 'orientation': NP.array([[90.0, 270.0]]), 'groundplane': None}
lat, lst0 = cfg['latitude'], 30.0
skymod = bench.radec_skymodel(cfg, lat, lst0)
bl = sharding.shard_rows(cfg['baselines'], nranks, 0)[0]
ch = cfg['channels']
dlst = cfg['t_acc'] * 360.0 * 1.00273790935 / 86400.0
tsys, bp, pc = {'Tnet': 100.0}, NP.ones(ch.size), NP.array([0.0, lat])
for rep in range(3):
    t = [time.perf_counter()]
    ia = RI.InterferometerArray(['b%d' % i for i in range(bl.shape[0])], bl, ch, telescope=tel, latitude=lat, skycoords='radec', pointing_coords='hadec', device=0)
    ia.reserve(4 * n_acc)
    ia.set_external_beam(cfg['beam_table'], cfg['beam_freqs'])
    ia._ctx.sync()
    t.append(time.perf_counter())
    lsts = lst0 + NP.arange(4 * n_acc) * dlst
    times = [(2455000.0 + j * cfg['t_acc'] / 86400.0, float(lsts[j])) for j in range(4 * n_acc)]
    marks = ['create+reserve+beam']
    for lo, hi in ((0, 1), (1, 2), (2, 16), (16, 32), (32, 64), (64, 96), (96, 128)):
        ia.observe_batch(times[lo:hi], tsys, bp, pc, skymod, cfg['t_acc'], memsave=True)
        ia._ctx.sync()
        t.append(time.perf_counter())
        marks.append('snapshots %d-%d' % (lo, hi))
    print('rep', rep, ' | '.join('%s: %.2f ms (%.2f / snapshot)' % (m, 1e3 * (b - a), 1e3 * (b - a) / max(1, (int(m.split('-')[-1]) - int(m.split()[-1].split('-')[0])) if m.startswith('snap') else 1))
                                 for m, a, b in zip(marks, t[:-1], t[1:])), flush=True)
    ia._ctx.close()
    del ia

# host profile of the very first snapshot of a fresh instance (time inside the ctypes calls = allocations, uploads, stream creation)
import cProfile
import pstats
ia = RI.InterferometerArray(['b%d' % i for i in range(bl.shape[0])], bl, ch, telescope=tel, latitude=lat, skycoords='radec', pointing_coords='hadec', device=0)
ia.reserve(n_acc)
ia.set_external_beam(cfg['beam_table'], cfg['beam_freqs'])
ia._ctx.sync()
pr = cProfile.Profile()
pr.enable()
ia.observe_batch(times[0:1], tsys, bp, pc, skymod, cfg['t_acc'], memsave=True)
ia._ctx.sync()
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(14)
