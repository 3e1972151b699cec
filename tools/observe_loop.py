"""A plain loop of observe() on config 2 (HERA-19), for a kernel trace: rocprofv3 --kernel-trace -- python3 tools/observe_loop.py [n] [mode]
mode: plain | memsave | grad.  Prints the loop's us per snapshot."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as NP
import bench
from prisim_amd import interferometry as RI, workloads as W

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
mode = sys.argv[2] if len(sys.argv) > 2 else 'plain'
kw = {'memsave': True} if mode == 'memsave' else ({'gradient_mode': 'baseline'} if mode == 'grad' else {})
cfg = W.config2()
lat, lst0 = -30.7224, 40.0
skymod = bench.radec_skymodel(cfg, lat, lst0)
bl, ch = cfg['baselines'], cfg['channels']
ia = RI.InterferometerArray(['b%d' % i for i in range(bl.shape[0])], bl, ch, telescope={'id': 'hera', 'orientation': [90.0, 270.0], 'ocoords': 'altaz'},
                            latitude=lat, skycoords='radec', pointing_coords='hadec')
ia.reserve(2 * n)
for ps in range(2):
    ia._ctx.sync()
    t0 = time.perf_counter()
    for j in range(ps * n, (ps + 1) * n):
        ia.observe((2457000.5 + j * 1e-4, lst0 + j * 0.05), {'Tnet': 100.0}, NP.ones(ch.size), [0.0, lat], skymod, 10.0, **kw)
    th = time.perf_counter() - t0
    ia._ctx.sync()
    print('pass %d: host %.1f us, wall %.1f us per snapshot' % (ps, 1e6 * th / n, 1e6 * (time.perf_counter() - t0) / n), flush=True)
ia.close()
