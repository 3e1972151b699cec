"""A/B: observe() per snapshot on config 2 -- the per-snapshot launch chain (default) against the count-free batched launch with ONE
snapshot (PRISIM_HIP_WAVE_BATCH_SINGLE=1), at several (tile, split) cuts.  us per snapshot, second pass of a resident instance."""
import json, os, subprocess, sys, time
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import numpy as NP
    import bench
    from prisim_amd import interferometry as RI, workloads as W
    cfg = W.config2()
    lat, lst0 = -30.7224, 40.0
    skymod = bench.radec_skymodel(cfg, lat, lst0)
    bl, ch = cfg['baselines'], cfg['channels']
    ia = RI.InterferometerArray(['b%d' % i for i in range(bl.shape[0])], bl, ch, telescope={'id': 'hera', 'orientation': [90.0, 270.0], 'ocoords': 'altaz'},
                                latitude=lat, skycoords='radec', pointing_coords='hadec')
    ct, ns = int(sys.argv[2]), int(sys.argv[3])
    if ct or ns:
        ia._ctx.set_tuning(ct, 0, ns)
    n = 256
    ia.reserve(2 * n)
    res = []
    for ps in range(2):
        ia._ctx.sync()
        t0 = time.perf_counter()
        for j in range(ps * n, (ps + 1) * n):
            ia.observe((2457000.5 + j * 1e-4, lst0 + j * 0.05), {'Tnet': 100.0}, NP.ones(ch.size), [0.0, lat], skymod, 10.0)
        th = time.perf_counter() - t0
        ia._ctx.sync()
        res.append((1e6 * th / n, 1e6 * (time.perf_counter() - t0) / n))
    tm = ia._ctx.timing()
    print(json.dumps({'single_via_batch': os.environ.get('PRISIM_HIP_WAVE_BATCH_SINGLE', '0'), 'tune': [ct, ns], 'host_us': res[1][0], 'wall_us': res[1][1],
                      'chan_tile': tm['last_chan_tile'], 'nsplit': tm['last_nsplit'], 'per_launch': tm['last_batch_snapshots']}))
else:
    for env, ct, ns in (('0', 0, 0), ('1', 0, 0), ('1', 16, 42), ('1', 16, 24), ('1', 32, 16), ('1', 16, 12), ('1', 32, 8)):
        e = dict(os.environ, PRISIM_HIP_WAVE_BATCH_SINGLE=env)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), 'child', str(ct), str(ns)], env=e, capture_output=True, text=True)
        print(r.stdout.strip() or r.stderr[-400:], flush=True)
