"""Host time of the steps of one observe() on config 2 (PRISIM_HIP_TRACE_ALLOC spans of the C library, averaged) + Python's share."""
import collections, os, re, subprocess, sys, time
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
    n = 200
    ia.reserve(n + 8)
    for j in range(8):
        ia.observe((2457000.5 + j * 1e-4, lst0 + j * 0.05), {'Tnet': 100.0}, NP.ones(ch.size), [0.0, lat], skymod, 10.0)
    ia._ctx.sync()
    sys.stderr.write('=== START ===\n')
    t0 = time.perf_counter()
    tc = 0.0
    orig = ia._ctx.observe_catalog
    def timed(*a, **k):
        global tc
        t1 = time.perf_counter()
        r = orig(*a, **k)
        tc += time.perf_counter() - t1
        return r
    ia._ctx.observe_catalog = timed
    for j in range(8, 8 + n):
        ia.observe((2457000.5 + j * 1e-4, lst0 + j * 0.05), {'Tnet': 100.0}, NP.ones(ch.size), [0.0, lat], skymod, 10.0)
    t_loop = time.perf_counter() - t0
    ia._ctx.sync()
    t_all = time.perf_counter() - t0
    sys.stderr.write('=== END ===\n')
    print('per snapshot: loop %.1f us, with final sync %.1f us, inside ctx.observe_catalog (ctypes + C) %.1f us' % (1e6 * t_loop / n, 1e6 * t_all / n, 1e6 * tc / n))
else:
    env = dict(os.environ, PRISIM_HIP_TRACE_ALLOC='1')
    r = subprocess.run([sys.executable, os.path.abspath(__file__), 'child'], env=env, capture_output=True, text=True)
    print(r.stdout.strip())
    err = r.stderr
    body = err[err.index('=== START ==='):err.index('=== END ===')] if '=== START ===' in err else err
    agg = collections.defaultdict(lambda: [0, 0.0])
    for m in re.finditer(r'\[prisim_hip host\] (.*?): ([0-9.]+) us', body):
        agg[m.group(1)][0] += 1; agg[m.group(1)][1] += float(m.group(2))
    for k, (c, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print('%-60s n=%4d mean %.1f us' % (k, c, tot / c))
    print('allocations during the loop:', len(re.findall(r'\[prisim_hip alloc\]', body)))
