"""A/B through the class surface on config 2 (HERA-19): 256 observe() calls and one observe_batch of 256 accumulations with memsave=True
(fp32 request) and with gradient_mode='baseline', each through the batched launch and through the per-snapshot chain
(PRISIM_HIP_BATCH_FP32_AS_FP64=0 / PRISIM_HIP_WAVE_BATCH_GRAD=0).  us per snapshot, second pass of a resident instance."""
import json, os, subprocess, sys, time
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import numpy as NP
    import bench
    from prisim_amd import interferometry as RI, workloads as W
    mode = sys.argv[2]
    cfg = W.config2()
    lat, lst0 = -30.7224, 40.0
    skymod = bench.radec_skymodel(cfg, lat, lst0)
    bl, ch = cfg['baselines'], cfg['channels']
    kw = {'memsave': True} if mode == 'memsave' else ({'gradient_mode': 'baseline'} if mode == 'grad' else {})
    out = {'mode': mode, 'env': {k: v for k, v in os.environ.items() if k.startswith('PRISIM_HIP_')}}
    n = 256
    for what in ('observe', 'observe_batch'):
        ia = RI.InterferometerArray(['b%d' % i for i in range(bl.shape[0])], bl, ch, telescope={'id': 'hera', 'orientation': [90.0, 270.0], 'ocoords': 'altaz'},
                                    latitude=lat, skycoords='radec', pointing_coords='hadec')
        ia.reserve(2 * n)
        res = []
        for ps in range(2):
            ia._ctx.sync()
            t0 = time.perf_counter()
            if what == 'observe':
                for j in range(ps * n, (ps + 1) * n):
                    ia.observe((2457000.5 + j * 1e-4, lst0 + j * 0.05), {'Tnet': 100.0}, NP.ones(ch.size), [0.0, lat], skymod, 10.0, **kw)
            else:
                times = [(2457000.5 + j * 1e-4, lst0 + j * 0.05) for j in range(ps * n, (ps + 1) * n)]
                ia.observe_batch(times, {'Tnet': 100.0}, NP.ones(ch.size), [0.0, lat], skymod, 10.0, **kw)
            ia._ctx.sync()
            res.append(1e6 * (time.perf_counter() - t0) / n)
        tm = ia._ctx.timing()
        out[what] = {'us_per_snapshot': round(res[1], 2), 'first_pass_us': round(res[0], 2), 'per_launch': tm['last_batch_snapshots'],
                     'chan_tile': tm['last_chan_tile'], 'nsplit': tm['last_nsplit']}
        ia.close()
    print(json.dumps(out))
else:
    for mode, env in (('plain', {}), ('memsave', {}), ('memsave', {'PRISIM_HIP_BATCH_FP32_AS_FP64': '0'}), ('grad', {}), ('grad', {'PRISIM_HIP_WAVE_BATCH_GRAD': '0'})):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), 'child', mode], env=dict(os.environ, **env), capture_output=True, text=True)
        print(r.stdout.strip() or r.stderr[-600:], flush=True)
