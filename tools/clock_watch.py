"""Development helper: sample rocm-smi (clock, power, temperature) while one of the sky-sum kernels runs back to back.

    python tools/clock_watch.py [fp32|fp64] [cfg3|cfg3d] [launches]

cfg3 = headline sky (no taper), cfg3d = config 3 with its diffuse half (taper on)."""
import sys, os, time, subprocess, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as NP
from prisim_amd import _abi, workloads as W
prec_name = sys.argv[1] if len(sys.argv) > 1 else 'fp32'
wl = sys.argv[2] if len(sys.argv) > 2 else 'cfg3'
nlaunch = int(sys.argv[3]) if len(sys.argv) > 3 else 100
prec = _abi.PRISIM_FP32 if prec_name == 'fp32' else _abi.PRISIM_FP64
cfg = W.config3(with_diffuse=(wl == 'cfg3d')); bl, ch, sky = cfg['baselines'], cfg['channels'], cfg['sky']
zen = NP.array([0.0, 0.0, 1.0])
ctx = _abi.Context(0); ctx.set_array(bl, ch, nt_max=1)
ctx.set_sky_analytic(sky['dircos'], sky['flux_ref'], sky['spindex'], sky['ref_freq'], _abi.PRISIM_BEAM_AIRY, 14.0, zen, zen,
                     fwhm_deg=(sky['fwhm_deg'] if cfg['taper'] else None))
stop = False
def poll():
    while not stop:
        try:
            out = subprocess.run(['rocm-smi', '--showclocks', '--showpower', '--showtemp', '-d', '0'], capture_output=True, text=True, timeout=20).stdout
            keep = [l.strip() for l in out.splitlines() if any(k in l for k in ('sclk', 'Power', 'Temperature (Sensor junction)', 'mclk'))]
            print(' | '.join(keep)[:400], flush=True)
        except Exception as exc:
            print('rocm-smi failed:', exc, flush=True)
        time.sleep(1.0)
print('# %s %s: %s, %d sources, taper %s' % (prec_name, wl, cfg['name'], sky['dircos'].shape[0], cfg['taper']))
print('idle:');
t = threading.Thread(target=poll); t.start(); time.sleep(2.5)
print('running %d launches:' % nlaunch, flush=True)
ctx.compute(precision=prec); ctx.sync(); ctx.timing(reset=True)
t0 = time.perf_counter()
for i in range(nlaunch):
    ctx.compute(precision=prec)
ctx.sync()
dt = time.perf_counter() - t0
stop = True; t.join()
tm = ctx.timing()
print('avg step %.2f ms over %d launches; kernel avg %.2f ms' % (dt * 1e3 / nlaunch, nlaunch, tm['sum_kernel_ms'] / max(1, tm['n_kernel'])))
