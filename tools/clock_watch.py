"""Development helper: sample rocm-smi (clock, power, temperature) while the headline kernel runs back to back."""
import sys, os, time, subprocess, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as NP
from prisim_amd import _abi, workloads as W
cfg = W.config3(); bl, ch, sky = cfg['baselines'], cfg['channels'], cfg['sky']
zen = NP.array([0.0, 0.0, 1.0])
ctx = _abi.Context(0); ctx.set_array(bl, ch, nt_max=1)
ctx.set_sky_analytic(sky['dircos'], sky['flux_ref'], sky['spindex'], sky['ref_freq'], _abi.PRISIM_BEAM_AIRY, 14.0, zen, zen)
stop = False
def poll():
    while not stop:
        try:
            out = subprocess.run(['rocm-smi', '--showclocks', '--showpower', '--showtemp', '-d', '0'], capture_output=True, text=True, timeout=20).stdout
            keep = [l.strip() for l in out.splitlines() if any(k in l for k in ('sclk', 'Power', 'Temperature (Sensor junction)', 'Temperature (Sensor edge)', 'mclk'))]
            print(' | '.join(keep)[:400], flush=True)
        except Exception as exc:
            print('rocm-smi failed:', exc, flush=True)
        time.sleep(1.0)
print('idle:'); 
t = threading.Thread(target=poll); t.start(); time.sleep(2.5)
print('running 100 launches:', flush=True)
t0 = time.perf_counter()
for i in range(100):
    ctx.compute(precision=_abi.PRISIM_FP32)
ctx.sync()
dt = time.perf_counter() - t0
stop = True; t.join()
print('avg step %.2f ms over 100 launches' % (dt * 10))
