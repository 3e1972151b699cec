"""Development helper: quick GPU-vs-oracle parity sweep over kernel variants (not a test)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as NP
from prisim_amd import _abi
from oracle import skyvis_oracle as O

rng = NP.random.default_rng(7)

def make(nbl, nchan, nsrc, maxbl=300.0, df=97656.25, f0=100e6, fwhm=False):
    bl = rng.uniform(-maxbl, maxbl, size=(nbl, 3)); bl[:, 2] *= 0.01
    freqs = f0 + NP.arange(nchan) * df
    alt = NP.degrees(NP.arcsin(rng.uniform(NP.sin(NP.radians(5.0)), 1.0, nsrc)))
    az = rng.uniform(0, 360, nsrc)
    dc = O.altaz2dircos(NP.stack((alt, az), axis=1))
    pb = rng.uniform(0.0, 10.0, size=(nsrc, 1)) * (freqs / 150e6).reshape(1, -1) ** -0.8 * rng.uniform(0.2, 1.0, size=(nsrc, nchan))
    pc = O.altaz2dircos(NP.array([[80.0, 30.0]]))[0]
    fw = rng.uniform(0.0, 1.5, nsrc) if fwhm else None
    if fwhm: fw[::5] = 0.0
    return bl, freqs, dc, pb, pc, fw

ctx = _abi.Context(0)
print(ctx.device_info())
fails = 0
for (nbl, nchan, nsrc) in [(3, 64, 100), (171, 256, 1536), (300, 100, 777), (1000, 1024, 130)]:
    for fwhm in (False, True):
        bl, freqs, dc, pb, pc, fw = make(nbl, nchan, nsrc, fwhm=fwhm)
        ref, gref = O.skyvis(bl, freqs, dc, pb, pc, fwhm_deg=fw, gradient=True)
        scale = O.abs_flux_sum(pb)[None, :]
        ctx.set_array(bl, freqs, nt_max=1)
        for prec, tol in ((_abi.PRISIM_FP64, 1e-11), (_abi.PRISIM_FP32, 5e-6)):
            for kern in (_abi.PRISIM_KERNEL_RECURRENCE, _abi.PRISIM_KERNEL_DIRECT):
                if kern == _abi.PRISIM_KERNEL_DIRECT and prec == _abi.PRISIM_FP32: continue
                cts = [0, 8, 16, 32] + ([64] if prec == _abi.PRISIM_FP32 else [])
                if kern == _abi.PRISIM_KERNEL_DIRECT: cts = [0]
                for ct in cts:
                    for nsplit in (0, 1, 3):
                        if kern == _abi.PRISIM_KERNEL_DIRECT and nsplit: continue
                        ctx.set_tuning(ct, 0, nsplit)
                        t0 = time.time()
                        v, g = ctx.skyvis(dc, pb, pc, fwhm_deg=fw, precision=prec, kernel=kern, want_grad=True)
                        dt = time.time() - t0
                        err = NP.max(NP.abs(v - ref) / scale)
                        gerr = NP.max(NP.abs(g - gref) / scale[None])
                        tm = ctx.timing()
                        ok = err <= tol and gerr <= tol
                        fails += (not ok)
                        print('%s nbl=%d nchan=%d nsrc=%d taper=%d prec=%d kern=%d ct=%d(%d) nsplit=%d(%d) err=%.2e gerr=%.2e kern_ms=%.3f wall=%.3f' % (
                            'ok  ' if ok else 'FAIL', nbl, nchan, nsrc, fwhm, prec, kern, ct, tm['last_chan_tile'], nsplit, tm['last_nsplit'], err, gerr, tm['last_kernel_ms'], dt))
print('FAILS', fails)
sys.exit(1 if fails else 0)
