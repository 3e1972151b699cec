"""Development helper: randomized GPU-vs-oracle parity sweep over shapes, precisions, tapers and tuning knobs."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as NP
from prisim_amd import _abi
from oracle import c_oracle as CO, skyvis_oracle as O

rng = NP.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 150
ctx = _abi.Context(0)
TOL = {_abi.PRISIM_FP64: 1e-11, _abi.PRISIM_FP32: 5e-6}
fails = 0
t0 = time.time()
for case in range(ncase):
    nbl = int(rng.choice([1, 2, 63, 64, 65, 200, 256, 257, 511, 700]))
    nchan = int(rng.choice([1, 7, 8, 16, 31, 33, 64, 65, 100, 128, 200]))
    nsrc = int(rng.choice([0, 1, 5, 31, 63, 64, 65, 71, 72, 100, 127, 500, 1500, 3000, 17000, 40000]))
    if nsrc > 3000:
        pass
    taper = bool(rng.integers(0, 2))
    maxbl = float(rng.choice([20.0, 300.0, 1500.0]))
    df = float(rng.choice([24414.0625, 97656.25, 390625.0, 1.5e6]))
    f0 = float(rng.choice([50e6, 100e6, 180e6]))
    bl = rng.uniform(-maxbl, maxbl, size=(nbl, 3)); bl[:, 2] *= 0.02
    if rng.integers(0, 2):
        bl = bl[NP.argsort(NP.sqrt(NP.sum(bl ** 2, axis=1)))]      # by length, as the driver lists them
    ch = f0 + NP.arange(nchan) * df
    alt = NP.degrees(NP.arcsin(rng.uniform(0.05, 1.0, nsrc)))
    if rng.integers(0, 2):
        alt = NP.sort(alt)[::-1]          # by decreasing altitude, as observe() lists a run when the taper culling can skip sources
    dc = O.altaz2dircos(NP.stack((alt, rng.uniform(0, 360, nsrc)), axis=1)) if nsrc else NP.zeros((0, 3))
    pb = rng.uniform(0.0, 10.0, size=(nsrc, 1)) * rng.uniform(0.2, 1.0, size=(nsrc, nchan))
    pc = O.altaz2dircos(NP.array([[rng.uniform(40, 90), rng.uniform(0, 360)]]))[0]
    fw = (rng.uniform(0.0, 1.0, nsrc) * (rng.uniform(size=nsrc) > 0.2)) if taper else None
    if taper and nsrc > 0 and rng.integers(0, 2):
        # sources in one to three runs of one size each (HEALPix skies, point sources + diffuse): the split taper form of the packed
        # fp32 kernel when it gets 64-channel tiles, small sizes so that its exponent guard does not always send it back
        nrun = int(rng.integers(1, 4))
        cuts = NP.sort(rng.integers(0, nsrc + 1, nrun - 1)) if nrun > 1 else NP.zeros(0, dtype=int)
        sizes = rng.choice([0.0, 0.03, 0.1, 0.229, 0.458], nrun)
        fw = NP.zeros(nsrc)
        lo = 0
        for r in range(nrun):
            hi = int(cuts[r]) if r < nrun - 1 else nsrc
            fw[lo:hi] = sizes[r]
            lo = hi
    cull_case = nsrc > 64 and rng.integers(0, 6) == 0
    if cull_case:
        # a case built for the taper culling: long baselines listed by length, degree-size sources listed by decreasing altitude
        taper = True
        bl = rng.normal(0.0, 900.0, size=(nbl, 3)); bl[:, 2] *= 0.002
        bl = bl[NP.argsort(NP.sqrt(NP.sum(bl ** 2, axis=1)))]
        alt = NP.sort(alt)[::-1]
        dc = O.altaz2dircos(NP.stack((alt, rng.uniform(0, 360, nsrc)), axis=1))
        fw = NP.full(nsrc, float(rng.choice([0.458, 0.916])))
    ref = CO.skyvis(bl, ch, dc, pb, pc, fwhm_deg=fw) if nsrc else NP.zeros((nbl, nchan), dtype=complex)
    scale = NP.maximum(NP.sum(NP.abs(pb), axis=0), 1e-300)[None, :]
    ctx.set_array(bl, ch)
    ctx.set_sky(dc, pb, pc, fwhm_deg=fw)
    for prec in (_abi.PRISIM_FP64, _abi.PRISIM_FP32):
        cts = [0, 8, 16, 32] + ([64] if prec == _abi.PRISIM_FP32 else [])
        ct = int(rng.choice(cts)); nsplit = int(rng.choice([0, 1, 2, 5])); chunk = int(rng.choice([0, 1, 16, 64]))
        if cull_case and prec == _abi.PRISIM_FP32:
            ct = int(rng.choice([32, 64]))
        flush = int(rng.choice([0, 0, 0, 7, 64]))
        if flush: os.environ['PRISIM_HIP_FLUSH_SRC'] = str(flush)
        else: os.environ.pop('PRISIM_HIP_FLUSH_SRC', None)
        ctx.set_tuning(ct, chunk, nsplit)
        grad = bool(rng.integers(0, 4) == 0) and nsrc > 0
        ctx.compute(precision=prec, want_grad=grad)
        v = ctx.get_vis(want_grad=grad)
        if grad:
            v, g = v
            for k in range(3):
                gref = CO.skyvis(bl, ch, dc, pb * dc[:, k:k + 1], pc, fwhm_deg=fw)
                gerr = float(NP.max(NP.abs(g[k] - gref) / scale))
                if not (NP.all(NP.isfinite(g[k])) and gerr <= TOL[prec]):
                    fails += 1
                    print('FAIL-GRAD k=%d nbl=%d nchan=%d nsrc=%d taper=%d prec=%d ct=%d nsplit=%d flush=%d err=%.3e' % (k, nbl, nchan, nsrc, taper, prec, ct, nsplit, flush, gerr), flush=True)
        err = float(NP.max(NP.abs(v - ref) / scale)) if nsrc else float(NP.max(NP.abs(v)))
        ok = NP.all(NP.isfinite(v)) and err <= TOL[prec]
        if ctx.timing().get('last_culled_fraction', 0.0) > 0:
            globals()['nculled'] = globals().get('nculled', 0) + 1
        if ctx.timing().get('last_taper_split', 0) > 0:
            nsplitform = globals().get('nsplitform', 0) + 1
            globals()['nsplitform'] = nsplitform
        if not ok:
            fails += 1
            print('FAIL nbl=%d nchan=%d nsrc=%d taper=%d maxbl=%g df=%g f0=%g prec=%d ct=%d nsplit=%d chunk=%d flush=%d err=%.3e t=%s' %
                  (nbl, nchan, nsrc, taper, maxbl, df, f0, prec, ct, nsplit, chunk, flush, err, ctx.timing()), flush=True)
    if case % 25 == 0:
        print('case', case, 'fails', fails, '%.0fs' % (time.time() - t0), flush=True)
print('DONE cases', ncase, 'fails', fails, 'cases that ran the split taper form', globals().get('nsplitform', 0), 'cases with taper culling', globals().get('nculled', 0))
