"""Development helper: randomized sweep of the device-resident catalogue path (prisim_hip_set_catalog / _observe_catalog / _catalog_roi)
against the host statement of the snapshot's frame (prisim_amd/geometry.py frame_dircos / roi_select: index lists and direction cosines
must be bit-identical -- fall-back rotation, explicit 'date' frames and apparent-place frames of prisim_amd/frames.py) and the uploaded-sky path (set_sky_analytic + compute on the host-formed sky) --
array sizes around the wave / block / batch boundaries, catalogue sizes around the 256-source block boundaries, coordinate systems,
regions of interest, source runs, flux spectra, several snapshots per call (loop chunks and the batched launch), both precisions, gradients.

    python tools/fuzz_catalog.py [seed] [cases]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as NP

from prisim_amd import _abi, frames as FR, geometry as GEOM, primary_beams as PB, workloads as W

rng = NP.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 100
TOL_VS_UPLOADED = {_abi.PRISIM_FP64: 2e-13, _abi.PRISIM_FP32: 3e-6}     # (fp32: the altitude-sorted order sums in another order)
TOL_EXT = 2e-7          # external beam: stored as float32 (:4466); direction cosines an ulp apart can land on either side of a rounding boundary
fails = 0
stats = {'external beam': 0, 'batched launches': 0, 'snapshots compared': 0, 'gradient': 0}
t0 = time.time()
ctx, ref = _abi.Context(0), _abi.Context(0)
EXT_BEAM = W.synthetic_healpix_beam(8, NP.linspace(80e6, 260e6, 7))
for case in range(ncase):
    nbl = int(rng.choice([1, 3, 63, 64, 65, 171, 255, 256, 257, 400, 700]))
    nchan = int(rng.choice([16, 17, 31, 32, 40, 64, 96, 128]))
    ncat = int(rng.choice([0, 1, 2, 255, 256, 257, 511, 512, 513, 1000, 3000, 9000]))
    k = int(rng.choice([1, 1, 2, 3, 5, 9, 17]))
    coords = str(rng.choice(['radec', 'radec', 'hadec', 'altaz']))
    lat = float(rng.choice([-30.7224, -26.701, 0.0, 34.079, 71.0]))
    maxbl = float(rng.choice([20.0, 300.0, 2000.0]))
    bl = rng.uniform(-maxbl, maxbl, size=(nbl, 3))
    bl[:, 2] *= 0.02
    bl = bl[NP.argsort(NP.sqrt(NP.sum(bl ** 2, axis=1)))]
    ch = float(rng.choice([100e6, 150e6, 185e6])) + NP.arange(nchan) * float(rng.choice([40e3, 97656.25, 390625.0]))
    if coords == 'altaz':
        loc = NP.stack((rng.uniform(-20.0, 90.0, ncat), rng.uniform(0.0, 360.0, ncat)), axis=1)
    else:
        loc = NP.stack((rng.uniform(0.0, 360.0, ncat) - (180.0 if coords == 'hadec' else 0.0), NP.degrees(NP.arcsin(rng.uniform(-1.0, 1.0, ncat)))), axis=1)
    shape_kind = int(rng.integers(0, 4))          # 0: no shapes, 1: sizes vary, 2: one run, 3: up to three runs (point sources + maps)
    fw = None
    if shape_kind == 1:
        fw = rng.uniform(0.0, 1.2, ncat) * (rng.uniform(size=ncat) > 0.2)
    elif shape_kind == 2:
        fw = NP.full(ncat, float(rng.choice([0.0, 0.229, 0.916, 3.66])))
    elif shape_kind == 3:
        cuts = NP.sort(rng.integers(0, ncat + 1, 2))
        fw = NP.zeros(ncat)
        fw[cuts[0]:cuts[1]] = float(rng.choice([0.229, 0.916]))
        fw[cuts[1]:] = float(rng.choice([0.458, 3.66]))
    spectra = rng.integers(0, 3) == 0
    flux_ref, spindex = rng.uniform(0.5, 10.0, ncat), rng.uniform(-1.2, 0.0, ncat)
    spec = (flux_ref[:, None] * (ch[None, :] / 150e6) ** spindex[:, None] * rng.uniform(0.8, 1.2, (ncat, nchan))) if spectra else None
    roi_radius = float(rng.choice([90.0, 90.0, 75.0, 30.0]))
    roi_center = 'pointing_center' if rng.integers(0, 5) == 0 else 'zenith'
    beam = int(rng.choice([_abi.PRISIM_BEAM_DELTA, _abi.PRISIM_BEAM_GAUSSIAN, _abi.PRISIM_BEAM_AIRY]))
    ext_beam = bool(rng.integers(0, 4) == 0)         # the external HEALPix beam instead (its own batched preparation)
    lsts = rng.uniform(0.0, 360.0, k)
    pcs = GEOM.altaz2dircos(NP.stack((rng.uniform(60.0, 90.0, k), rng.uniform(0.0, 360.0, k)), axis=1), 'degrees')
    want_grad = bool(rng.integers(0, 6) == 0)
    # the snapshots' frames: the library's fall-back rotation, or explicit frames -- the same rotation, or a full apparent-place frame
    frame_kind = str(rng.choice(['fallback', 'date', 'apparent', 'mean']))
    jd0 = float(rng.uniform(2451545.0, 2466154.0))              # 2000 ... 2040
    host_frames = [FR.snapshot_frame(coords, float(lsts[t]), lat, jd=jd0 + 0.01 * t, epoch='J2000', model=('date' if frame_kind == 'fallback' else frame_kind))
                   for t in range(k)]
    frames = None if frame_kind == 'fallback' else host_frames
    what = dict(case=case, nbl=nbl, nchan=nchan, ncat=ncat, k=k, coords=coords, lat=lat, shape_kind=shape_kind, spectra=bool(spectra), roi=(roi_radius, roi_center),
                beam=beam, ext_beam=ext_beam, grad=want_grad, maxbl=maxbl, frame=frame_kind)
    stats['external beam'] += int(ext_beam)
    stats['gradient'] += int(want_grad)
    try:
        for c, nt in ((ctx, k), (ref, 1)):
            c.set_array(bl, ch, nt_max=nt)
            if ext_beam:
                c.set_external_beam(EXT_BEAM, PB.spectral_interp_matrix(NP.linspace(80e6, 260e6, 7), ch, kind='cubic', chromatic=True, select_freq=None))
        if spectra:
            ctx.set_catalog(loc, coords, flux_spectrum=spec, fwhm_deg=fw)
        else:
            ctx.set_catalog(loc, coords, flux_ref=flux_ref, spindex=spindex, ref_freq_hz=150e6, fwhm_deg=fw)
        if ext_beam:
            obs = ctx.make_obs(lat, roi_radius_deg=roi_radius, roi_center=roi_center, use_external_beam=True)
        else:
            obs = ctx.make_obs(lat, roi_radius_deg=roi_radius, roi_center=roi_center, beam_kind=beam, diameter_m=14.0)
        for prec in (_abi.PRISIM_FP64, _abi.PRISIM_FP32):
            counts = ctx.observe_catalog(obs, lsts, pcs, pcs, precision=prec, want_grad=want_grad, frames=frames)
            batched = ctx.timing()['last_batch_snapshots'] if k > 1 else 1
            stats['batched launches'] += int(batched > 1)
            for t in range(k):
                # the host statement of the device's cat_source(): bit-identical by construction, no rim exceptions
                dc_all = GEOM.frame_dircos(GEOM.catalog_unitvec(loc, coords), host_frames[t][0], host_frames[t][1]) if ncat else NP.zeros((0, 3))
                m2 = GEOM.roi_select(dc_all, roi_center, roi_radius, pcs[t]) if ncat else NP.zeros(0, dtype=NP.int64)
                altaz = GEOM.dircos2altaz(dc_all) if ncat else NP.zeros((0, 2))
                if counts[t] != m2.size:
                    raise AssertionError('ROI count %d vs host %d (snapshot %d)' % (counts[t], m2.size, t))
                if prec == _abi.PRISIM_FP64 and t == 0:
                    idx, dc = ctx.catalog_roi(obs, lsts[t], pcs[t], frame=None if frames is None else frames[t])
                    if not NP.array_equal(idx, m2):
                        raise AssertionError('index lists differ')
                    if m2.size and not NP.array_equal(dc, dc_all[m2]):
                        raise AssertionError('dircos differ by %.2e' % float(NP.max(NP.abs(dc - dc_all[m2]))))
                got = ctx.get_vis(slot=t, want_grad=want_grad)
                if m2.size == 0:
                    v = got[0] if want_grad else got
                    if NP.any(v != 0):
                        raise AssertionError('empty ROI but non-zero visibilities')
                    continue
                # the uploaded path on the host-formed sky (the order the catalogue path uses when it sorts by altitude does not matter to the sum)
                fwm = None if fw is None else fw[m2]
                if ext_beam:
                    ref.set_sky_external_analytic(dc_all[m2], None if spectra else flux_ref[m2], None if spectra else spindex[m2], None if spectra else 150e6,
                                                  pcs[t], fwhm_deg=fwm, flux_spectrum=spec[m2] if spectra else None)
                elif spectra:
                    ref.set_sky_analytic(dc_all[m2], None, None, None, beam, 14.0, pcs[t], pcs[t], fwhm_deg=fwm, flux_spectrum=spec[m2])
                else:
                    ref.set_sky_analytic(dc_all[m2], flux_ref[m2], spindex[m2], 150e6, beam, 14.0, pcs[t], pcs[t], fwhm_deg=fwm)
                ref.compute(precision=prec, want_grad=want_grad)
                want = ref.get_vis(want_grad=want_grad)
                # (fp32: a sky whose whole beam-weighted flux lies below the normal range of a float -- one source 60 degrees off a Gaussian
                # beam is 1e-50 Jy -- underflows in the fp32 chain of the uploaded path and not in the batched launch, which serves fp32
                # requests of small arrays in fp64: relative to the range of the type there)
                scale = NP.maximum(NP.sum(NP.abs(ref.get_pbflux()), axis=0), 1e-300 if prec == _abi.PRISIM_FP64 else 1e-30)[None, :]
                pairs = [(got[0], want[0])] + [(got[1][i], want[1][i]) for i in range(3)] if want_grad else [(got, want)]
                # The two paths' direction cosines differ by a few ulp (tests/test_gpu_catalog.py: <= 2.5e-15, and 2 ulp / cos(alt) near the
                # zenith, where the reference's chain alt = arcsin(.), cos(alt) is ill-conditioned on the host and on the device alike).  On a
                # baseline of length L that is a phase difference of up to 2 pi L delta_s f / c for source s: the first-order bound on the
                # difference of the sums, relative to sum|pbflux|, is the |pbflux|-weighted mean of that -- 8e-12 at 2 km and 200 MHz from the
                # ulps alone, the same order as the fp64 tolerance itself.
                lmax = float(NP.max(NP.sqrt(NP.sum(bl ** 2, axis=1))))
                pbr = NP.abs(ref.get_pbflux())
                delta = 2.5e-15 + 4.5e-16 / NP.maximum(NP.cos(NP.radians(altaz[m2, 0])), 1e-6)
                bound = float(NP.max(NP.sum(pbr * delta[:, None], axis=0) / NP.maximum(NP.sum(pbr, axis=0), 1e-300))) * 2.0 * NP.pi * lmax * float(ch[-1]) / 299792458.0
                # (one source beside a null of the Airy pattern also turns 1e-15 of direction into 1e-12 of its own beam value)
                tol = (max(TOL_VS_UPLOADED[prec], TOL_EXT if ext_beam else 0.0) + bound) * max(1.0, 32.0 / m2.size)
                stats['snapshots compared'] += 1
                for a, b in pairs:
                    err = float(NP.max(NP.abs(a - b) / scale))
                    if not err <= tol:
                        raise AssertionError('snapshot %d prec %d: %.3e (batched %d)' % (t, prec, err, batched))
    except Exception as exc:      # noqa
        fails += 1
        print('FAIL', what, repr(exc), flush=True)
print('fuzz_catalog: %d cases, %d fails, %.1f s; %s' % (ncase, fails, time.time() - t0, stats))
sys.exit(1 if fails else 0)
