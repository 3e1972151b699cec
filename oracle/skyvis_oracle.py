"""numpy fp64 restatement of the PRISim per-baseline sky-sum (TEST INFRASTRUCTURE).

Follows, statement by statement, the reference hot path
``InterferometerArray.observe`` in /root/reference/prisim/interferometry.py:

  * phase-centre delay offsets            :6155-6167
  * pbfluxes = pb * fluxes                :6254
  * geometric delays tau = dc . bl^T / c  :6255  -> baseline_delay_horizon.py:236-240
  * source-shape taper                    :6257-6283
  * fp64 sum                              :6332-6343
  * fp32 (``memsave``) sum                :6323-6330
  * source-slab serialisation             :6348-6376

It never allocates more than ``slab_bytes`` for the (nsrc, nbl, nchan) temporary:
sources are processed in slabs exactly like the reference's memory-shortage loop
(:6348-6376), whose result equals the one-shot sum up to fp addition order.

This module is the checker.  It is not imported by the product (prisim_amd).
"""
import numpy as NP

C_LIGHT = 299792458.0   # scipy.constants.c, used at baseline_delay_horizon.py:236


def altaz2dircos(altaz_deg):
    """(alt, az) in degrees -> ENU direction cosines.

    astroutils.geometry.altaz2dircos is not in the reference tree (PARITY UNPINNED);
    the convention is fixed by in-tree statements: altaz [0,90] == dircos [1,0,0]
    (primary_beams.py:255-258), altaz [90,270] == dircos [0,0,1] (:275-278),
    dircos "aligned with local East, North, Up" (:122-123), alt is column 0 (:579).
    """
    altaz = NP.asarray(altaz_deg, dtype=NP.float64).reshape(-1, 2)
    alt = NP.radians(altaz[:, 0])
    az = NP.radians(altaz[:, 1])
    return NP.stack((NP.cos(alt) * NP.sin(az), NP.cos(alt) * NP.cos(az), NP.sin(alt)), axis=1)


def hadec2altaz(hadec_deg, latitude_deg):
    """(HA, Dec) degrees -> (alt, az) degrees, az from North through East.

    Stands in for astroutils.geometry.hadec2altaz (PARITY UNPINNED) used at
    interferometry.py:6157.  Standard spherical astronomy.
    """
    hadec = NP.asarray(hadec_deg, dtype=NP.float64).reshape(-1, 2)
    ha = NP.radians(hadec[:, 0])
    dec = NP.radians(hadec[:, 1])
    lat = NP.radians(latitude_deg)
    sin_alt = NP.sin(dec) * NP.sin(lat) + NP.cos(dec) * NP.cos(lat) * NP.cos(ha)
    alt = NP.arcsin(NP.clip(sin_alt, -1.0, 1.0))
    # East, North components of the unit vector
    east = -NP.cos(dec) * NP.sin(ha)
    north = NP.sin(dec) * NP.cos(lat) - NP.cos(dec) * NP.sin(lat) * NP.cos(ha)
    az = NP.arctan2(east, north)
    az = NP.where(az < 0.0, az + 2 * NP.pi, az)
    return NP.degrees(NP.stack((alt, az), axis=1))


def geometric_delay(baselines, dircos):
    """tau[s,b] = dc . bl^T / c   (baseline_delay_horizon.py:236-240, dircos path)."""
    baselines = NP.asarray(baselines, dtype=NP.float64).reshape(-1, 3)
    dircos = NP.asarray(dircos, dtype=NP.float64).reshape(-1, 3)
    return NP.dot(dircos, baselines.T) / C_LIGHT


def taper_weights(baselines, geometric_delays, channels, fwhm_deg):
    """Source-shape visibility weights (interferometry.py:6262-6283).

    fwhm_deg = sqrt(maj*min) per source (:6267).  Returns (nsrc, nbl, nchan).
    Zero-size sources give sigma=inf -> w=1 (reference reaches this through a
    divide-by-zero warning; here it is silenced).
    """
    baselines = NP.asarray(baselines, dtype=NP.float64).reshape(-1, 3)
    channels = NP.asarray(channels, dtype=NP.float64).ravel()
    baseline_lengths = NP.sqrt(NP.sum(baselines ** 2, axis=1))          # :5684
    wl = C_LIGHT / channels                                              # :6262
    with NP.errstate(divide='ignore', invalid='ignore'):
        psf = NP.sqrt(baseline_lengths.reshape(1, -1, 1) ** 2
                      - (C_LIGHT * geometric_delays[:, :, NP.newaxis]) ** 2) / wl.reshape(1, 1, -1)   # :6265
        src_fwhm_dircos = 2.0 * NP.sin(0.5 * NP.radians(NP.asarray(fwhm_deg, dtype=NP.float64))).reshape(-1, 1)  # :6268
        sigma = 1.0 / NP.sqrt(2.0 * NP.log(2.0)) / src_fwhm_dircos      # :6270
        w = NP.exp(-0.5 * (psf / sigma[:, :, NP.newaxis]) ** 2)          # :6283
    return w


def skyvis(baselines, channels, dircos, pbfluxes, pc_dircos, fwhm_deg=None,
           gradient=False, memsave=False, slab_bytes=256 * 2 ** 20):
    """V[b,f] = sum_s pbfluxes[s,f] * w[s,b,f] * exp(-i 2 pi (tau[s,b]-taupc[b]) f[f]).

    baselines (nbl,3) ENU metres; channels (nchan,) Hz; dircos (nsrc,3) ENU unit
    vectors of the sources in the region of interest; pbfluxes (nsrc,nchan) =
    beam x flux (interferometry.py:6254); pc_dircos (3,) phase-centre direction
    (:6164); fwhm_deg (nsrc,) or None (taper only when the sky model carries
    src_shape, :6258).

    memsave=False -> fp64 path (:6332-6343), complex128 result.
    memsave=True  -> the reference's single-precision path (:6323-6330): tau, taupc,
                     channels, 2 pi, pbfluxes, w all cast to float32 BEFORE the phase
                     is formed; complex64 result.
    Returns skyvis (nbl,nchan) and, if gradient, skyvis_gradient (3,nbl,nchan).
    """
    baselines = NP.asarray(baselines, dtype=NP.float64).reshape(-1, 3)
    channels = NP.asarray(channels, dtype=NP.float64).ravel()
    dircos = NP.asarray(dircos, dtype=NP.float64).reshape(-1, 3)
    pbfluxes = NP.asarray(pbfluxes)
    nbl, nchan, nsrc = baselines.shape[0], channels.size, dircos.shape[0]
    ftype, ctype = (NP.float32, NP.complex64) if memsave else (NP.float64, NP.complex128)
    out = NP.zeros((nbl, nchan), dtype=ctype)                             # :6186
    grad = NP.zeros((3, nbl, nchan), dtype=ctype) if gradient else None   # :6313
    if nsrc == 0:                                                         # :6378-6382
        return (out, grad) if gradient else out
    pc_delay = geometric_delay(baselines, NP.asarray(pc_dircos, dtype=NP.float64).reshape(1, 3))  # :6165 (1,nbl)
    tau = geometric_delay(baselines, dircos)                              # :6255 (nsrc,nbl)
    bytes_per_src = nbl * nchan * (8 if memsave else 16) * (4 if gradient else 1) * 3
    step = max(1, int(slab_bytes // max(bytes_per_src, 1)))
    for s0 in range(0, nsrc, step):                                       # :6355 / :6367
        s1 = min(s0 + step, nsrc)
        t = tau[s0:s1]
        w = None
        if fwhm_deg is not None:
            w = taper_weights(baselines, t, channels, NP.asarray(fwhm_deg)[s0:s1])
        if memsave:
            pm = NP.exp(-1j * NP.asarray(2.0 * NP.pi).astype(NP.float32)
                        * (t[:, :, NP.newaxis].astype(NP.float32) - pc_delay.astype(NP.float32).reshape(1, -1, 1))
                        * channels.astype(NP.float32).reshape(1, 1, -1)).astype(NP.complex64)    # :6323 / :6356
            if w is not None:
                pm *= w.astype(NP.float32)                                # :6326 / :6358
            pm *= pbfluxes[s0:s1, NP.newaxis, :].astype(NP.float32)       # :6327 / :6361
        else:
            pm = NP.exp(-1j * NP.asarray(2.0 * NP.pi).astype(NP.float64)
                        * (t[:, :, NP.newaxis] - pc_delay.reshape(1, -1, 1))
                        * channels.reshape(1, 1, -1)).astype(NP.complex128)                        # :6332 / :6368
            if w is not None:
                pm *= w                                                   # :6335 / :6370
            pm *= pbfluxes[s0:s1, NP.newaxis, :].astype(NP.float64)       # :6340 / :6372
        out += NP.sum(pm, axis=0)                                         # :6362 / :6373
        if gradient:
            grad += NP.sum(dircos[s0:s1, :, NP.newaxis, NP.newaxis].astype(ftype) * pm[:, NP.newaxis, :, :], axis=0)  # :6365 / :6376
    return (out, grad) if gradient else out


def abs_flux_sum(pbfluxes):
    """sum_s |pbfluxes[s,f]| -- the per-channel scale the parity tolerances are quoted against
    (fp64 max|dV| <= 1e-11 * this, fp32 <= 5e-6 * this)."""
    return NP.sum(NP.abs(NP.asarray(pbfluxes, dtype=NP.float64)), axis=0)
