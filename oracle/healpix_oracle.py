"""numpy restatement of the external-beam interpolation (TEST INFRASTRUCTURE).

Reference call site: scripts/run_prisim.py:2091-2103 (and :1897-1908):
    interp_logbeam = OPS.healpix_interp_along_axis(log10(external_beam), theta_phi, inloc_axis=beam_freqs,
                                                   outloc_axis=chans, axis=1, kind=spec_interp)
    interp_logbeam -= max(nanmax(interp_logbeam, axis=0), 0)        (:2098-2101)
    pbeam = 10**interp_logbeam                                      (:2102)       stored float32 (interferometry.py:4466)
OPS.healpix_interp_along_axis (astroutils, un-vendored: PARITY UNPINNED) = scipy interp1d along the frequency axis
followed by healpy.get_interp_val (bilinear over the 4 nearest RING pixels).  healpy is not available either; the
bilinear weights below restate HEALPix' published Healpix_Base::get_interpol algorithm (Gorski et al. 2005;
healpix_base.cc), checked by known answers in tests/test_oracle_kats.py.
"""
import numpy as NP
from scipy.interpolate import interp1d


def _ring_info(nside, ir):
    """startpix, ringpix, theta, shift(0/1) of ring ir (1 .. 4 nside - 1); vectorised."""
    ir = NP.asarray(ir, dtype=NP.int64)
    npix = 12 * nside * nside
    ncap = 2 * nside * (nside - 1)
    fact2 = 4.0 / npix
    fact1 = (2 * nside) * fact2
    north = NP.where(ir > 2 * nside, 4 * nside - ir, ir)          # mirrored ring index
    cap = north < nside
    tmp = north.astype(float) ** 2 * fact2
    theta_cap = NP.arctan2(NP.sqrt(tmp * (2 - tmp)), 1 - tmp)
    theta_eq = NP.arccos(NP.clip((2 * nside - north) * fact1, -1, 1))
    theta = NP.where(cap, theta_cap, theta_eq)
    ringpix = NP.where(cap, 4 * north, 4 * nside)
    start = NP.where(cap, 2 * north * (north - 1), ncap + (north - nside) * 4 * nside)
    shift = NP.where(cap, 1, ((north - nside) & 1) == 0).astype(NP.int64)
    south = ir > 2 * nside
    theta = NP.where(south, NP.pi - theta, theta)
    start = NP.where(south, npix - start - ringpix, start)
    return start, ringpix, theta, shift


def _ring_above(nside, z):
    az = NP.abs(z)
    eq = (nside * (2 - 1.5 * z)).astype(NP.int64)
    ir = (nside * NP.sqrt(3 * (1 - az))).astype(NP.int64)
    return NP.where(az <= 2.0 / 3.0, eq, NP.where(z > 0, ir, 4 * nside - ir - 1))


def get_interp_weights(nside, theta, phi):
    """(pix [4,n], wgt [4,n]) of the bilinear interpolation in the RING scheme (Healpix_Base::get_interpol)."""
    theta = NP.asarray(theta, dtype=NP.float64).ravel()
    phi = NP.mod(NP.asarray(phi, dtype=NP.float64).ravel(), 2 * NP.pi)
    phi = NP.where(phi >= 2 * NP.pi, phi - 2 * NP.pi, phi)      # (mod of -1e-17 rounds to 2 pi itself: that direction is phi = 0)
    n = theta.size
    npix = 12 * nside * nside
    z = NP.cos(theta)
    ir1 = _ring_above(nside, z)
    ir2 = ir1 + 1
    pix = NP.zeros((4, n), dtype=NP.int64)
    wgt = NP.zeros((4, n), dtype=NP.float64)
    theta1 = NP.zeros(n)
    theta2 = NP.zeros(n)
    for k, (ir, ok) in enumerate(((ir1, ir1 > 0), (ir2, ir2 < 4 * nside))):
        irc = NP.clip(ir, 1, 4 * nside - 1)
        sp, nr, th, shift = _ring_info(nside, irc)
        dphi = 2 * NP.pi / nr
        tmp = phi / dphi - 0.5 * shift
        i1 = NP.where(tmp < 0, tmp.astype(NP.int64) - 1, tmp.astype(NP.int64))
        w1 = (phi - (i1 + 0.5 * shift) * dphi) / dphi
        i2 = i1 + 1
        i1 = NP.where(i1 < 0, i1 + nr, i1)
        i2 = NP.where(i2 >= nr, i2 - nr, i2)
        pix[2 * k] = NP.where(ok, sp + i1, 0)
        pix[2 * k + 1] = NP.where(ok, sp + i2, 0)
        wgt[2 * k] = NP.where(ok, 1 - w1, 0)
        wgt[2 * k + 1] = NP.where(ok, w1, 0)
        if k == 0:
            theta1 = th
        else:
            theta2 = th
    npole = ir1 == 0
    spole = ir2 == 4 * nside
    mid = ~(npole | spole)
    with NP.errstate(divide='ignore', invalid='ignore'):
        wth = NP.where(mid, (theta - theta1) / (theta2 - theta1), 0.0)
    wgt[0] = NP.where(mid, wgt[0] * (1 - wth), wgt[0]); wgt[1] = NP.where(mid, wgt[1] * (1 - wth), wgt[1])
    wgt[2] = NP.where(mid, wgt[2] * wth, wgt[2]); wgt[3] = NP.where(mid, wgt[3] * wth, wgt[3])
    # north pole
    with NP.errstate(divide='ignore', invalid='ignore'):
        wtn = NP.where(npole, theta / theta2, 0.0)
    fac = (1 - wtn) * 0.25
    wgt[2] = NP.where(npole, wgt[2] * wtn + fac, wgt[2]); wgt[3] = NP.where(npole, wgt[3] * wtn + fac, wgt[3])
    wgt[0] = NP.where(npole, fac, wgt[0]); wgt[1] = NP.where(npole, fac, wgt[1])
    pix[0] = NP.where(npole, (pix[2] + 2) & 3, pix[0]); pix[1] = NP.where(npole, (pix[3] + 2) & 3, pix[1])
    # south pole
    with NP.errstate(divide='ignore', invalid='ignore'):
        wts = NP.where(spole, (theta - theta1) / (NP.pi - theta1), 0.0)
    fac = wts * 0.25
    wgt[0] = NP.where(spole, wgt[0] * (1 - wts) + fac, wgt[0]); wgt[1] = NP.where(spole, wgt[1] * (1 - wts) + fac, wgt[1])
    wgt[2] = NP.where(spole, fac, wgt[2]); wgt[3] = NP.where(spole, fac, wgt[3])
    pix[2] = NP.where(spole, ((pix[0] + 2) & 3) + npix - 4, pix[2]); pix[3] = NP.where(spole, ((pix[1] + 2) & 3) + npix - 4, pix[3])
    return pix, wgt


def get_interp_val(hmap, theta, phi):
    """healpy.get_interp_val for a RING map [npix] or [npix, ncol]."""
    hmap = NP.asarray(hmap)
    nside = int(round(NP.sqrt(hmap.shape[0] / 12.0)))
    pix, wgt = get_interp_weights(nside, theta, phi)
    if hmap.ndim == 1:
        return NP.sum(hmap[pix] * wgt, axis=0)
    return NP.sum(hmap[pix] * wgt[:, :, None], axis=0)


def spectral_interp_matrix(in_freqs, out_freqs, kind='cubic'):
    """(nout, nin) matrix M with interp1d(in_freqs, y, kind)(out_freqs) == M @ y for every y (the interpolation is linear
    in the data).  Built by interpolating unit vectors."""
    in_freqs = NP.asarray(in_freqs, dtype=float)
    eye = NP.eye(in_freqs.size)
    return interp1d(in_freqs, eye, kind=kind, axis=0, bounds_error=False, fill_value='extrapolate', assume_sorted=True)(out_freqs)


def external_beam(beam, beam_freqs, theta, phi, chans, kind='cubic', chromatic=True, select_freq=None, quantise_f32=True):
    """pbeam (nsrc, nchan) from an external HEALPix beam [npix, nfreq] in the local (zenith angle, azimuth) frame."""
    beam = NP.asarray(beam, dtype=float)
    chans = NP.asarray(chans, dtype=float)
    with NP.errstate(divide='ignore'):
        logbeam = NP.log10(beam)
    if chromatic:
        logchan = interp1d(NP.asarray(beam_freqs, dtype=float), logbeam, kind=kind, axis=1, bounds_error=False,
                           fill_value='extrapolate', assume_sorted=True)(chans)                       # run_prisim.py:2094
    else:
        j = int(NP.argmin(NP.abs(NP.asarray(beam_freqs) - select_freq)))                               # :2096
        logchan = NP.repeat(logbeam[:, [j]], chans.size, axis=1)                                       # :2097
    interp_logbeam = get_interp_val(logchan, theta, phi)
    return normalise_logbeam(interp_logbeam, quantise_f32=quantise_f32)


def normalise_logbeam(interp_logbeam, quantise_f32=True):
    """scripts/run_prisim.py:2099-2103 (per-channel maximum over the sources, clamped at 0, subtracted; 10 ** (.)) and the float32
    storage of a supplied beam (interferometry.py:4466).  Pinned by tests/golden/golden_aux.npz (the reference's statements executed)."""
    interp_logbeam = NP.asarray(interp_logbeam, dtype=NP.float64)
    mx = NP.nanmax(interp_logbeam, axis=0)                                                             # :2099
    mx = NP.where(mx <= 0.0, 0.0, mx).reshape(1, -1)                                                   # :2100-2101
    pb = 10 ** (interp_logbeam - mx)                                                                   # :2102-2103
    return pb.astype(NP.float32).astype(NP.float64) if quantise_f32 else pb                            # interferometry.py:4466
