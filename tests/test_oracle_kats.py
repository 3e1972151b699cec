"""CPU: analytic known-answer tests of the oracle (SURVEY.md 8(c) KAT-1 .. KAT-8)."""
import numpy as NP

from oracle import skyvis_oracle as O, beams_oracle as BO, delay_oracle as DO

C = 299792458.0
BL = NP.array([[14.6, 0.0, 0.0], [7.3, 12.644, 0.0], [-250.0, 120.0, 1.5], [0.0, 0.0, 0.0]])
CH = 150e6 + (NP.arange(32) - 16) * 390625.0


def test_kat1_source_at_phase_centre():
    pc = O.altaz2dircos([[70.0, 123.0]])[0]
    p = NP.linspace(1.0, 2.0, CH.size)[None, :]
    v = O.skyvis(BL, CH, pc[None, :], p, pc)
    assert NP.max(NP.abs(v - p)) <= 1e-12


def test_kat2_single_source_closed_form():
    pc = NP.array([0.0, 0.0, 1.0])
    s = O.altaz2dircos([[40.0, 250.0]])[0]
    p = NP.full((1, CH.size), 3.0)
    v = O.skyvis(BL, CH, s[None, :], p, pc)
    expected = 3.0 * NP.exp(-2j * NP.pi * CH[None, :] * (BL @ (s - pc))[:, None] / C)
    assert NP.max(NP.abs(v - expected)) <= 1e-10


def test_kat3_two_equal_sources_envelope():
    pc = NP.array([0.0, 0.0, 1.0])
    s = O.altaz2dircos([[60.0, 10.0], [75.0, 200.0]])
    p = NP.full((2, CH.size), 2.0)
    v = O.skyvis(BL, CH, s, p, pc)
    dphi = NP.pi * CH[None, :] * (BL @ (s[0] - s[1]))[:, None] / C
    assert NP.max(NP.abs(NP.abs(v) - NP.abs(4.0 * NP.cos(dphi)))) <= 1e-10


def test_kat4_hermitian():
    rng = NP.random.default_rng(4)
    s = O.altaz2dircos(NP.stack((rng.uniform(10, 90, 20), rng.uniform(0, 360, 20)), 1))
    p = rng.uniform(0, 5, (20, CH.size))
    pc = O.altaz2dircos([[85.0, 0.0]])[0]
    assert NP.max(NP.abs(O.skyvis(-BL, CH, s, p, pc) - NP.conj(O.skyvis(BL, CH, s, p, pc)))) <= 1e-12


def test_kat5_linearity_and_additivity():
    rng = NP.random.default_rng(5)
    s = O.altaz2dircos(NP.stack((rng.uniform(10, 90, 30), rng.uniform(0, 360, 30)), 1))
    p = rng.uniform(0, 5, (30, CH.size))
    pc = NP.array([0.0, 0.0, 1.0])
    full = O.skyvis(BL, CH, s, p, pc)
    assert NP.max(NP.abs(O.skyvis(BL, CH, s, 2.5 * p, pc) - 2.5 * full)) <= 1e-11
    parts = O.skyvis(BL, CH, s[:11], p[:11], pc) + O.skyvis(BL, CH, s[11:], p[11:], pc)
    assert NP.max(NP.abs(parts - full)) <= 1e-11


def test_kat6_taper_limits():
    pc = NP.array([0.0, 0.0, 1.0])
    s = O.altaz2dircos([[50.0, 90.0]])
    p = NP.ones((1, CH.size))
    # FWHM = 0  ->  w = 1
    assert NP.max(NP.abs(O.skyvis(BL, CH, s, p, pc, fwhm_deg=[0.0]) - O.skyvis(BL, CH, s, p, pc))) == 0.0
    # baseline parallel to the source direction -> u_perp = 0 -> w = 1
    blpar = 100.0 * s
    w = O.taper_weights(blpar, O.geometric_delay(blpar, s), CH, [1.0])
    assert NP.max(NP.abs(w - 1.0)) <= 1e-9
    # general: w = exp(-ln2 FWHMdc^2 (|b|^2-(b.s)^2) f^2/c^2)
    w = O.taper_weights(BL, O.geometric_delay(BL, s), CH, [0.7])
    fd = 2 * NP.sin(0.5 * NP.radians(0.7))
    perp2 = NP.sum(BL ** 2, 1) - (BL @ s[0]) ** 2
    expected = NP.exp(-NP.log(2.0) * fd ** 2 * perp2[None, :, None] * CH[None, None, :] ** 2 / C ** 2)
    assert NP.max(NP.abs(w - expected)) <= 1e-12


def test_kat7_beams():
    f = NP.array([150e6])
    lam = C / f[0]
    # Gaussian: 1 at the pointing centre, half power where sin(theta) = sigma_dc sqrt(ln 2)
    sig_dc = 1.0 / (2 * NP.pi * (14.0 / (2 * NP.sqrt(2 * NP.log(2))) / lam))
    th = NP.arcsin(sig_dc * NP.sqrt(NP.log(2.0)))
    pb = BO.gaussian_beam(14.0, [[90.0, 0.0], [90.0 - NP.degrees(th), 33.0]], f)
    assert abs(pb[0, 0] - 1.0) <= 1e-15 and abs(pb[1, 0] - 0.5) <= 1e-12
    # Airy: peak 1, first null at sin(theta) = 1.21967 lambda / D
    th0 = NP.arcsin(1.2196698912665045 * lam / 14.0)
    pb = BO.airy_disk_pattern(14.0, [[90.0, 0.0], [90.0 - NP.degrees(th0), 120.0]], f)
    assert abs(pb[0, 0] - 1.0) <= 1e-12 and pb[1, 0] <= 1e-20


def test_kat8_delay_transform_tone():
    """A single-delay tone p exp(-2 pi i f tau0) transforms to a peak at lag +tau0 of height p N df."""
    nchan, df = 64, 1.0e5
    f = 150e6 + NP.arange(nchan) * df
    k0 = 5
    tau0 = k0 / (nchan * df)                  # exactly on the lag grid
    v = (2.0 * NP.exp(-2j * NP.pi * (f - f[0]) * tau0))[None, :, None]
    ones = NP.ones_like(v, dtype=float)
    for pad in (0.0, 1.0):
        lag, lags = DO.delay_transform(v, ones, ones, df, pad=pad)
        assert lag.shape == (1, nchan, 1)
        ipk = int(NP.argmax(NP.abs(lag[0, :, 0])))
        assert abs(lags[ipk] - tau0) <= 1e-12
        assert abs(abs(lag[0, ipk, 0]) - 2.0 * nchan * df) <= 1e-6 * nchan * df
    # lags are fftshift(fftfreq)
    assert NP.allclose(lags, NP.fft.fftshift(NP.fft.fftfreq(nchan, df)))


def test_kat9_healpix_bilinear_interpolation():
    """healpy.get_interp_val restatement: exact at pixel centres, partition of unity, second-order convergence, poles."""
    from oracle import healpix_oracle as H
    from prisim_amd import geometry as GEOM
    rng = NP.random.default_rng(9)
    t = NP.arccos(rng.uniform(-1, 1, 3000))
    p = rng.uniform(0, 2 * NP.pi, 3000)
    f = lambda th, ph: 1 + 0.5 * NP.cos(th) + 0.3 * NP.sin(th) * NP.cos(ph)
    errs = []
    for nside in (2, 4, 8, 16, 32):
        th, ph = GEOM.healpix_pix2ang_ring(nside)
        m = rng.normal(size=th.size)
        assert NP.max(NP.abs(H.get_interp_val(m, th, ph) - m)) <= 1e-10          # pixel centres reproduce the map
        pix, w = H.get_interp_weights(nside, t, p)
        assert NP.max(NP.abs(w.sum(0) - 1)) <= 1e-14 and w.min() >= -1e-15 and pix.min() >= 0 and pix.max() < th.size
        assert NP.max(NP.abs(H.get_interp_val(NP.full(th.size, 3.5), t, p) - 3.5)) <= 1e-14
        errs.append(NP.max(NP.abs(H.get_interp_val(f(th, ph), t, p) - f(t, p))))
    assert all(e2 < 0.6 * e1 for e1, e2 in zip(errs, errs[1:])) and errs[-1] < errs[0] / 40   # ~ h^2
    pix, w = H.get_interp_weights(4, [0.0, NP.pi], [0.3, 0.3])
    assert sorted(pix[:, 0]) == [0, 1, 2, 3] and sorted(pix[:, 1]) == [188, 189, 190, 191] and NP.allclose(w, 0.25)


def test_kat10_external_beam_normalisation_and_spectral_matrix():
    from oracle import healpix_oracle as H
    from prisim_amd import geometry as GEOM, primary_beams as PB
    nside = 8
    th, ph = GEOM.healpix_pix2ang_ring(nside)
    bf = NP.linspace(100e6, 200e6, 11)
    beam = (NP.cos(th / 2)[:, None] ** 4 + 1e-3) * (1 + 0.3 * (bf[None, :] / 150e6 - 1)) * 7.0      # peak 7 at the zenith pixel ring
    ch = 150e6 + (NP.arange(16) - 8) * 1e6
    srct = NP.radians([0.0, 10.0, 45.0, 80.0]); srcp = NP.radians([0.0, 33.0, 120.0, 300.0])
    pb = H.external_beam(beam, bf, srct, srcp, ch, kind='cubic')
    assert pb.shape == (4, 16) and NP.allclose(pb[0], 1.0, atol=1e-6)           # peak-normalised per channel over the sources
    assert NP.all(NP.diff(pb[:, 3]) < 0)                                        # falls with zenith angle
    m = PB.spectral_interp_matrix(bf, ch, kind='cubic')
    from scipy.interpolate import interp1d
    y = NP.log10(beam[5])
    assert NP.max(NP.abs(m @ y - interp1d(bf, y, kind='cubic')(ch))) <= 1e-13
    ma = PB.spectral_interp_matrix(bf, ch, chromatic=False, select_freq=151e6)
    assert NP.all(ma.sum(1) == 1) and NP.all(ma[:, 5] == 1)
    pba = H.external_beam(beam, bf, srct, srcp, ch, chromatic=False, select_freq=151e6)
    assert NP.allclose(pba, pba[:, [0]])                                        # achromatic: same in every channel
