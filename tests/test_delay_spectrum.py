"""Delay power-spectrum stage: prisim_amd.delay_spectrum against oracle/delay_ps_oracle.py, which restates
prisim/delay_spectrum.py:389, 395-489, 1303-1331, 3640-3663, 3707, 3775, 3992-3993.  PARITY UNPINNED for the cosmology (astropy in the
reference; a stated flat LambdaCDM here, two independent quadratures + the Einstein-de Sitter closed form) and for the DSP.* transform
(tests/test_oracle_kats.py KAT-8).  The CPU tests drive the product's classes through the OracleContext seam; the `gpu` test runs the whole
chain on the device at config-2 size."""
import os
import sys

import numpy as NP
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from oracle import delay_ps_oracle as DPO, delay_oracle as DO, skyvis_oracle as O, beams_oracle as BO
from prisim_amd import _abi, delay_spectrum as DS, geometry as GEOM, workloads as W
from prisim_amd import skymodel as SM

C = 299792458.0


def test_cosmology_two_quadratures_and_closed_form():
    z = DS.REST_FREQ_HI / 150e6 - 1
    a = DS.cosmo100.comoving_distance(z).to('Mpc').value
    b = DPO.comoving_distance(z)
    assert abs(a - b) <= 1e-9 * a and 6000.0 < a < 7500.0                        # Mpc/h to z = 8.47
    assert abs(float(DS.cosmo100.efunc(z)) - float(DPO.efunc(z))) <= 1e-12 * float(DPO.efunc(z))
    eds = DS.FlatLambdaCDM(H0=100.0, Om0=1.0, Tcmb0=0.0)                          # Einstein-de Sitter: D = 2c/H0 (1 - 1/sqrt(1+z))
    for zz in (0.5, 3.0, 8.47):
        want = 2 * C / 1e3 / 100.0 * (1 - 1 / NP.sqrt(1 + zz))
        assert abs(eds.comoving_distance(zz).value - want) <= 1e-10 * want
        assert abs(DPO.comoving_distance(zz, Om0=1.0, Tcmb0=0.0) - want) <= 1e-9 * want
    zs = NP.array([0.0, 1.0, 7.0])
    assert DS.cosmo100.comoving_distance(zs).value.shape == (3,) and DS.cosmo100.comoving_distance(zs).value[0] == 0.0
    k = DS.dkprll_deta(z)
    assert abs(k - DPO.power_factor(150e6 + 1e5 * (NP.arange(8) - 4), {'shape': 'delta'})['dkprll_deta']) <= 1e-12 * k
    with pytest.raises(ValueError):
        DS.dkprll_deta(-0.1)
    with pytest.raises(TypeError):
        DS.dkprll_deta('1')
    with pytest.raises(TypeError):
        DS.dkprll_deta(1.0, cosmo=object())


def test_beam3dvol_matches_the_literal_restatement_and_rejects_what_the_reference_rejects():
    f = 150e6 + 2e5 * (NP.arange(24) - 12)
    th, ph = DPO.healpix_ring_angles(16)
    th2, ph2 = GEOM.healpix_pix2ang_ring(16)
    assert NP.max(NP.abs(th - th2)) <= 1e-14 and NP.max(NP.abs(ph - ph2)) <= 1e-14
    beam = DPO.telescope_beam_on_healpix({'id': 'hera'}, f, nside=16)
    wts = NP.stack((NP.blackman(f.size) + 0.1, NP.ones(f.size)))
    for w in (None, wts[0], wts):
        a, b = DS.beam3Dvol(beam, f, freq_wts=w), DPO.beam3Dvol(beam, f, freq_wts=w)
        assert a.shape == b.shape and NP.max(NP.abs(a - b)) <= 1e-13 * NP.max(b)
    ach = DS.beam3Dvol(beam[:, 12], f)                                            # (npix,) -> the same beam at every channel
    assert abs(ach[0] - DPO.beam3Dvol(NP.repeat(beam[:, [12]], f.size, axis=1), f)[0]) <= 1e-13 * ach[0]
    with pytest.raises(TypeError):
        DS.beam3Dvol(beam.tolist(), f)
    with pytest.raises(ValueError):
        DS.beam3Dvol(beam, f[:1])
    with pytest.raises(ValueError):
        DS.beam3Dvol(beam[:, :5], f)
    with pytest.raises(ValueError):
        DS.beam3Dvol(2.0 * beam, f)                                               # not peak-normalised
    with pytest.raises(ValueError):
        DS.beam3Dvol(beam[:100], f)                                               # not a HEALPix pixel count


def _observed_array(monkeypatch, nbl=7, nchan=32, nt=3, reserve=False, ctxcls=None, seed=5):
    from prisim_amd import interferometry as RI
    if ctxcls is not None:
        monkeypatch.setattr(_abi, 'Context', ctxcls)
    rng = NP.random.default_rng(seed)
    ch = 150e6 + 2e5 * (NP.arange(nchan) - nchan // 2)
    bl = rng.uniform(-80.0, 80.0, size=(nbl, 3)) * NP.array([1.0, 1.0, 0.02])
    alt, az = rng.uniform(20.0, 89.0, 50), rng.uniform(0.0, 360.0, 50)
    skymod = SM.SkyModel(location=NP.stack((alt, az), axis=1), flux_ref=rng.uniform(0.5, 5.0, 50), spindex=rng.uniform(-1.0, 0.0, 50), ref_freq=150e6)
    ia = RI.InterferometerArray(['b%d' % i for i in range(nbl)], bl, ch, telescope={'id': 'hera'}, latitude=-30.7, skycoords='altaz',
                                pointing_coords='hadec')
    if reserve:
        ia.reserve(nt)
    bpass = 0.5 + 0.5 * NP.hanning(nchan + 2)[1:-1]
    for j in range(nt):
        ia.observe((2457000.5 + j, 10.0 + j), {'Tnet': 100.0}, bpass, [0.0, -30.7], skymod, 10.0)
    return ia, bpass


def test_delay_spectrum_classes_on_the_oracle_seam(monkeypatch):
    import fake_context
    ia, bpass = _observed_array(monkeypatch, ctxcls=fake_context.OracleContext)
    nbl, nchan, nt = ia.baselines.shape[0], ia.channels.size, ia.n_acc
    ds = DS.DelaySpectrum(ia)
    assert ds.f is ia.channels and ds.df == ia.freq_resolution and ds.n_acc == nt and ds.pad == 0.0
    assert ds.horizon_delay_limits.shape == (nt, nbl, 2)
    blen = NP.sqrt(NP.sum(ia.baselines ** 2, axis=1))
    assert NP.allclose(ds.horizon_delay_limits[0, :, 1] - ds.horizon_delay_limits[0, :, 0], 2 * blen / C, rtol=1e-14)
    assert NP.array_equal(ds.lags, NP.fft.fftshift(NP.fft.fftfreq(nchan, ds.df)))
    with pytest.raises(TypeError):
        DS.DelaySpectrum(object())
    with pytest.raises(TypeError):
        ds.delay_transform(pad='1')
    with pytest.raises(TypeError):
        ds.delay_transform(downsample=1, verbose=False)
    with pytest.raises(ValueError):
        ds.delay_transform(freq_wts=NP.ones(nchan + 1), verbose=False)

    win = NP.blackman(nchan) + 0.05
    vis = NP.asarray(ia.skyvis_freq)
    bp = NP.broadcast_to(bpass.reshape(1, -1, 1), vis.shape)
    for fw, wfull in ((win, NP.broadcast_to(win.reshape(1, -1, 1), vis.shape)),
                      (NP.outer(NP.linspace(1.0, 2.0, nbl), win), NP.broadcast_to(NP.outer(NP.linspace(1.0, 2.0, nbl), win)[:, :, None], vis.shape)),
                      (NP.outer(win, NP.linspace(1.0, 1.5, nt)), NP.broadcast_to(NP.outer(win, NP.linspace(1.0, 1.5, nt))[None, :, :], vis.shape))):
        for pad in (1.0, 0.0):
            res = ds.delay_transform(pad=pad, freq_wts=fw, verbose=False)
            ref, _ = DO.delay_transform(vis, bp, wfull, ds.df, pad=pad)
            assert res['pad'] == pad and res['skyvis_lag'].shape == ref.shape
            assert NP.max(NP.abs(res['skyvis_lag'] - ref)) <= 1e-12 * NP.max(NP.abs(ref))
            assert NP.array_equal(NP.asarray(res['freq_wts']), wfull)
            assert res['lags'].shape == (nchan,) and res['vis_lag'] is None
            kern, _ = DO.delay_transform(NP.ones_like(vis), bp, wfull, ds.df, pad=pad)
            assert NP.max(NP.abs(res['lag_kernel'] - kern)) <= 1e-12 * NP.max(NP.abs(kern))
    assert ds.skyvis_lag is None and ds.pad == 0.0                                # action=None stores nothing (:1333)
    res = ds.delay_transform(pad=1.0, freq_wts=win, action='store', verbose=False)
    assert ds.pad == 1.0 and NP.array_equal(ds.lags, res['lags']) and NP.array_equal(NP.asarray(ds.bp_wts)[2, :, 1], win)
    wfull = NP.broadcast_to(win.reshape(1, -1, 1), vis.shape)
    ref, reflags = DO.delay_transform(vis, bp, wfull, ds.df, pad=1.0)
    assert NP.max(NP.abs(ds.skyvis_lag - ref)) <= 1e-12 * NP.max(NP.abs(ref)) and NP.allclose(ds.lags, reflags, rtol=0, atol=1e-18)

    dps = DS.DelayPowerSpectrum(ds)
    pf = DPO.power_factor(ia.channels, {'id': 'hera'}, bp_wts_row=win)
    for name in ('z', 'bw', 'drz_los', 'rz_los', 'Jy2K'):
        assert abs(getattr(dps, name) - pf[name]) <= 1e-9 * abs(pf[name]), name
    assert abs(dps.jacobian1[0] - pf['jacobian1'][0]) <= 1e-9 * pf['jacobian1'][0]       # (the seam evaluates the oracle's Airy beam)
    assert abs(dps.jacobian2 - pf['jacobian2']) <= 1e-9 * pf['jacobian2']
    assert abs(dps.K2Jy * dps.Jy2K - 1.0) <= 1e-15 and dps.rz_transverse == dps.rz_los
    assert NP.allclose(dps.kprll, pf['dkprll_deta'] * ds.lags, rtol=1e-12)
    assert NP.allclose(dps.kperp, 2 * NP.pi * blen / dps.wl0 / pf['rz_los'], rtol=1e-9)
    assert dps.horizon_kprll_limits.shape == ds.horizon_delay_limits.shape and dps.dps['skyvis'] is None
    dps.compute_power_spectrum()
    want, _, _ = DPO.delay_power_spectrum(vis, bp, wfull, ia.channels, {'id': 'hera'}, pad=1.0)
    assert NP.max(NP.abs(dps.dps['skyvis'] - want)) <= 1e-9 * NP.max(want)
    assert 'vis' not in dps.dps and abs(dps.power_scale() - pf['factor'][0]) <= 1e-9 * pf['factor'][0]
    with pytest.raises(TypeError):
        DS.DelayPowerSpectrum(ia)
    with pytest.raises(TypeError):
        DS.DelayPowerSpectrum(ds, cosmo=1.0)


def test_resident_cube_keeps_spectra_and_power_on_the_device_side_of_the_seam(monkeypatch):
    """With reserve() the cube is resident: delay_transform(action='store') must go through delay_transform_device once (no per-snapshot
    host transforms) and compute_power_spectrum() must hand the factor to the device (power_scale), fetching on first read."""
    import fake_context
    calls = {'device': 0, 'host': 0, 'scale': None}

    class Spy(fake_context.OracleContext):
        def delay_transform_device(self, nt, **kw):
            calls['device'] += 1
            if kw.get('want_power'):
                calls['scale'] = kw.get('power_scale')
            return fake_context.OracleContext.delay_transform_device(self, nt, **kw)

        def delay_transform_host(self, vis, bpwts, pad):
            calls['host'] += 1
            return fake_context.OracleContext.delay_transform_host(self, vis, bpwts, pad)

    ia, bpass = _observed_array(monkeypatch, reserve=True, ctxcls=Spy)
    ds = DS.DelaySpectrum(ia)
    win = NP.blackman(ia.channels.size) + 0.05
    ds.delay_transform(pad=1.0, freq_wts=win, action='store', verbose=False)
    assert calls == {'device': 1, 'host': 0, 'scale': None}
    dps = DS.DelayPowerSpectrum(ds)
    dps.compute_power_spectrum()
    assert calls['device'] == 1                                                   # nothing computed or fetched until it is read
    got = dps.dps['skyvis']
    assert calls['device'] == 2 and calls['host'] == 0 and abs(calls['scale'] - dps.power_scale()) == 0.0
    vis = NP.asarray(ia.skyvis_freq)
    want, _, _ = DPO.delay_power_spectrum(vis, NP.broadcast_to(bpass.reshape(1, -1, 1), vis.shape), NP.broadcast_to(win.reshape(1, -1, 1), vis.shape),
                                          ia.channels, {'id': 'hera'}, pad=1.0)
    assert got.shape == want.shape and NP.max(NP.abs(got - want)) <= 1e-9 * NP.max(want)
    assert NP.max(NP.abs(NP.abs(ds.skyvis_lag) ** 2 * dps.power_scale() - want)) <= 1e-9 * NP.max(want)


def test_stored_spectra_survive_later_transforms_on_the_same_context(monkeypatch):
    """delay_transform(action='store') keeps skyvis_lag lazy in the context's single resident buffer; a later transform on
    the same context (another window without 'store', the array's own delay_transform, a power-spectrum fetch) overwrites that buffer.
    The stored result must not change (the reference stores a host copy): the transform is run again before the read."""
    import fake_context
    ia, bpass = _observed_array(monkeypatch, reserve=True, ctxcls=fake_context.OracleContext)
    w1 = NP.blackman(ia.channels.size) + 0.05
    w2 = NP.hanning(ia.channels.size) + 0.5
    ref = DS.DelaySpectrum(ia)
    want = NP.array(ref.delay_transform(pad=1.0, freq_wts=w1, action=None, verbose=False)['skyvis_lag'])
    ds = DS.DelaySpectrum(ia)
    ds.delay_transform(pad=1.0, freq_wts=w1, action='store', verbose=False)
    other = ds.delay_transform(pad=1.0, freq_wts=w2, action=None, verbose=False)['skyvis_lag']       # not stored: overwrites the buffer
    assert NP.max(NP.abs(other - want)) > 1e-3 * NP.max(NP.abs(want))
    ia.delay_transform(pad=1.0, freq_wts=w2, verbose=False)                                          # ... and so does the array's own
    got = ds.skyvis_lag
    assert NP.max(NP.abs(got - want)) <= 1e-12 * NP.max(NP.abs(want))
    # the array's own lazily fetched spectra are as safe against the DelaySpectrum's transforms
    ds2 = DS.DelaySpectrum(ia)
    ds2.delay_transform(pad=1.0, freq_wts=w1, action='store', verbose=False)
    assert NP.max(NP.abs(ia.skyvis_lag - other)) <= 1e-12 * NP.max(NP.abs(other))


@pytest.mark.gpu
def test_delay_power_spectrum_on_the_gpu_at_config2_size():
    """BASELINE config 2 (HERA-19, 256 channels, nside-16 diffuse, Airy 14 m) through observe() -> DelaySpectrum.delay_transform ->
    DelayPowerSpectrum.compute_power_spectrum on the device, against the oracle chain: visibilities (C oracle), delay transform and
    abs^2 * jacobian1 * jacobian2 * Jy2K^2 with the beam volume from the oracle's own beam on its own HEALPix grid."""
    from oracle import c_oracle as CO
    from prisim_amd import interferometry as RI
    cfg = W.config2()
    bl, ch, sky = cfg['baselines'], cfg['channels'], cfg['sky']
    skymod = SM.SkyModel(location=sky['altaz'], flux_ref=sky['flux_ref'], spindex=sky['spindex'], ref_freq=sky['ref_freq'],
                         src_shape=NP.stack((sky['fwhm_deg'], sky['fwhm_deg'], NP.zeros_like(sky['fwhm_deg'])), axis=1))
    pb = BO.airy_disk_pattern(14.0, sky['altaz'], ch) * sky['flux_ref'][:, None] * (ch[None, :] / sky['ref_freq']) ** sky['spindex'][:, None]
    zen = NP.array([0.0, 0.0, 1.0])
    ref_vis = CO.skyvis(bl, ch, sky['dircos'], pb, zen, fwhm_deg=sky['fwhm_deg'])
    win = NP.blackman(ch.size) + 0.01
    for reserve in (True, False):
        ia = RI.InterferometerArray(['b%d' % i for i in range(bl.shape[0])], bl, ch, telescope={'id': 'hera'}, latitude=-30.7224,
                                    skycoords='altaz', pointing_coords='altaz')
        if reserve:
            ia.reserve(2)
        for j in range(2):
            ia.observe((2457000.5 + j, 30.0), {'Tnet': 100.0}, NP.ones(ch.size), [90.0, 270.0], skymod, 10.7)
        ds = DS.DelaySpectrum(ia)
        res = ds.delay_transform(pad=1.0, freq_wts=win, action='store', verbose=False)
        dps = DS.DelayPowerSpectrum(ds)
        dps.compute_power_spectrum()
        vis2 = NP.repeat(ref_vis[:, :, None], 2, axis=2)
        ones = NP.ones_like(vis2, dtype=float)
        want, lags, pf = DPO.delay_power_spectrum(vis2, ones, NP.broadcast_to(win.reshape(1, -1, 1), vis2.shape), ch, {'id': 'hera'}, pad=1.0)
        assert abs(dps.jacobian1[0] - pf['jacobian1'][0]) <= 1e-9 * pf['jacobian1'][0]          # beam volume: device beam vs oracle beam
        got = dps.dps['skyvis']
        assert got.shape == want.shape == (bl.shape[0], ch.size, 2)
        assert NP.max(NP.abs(got - want)) <= 1e-9 * NP.max(want), reserve
        assert NP.allclose(res['lags'], lags, rtol=0, atol=1e-18)
        full = ds.delay_transform(pad=1.0, freq_wts=win, downsample=False, verbose=False)          # no decimation: 2 N lags
        assert full['skyvis_lag'].shape == (bl.shape[0], 2 * ch.size, 2) and full['lags'].size == 2 * ch.size
        assert NP.max(NP.abs(full['skyvis_lag'][:, ::2, :] - ds.skyvis_lag)) <= 1e-10 * NP.max(NP.abs(ds.skyvis_lag))
