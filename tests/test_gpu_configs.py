"""GPU (-m gpu): the BASELINE.json configurations at full array size through the reference's own entry point for the path
(InterferometerArray.observe / delay_transform), spot-checked against the C oracle.

  config 3 with its diffuse half   HERA-350 x 1024 ch x (1e4 point sources + nside=128 diffuse), taper on   run_prisim.py:1220-1246
  config 4                         MWA-128T x 768 ch, drift scan, external HEALPix beam                      run_prisim.py:2091-2103, 2165-2207
  config 5, two LSTs               HERA-350 x 1024 ch x nside=256 diffuse + delay transform                   interferometry.py:8052-8137
"""
import numpy as NP
import pytest

from oracle import c_oracle as CO, skyvis_oracle as O, beams_oracle as BO, delay_oracle as DO
from prisim_amd import workloads as W, geometry as GEOM
from prisim_amd import interferometry as RI, skymodel as SM

pytestmark = pytest.mark.gpu

SIDEREAL_DEG_PER_SEC = 360.0 * 1.00273790935 / 86400.0


def _radec_skymodel(sky, lat, lst0_deg):
    """The local-frame sky of a workload as a (RA, Dec) sky model that stands where the workload puts it at LST = lst0 (coordinates of
    date: epoch None, like the synthetic skies of prisim_amd/driver.py)."""
    hadec = GEOM.altaz2hadec(sky['altaz'], lat, units='degrees')
    radec = NP.stack(((lst0_deg - hadec[:, 0]) % 360.0, hadec[:, 1]), axis=1)
    n = radec.shape[0]
    return SM.SkyModel(location=radec, flux_ref=sky['flux_ref'], spindex=sky['spindex'], ref_freq=sky['ref_freq'],
                       src_shape=NP.stack((sky['fwhm_deg'], sky['fwhm_deg'], NP.zeros(n)), axis=1), epoch=None)


def _spot(bl, n=4):
    return NP.unique(NP.linspace(0, bl.shape[0] - 1, n).astype(int))


def _kappa(fwhm_deg):
    return float(NP.log(2.0) * (2.0 * NP.sin(0.5 * NP.radians(NP.max(fwhm_deg)))) ** 2)


def test_config4_drift_external_beam():
    """MWA-128T drift scan with the external HEALPix beam: 3 accumulations of config 4 at full array size, fp32 (memsave),
    every snapshot spot-checked on 4 baselines against the C oracle fed with the restated beam interpolation."""
    from oracle import healpix_oracle as H
    from conftest import body_class_sample
    cfg = W.config4(n_acc=3)
    bl, ch, sky, lat = cfg['baselines'], cfg['channels'], cfg['sky'], cfg['latitude']
    # every class of kernel body: lifting / non-lifting groups, the first groups whose leading sources are culled, the ragged last group
    sel, _ = body_class_sample(bl, ch, sky['dircos'], NP.array([0.0, 0.0, 1.0]), f32=True, kappa=_kappa(sky['fwhm_deg']))
    assert sel.size >= 12
    bl_run = NP.vstack((bl, -bl[sel]))                                   # the spot baselines once more, flipped
    lst0 = 40.0
    skymod = _radec_skymodel(sky, lat, lst0)
    labels = ['b%d' % i for i in range(bl_run.shape[0])]
    ia = RI.InterferometerArray(labels, bl_run, ch, telescope={'id': 'mwa'}, latitude=lat, skycoords='radec', pointing_coords='hadec')
    ia.reserve(cfg['n_acc'])
    ia.set_external_beam(cfg['beam_table'], cfg['beam_freqs'], spec_interp='cubic')
    zen = NP.array([0.0, 0.0, 1.0])
    for j in range(cfg['n_acc']):
        lst = lst0 + j * cfg['t_acc'] * SIDEREAL_DEG_PER_SEC
        ia.observe((2457000.5 + j * cfg['t_acc'] / 86400.0, lst), {'Tnet': 100.0}, NP.ones(ch.size), [0.0, lat], skymod, cfg['t_acc'],
                   memsave=True)
    assert ia.n_acc == 3 and all(isinstance(s, RI._DeviceSlot) for s in ia._cube)      # nothing was downloaded while observing
    # MWA baselines to 2.5 km over nside-64 pixels: observe() lists the sources by decreasing altitude and the library's taper
    # culling skips the zenith-most sources of the long-baseline groups (their weight underflows); the spot check below sums everything
    culled = ia._ctx.timing()['last_culled_fraction']
    print('config 4: taper culling skipped %.1f %% of the (source, baseline) pairs of the last snapshot' % (100 * culled))
    assert culled > 0.05
    cube = ia.skyvis_freq
    assert cube.shape == (bl_run.shape[0], ch.size, 3) and cube.dtype == NP.complex64
    nbl = bl.shape[0]
    for j in range(cfg['n_acc']):
        dc, altaz, keep = W.drift_snapshot_directions(sky, lat, j * cfg['t_acc'] * SIDEREAL_DEG_PER_SEC)
        assert NP.array_equal(ia.obs_catalog_indices[j], NP.flatnonzero(keep))
        flux = sky['flux_ref'][keep, None] * (ch[None, :] / sky['ref_freq']) ** sky['spindex'][keep, None]
        # the reference stores a supplied beam as float32 (interferometry.py:4466) before pb * fluxes (:6254): so does the checker
        beam = H.external_beam(cfg['beam_table'], cfg['beam_freqs'], NP.pi / 2 - NP.radians(altaz[:, 0]), NP.radians(altaz[:, 1]), ch)
        pb = beam.astype(NP.float32).astype(NP.float64) * flux
        ref = CO.skyvis(bl[sel], ch, dc, pb, zen, fwhm_deg=sky['fwhm_deg'][keep])
        scale = NP.sum(NP.abs(pb), axis=0)[None, :]
        err = float(NP.max(NP.abs(cube[sel, :, j] - ref) / scale))
        print('config 4 snapshot %d: max err / sum|pbflux| = %.3e' % (j, err))
        assert err <= 5e-6, j                      # the stated fp32 tolerance of the path
        assert NP.max(NP.abs(cube[nbl:, :, j] - NP.conj(cube[sel, :, j])) / scale) <= 1e-5          # V(-b) = conj V(b)
    assert NP.max(NP.abs(cube[:, :, 0] - cube[:, :, 2])) > 0                                      # the sky did drift


def test_config3_with_diffuse_half_one_snapshot():
    """BASELINE config 3 as worded: 1e4 point sources + nside=128 diffuse sky (108 048 sources above the horizon, taper on)
    on all 61 075 HERA-350 baselines x 1024 channels, fp32, spot-checked on 4 baselines."""
    cfg = W.config3(with_diffuse=True)
    bl, ch, sky = cfg['baselines'], cfg['channels'], cfg['sky']
    lat = -30.7224
    n = sky['dircos'].shape[0]
    assert n > 100000 and cfg['taper']
    skymod = SM.SkyModel(location=sky['altaz'], flux_ref=sky['flux_ref'], spindex=sky['spindex'], ref_freq=sky['ref_freq'],
                         src_shape=NP.stack((sky['fwhm_deg'], sky['fwhm_deg'], NP.zeros(n)), axis=1))
    ia = RI.InterferometerArray(['b%d' % i for i in range(bl.shape[0])], bl, ch, telescope={'id': 'hera', 'orientation': [90.0, 270.0], 'ocoords': 'altaz'},
                                latitude=lat, skycoords='altaz', pointing_coords='hadec')
    ia.observe((2457000.5, 0.0), {'Tnet': 100.0}, NP.ones(ch.size), [0.0, lat], skymod, 10.7, memsave=True)
    from conftest import body_class_sample
    sel, lift = body_class_sample(bl, ch, sky['dircos'], NP.array([0.0, 0.0, 1.0]), f32=True)
    assert sel.size >= 16 and lift.any() and (~lift).any()               # lifting and re-anchored groups, first / last / ragged
    pb = BO.airy_disk_pattern(14.0, sky['altaz'], ch, pointing_altaz=[90.0, 270.0]) * skymod.generate_spectrum(frequency=ch)
    ref = CO.skyvis(bl[sel], ch, sky['dircos'], pb, NP.array([0.0, 0.0, 1.0]), fwhm_deg=sky['fwhm_deg'])
    vis = ia._ctx.get_vis(slot=0)[sel]                       # complex128 device cube: five rows, not the whole 1 GB snapshot
    assert NP.max(NP.abs(vis - ref) / NP.sum(NP.abs(pb), axis=0)[None, :]) <= 5e-6
    tm = ia._ctx.timing()
    assert tm['last_terms'] == bl.shape[0] * ch.size * n and tm['last_chan_tile'] == 64 and tm['last_taper_group'] == 1


def test_config5_two_lsts_and_delay_transform():
    """Config 5 for two of its 120 LSTs at full size: 392 704-pixel diffuse sky drifting through HERA-350's beam, fp32, then the
    delay transform of the device-resident cube; visibilities and delay spectra spot-checked on 3 baselines."""
    cfg = W.config5(n_acc=2)
    bl, ch, sky, lat = cfg['baselines'], cfg['channels'], cfg['sky'], cfg['latitude']
    lst0 = 15.0
    skymod = _radec_skymodel(sky, lat, lst0)
    ia = RI.InterferometerArray(['b%d' % i for i in range(bl.shape[0])], bl, ch, telescope={'id': 'hera', 'orientation': [90.0, 270.0], 'ocoords': 'altaz'},
                                latitude=lat, skycoords='radec', pointing_coords='hadec')
    ia.reserve(2)
    for j in range(2):
        ia.observe((2457000.5 + j * cfg['t_acc'] / 86400.0, lst0 + j * cfg['t_acc'] * SIDEREAL_DEG_PER_SEC), {'Tnet': 100.0}, NP.ones(ch.size),
                   [0.0, lat], skymod, cfg['t_acc'], memsave=True)
    from conftest import body_class_sample
    zen = NP.array([0.0, 0.0, 1.0])
    sel, lift = body_class_sample(bl, ch, sky['dircos'], zen, f32=True)
    assert sel.size >= 16 and lift.any() and (~lift).any()
    w = NP.blackman(ch.size) + 0.05
    ia.delay_transform(pad=1.0, freq_wts=w, verbose=False)
    assert all(isinstance(s, RI._DeviceSlot) for s in ia._cube)          # the 2 GB cube stayed on the device through both stages
    lag = ia.skyvis_lag_rows(sel)                                        # (3, nlag, 2) from the device-resident spectra
    ref_cube = NP.empty((sel.size, ch.size, 2), dtype=NP.complex128)
    for j in range(2):
        dc, altaz, keep = W.drift_snapshot_directions(sky, lat, j * cfg['t_acc'] * SIDEREAL_DEG_PER_SEC)
        assert ia.obs_catalog_indices[j].size == int(keep.sum())
        pb = BO.airy_disk_pattern(14.0, altaz, ch, pointing_altaz=[90.0, 270.0]) \
            * (sky['flux_ref'][keep, None] * (ch[None, :] / sky['ref_freq']) ** sky['spindex'][keep, None])
        ref = CO.skyvis(bl[sel], ch, dc, pb, zen, fwhm_deg=sky['fwhm_deg'][keep])
        vis = ia._ctx.get_vis(slot=j)[sel]
        scale = NP.sum(NP.abs(pb), axis=0)[None, :]
        assert NP.max(NP.abs(vis - ref) / scale) <= 5e-6, j
        ref_cube[:, :, j] = vis
    ref_lag, _ = DO.delay_transform(ref_cube, NP.ones((sel.size, ch.size, 2)), NP.repeat(NP.repeat(w[None, :, None], sel.size, axis=0), 2, axis=2),
                                    ia.freq_resolution, pad=1.0)
    assert lag.shape == ref_lag.shape
    assert NP.max(NP.abs(lag - ref_lag)) <= 1e-10 * NP.max(NP.abs(ref_lag))


def test_config5_all_120_lsts_on_a_coarse_sky():
    """Config 5's full time axis inside the suite: HERA-350 x 1024 channels x ALL 120 LSTs of 10.7 s drift, fp32, with the sky coarsened
    to nside 32 (6.1e3 pixels instead of 3.9e5, so that the 120 sky-sums take seconds; the full sky runs in tools/run_config5_full.py:
    364 s fp32 / 846 s fp64).  The 120 GB cube (7.3e6 rows) stays in HBM; then the delay POWER spectra of every row on the device
    (K^2 (Mpc/h)^3, delay_spectrum.py:3659-3663, 3992).  Spot checks against the oracles: visibilities of the first, a middle and the last
    LST on 3 baselines, and their delay power against the oracle's transform of those rows."""
    from prisim_amd import delay_spectrum as DSM
    n_lst = 120
    cfg = W.config5(n_acc=n_lst, nside=32)
    bl, ch, sky, lat = cfg['baselines'], cfg['channels'], cfg['sky'], cfg['latitude']
    lst0 = 15.0
    skymod = _radec_skymodel(sky, lat, lst0)
    ia = RI.InterferometerArray(['b%d' % i for i in range(bl.shape[0])], bl, ch, telescope={'id': 'hera', 'orientation': [90.0, 270.0], 'ocoords': 'altaz'},
                                latitude=lat, skycoords='radec', pointing_coords='hadec')
    ia.reserve(n_lst)
    for j in range(n_lst):
        ia.observe((2457000.5 + j * cfg['t_acc'] / 86400.0, lst0 + j * cfg['t_acc'] * SIDEREAL_DEG_PER_SEC), {'Tnet': 100.0}, NP.ones(ch.size),
                   [0.0, lat], skymod, cfg['t_acc'], memsave=True)
    ia._ctx.sync()
    assert ia.n_acc == n_lst and all(isinstance(s, RI._DeviceSlot) for s in ia._cube)       # nothing left the device while observing
    sel = _spot(bl, 3)
    zen = NP.array([0.0, 0.0, 1.0])
    w = NP.blackman(ch.size) + 0.01
    k = DSM.power_constants(ch, {'id': 'hera'}, freq_wts=w)['factor']
    ia._ctx.delay_transform_device(n_lst, bpwts=w, pad=1.0, want_lag=False, want_power=True, power_scale=k)
    for j in (0, 61, n_lst - 1):
        dc, altaz, keep = W.drift_snapshot_directions(sky, lat, j * cfg['t_acc'] * SIDEREAL_DEG_PER_SEC)
        assert ia.obs_catalog_indices[j].size == int(keep.sum())
        pb = BO.airy_disk_pattern(14.0, altaz, ch, pointing_altaz=[90.0, 270.0]) \
            * (sky['flux_ref'][keep, None] * (ch[None, :] / sky['ref_freq']) ** sky['spindex'][keep, None])
        ref = CO.skyvis(bl[sel], ch, dc, pb, zen, fwhm_deg=sky['fwhm_deg'][keep])
        vis = ia._ctx.get_vis(slot=j)[sel]
        assert NP.max(NP.abs(vis - ref) / NP.sum(NP.abs(pb), axis=0)[None, :]) <= 5e-6, j
        ref_lag, _ = DO.delay_transform(vis[:, :, None], NP.ones((sel.size, ch.size, 1)), NP.repeat(w[None, :, None], sel.size, axis=0),
                                        ia.freq_resolution, pad=1.0)
        pw = ia._ctx.get_delay_power(j, 1, rows=sel)[0]                                    # (3, nlag)
        ref_pw = NP.abs(ref_lag[:, :, 0]) ** 2 * k
        assert pw.shape == ref_pw.shape and NP.all(NP.isfinite(pw))
        assert NP.max(NP.abs(pw - ref_pw)) <= 1e-9 * NP.max(ref_pw), j


@pytest.mark.parametrize('taper', [False, True])
def test_fused_gradient_at_config3_array_size(taper):
    """The fused V + baseline-gradient kernels (interferometry.py:6330, 6338, 6343) on the full HERA-350 array x 1024 channels -- every
    baseline group, lifting and plain bodies, the MFMA lane layout on partly filled last groups -- with 2000 sources, both precisions,
    checked on sampled baselines (first, last, around the lift / no-lift boundary) against the numpy oracle's gradient."""
    from prisim_amd import _abi
    cfg = W.config3(nsrc=2000)
    bl, ch, sky = cfg['baselines'], cfg['channels'], cfg['sky']
    zen = NP.array([0.0, 0.0, 1.0])
    fw = NP.full(sky['dircos'].shape[0], 0.46) if taper else None
    pb = BO.airy_disk_pattern(14.0, sky['altaz'], ch) * (sky['flux_ref'][:, None] * (ch[None, :] / sky['ref_freq']) ** sky['spindex'][:, None])
    sel = NP.unique(NP.concatenate((_spot(bl, 5), [57599, 57600, 61074])))
    ref, gref = O.skyvis(bl[sel], ch, sky['dircos'], pb, zen, fwhm_deg=fw, gradient=True)
    scale = NP.sum(NP.abs(pb), axis=0)[None, :]
    with _abi.Context(0) as ctx:
        ctx.set_array(bl, ch, nt_max=1)
        ctx.set_sky_analytic(sky['dircos'], sky['flux_ref'], sky['spindex'], sky['ref_freq'], _abi.PRISIM_BEAM_AIRY, 14.0, zen, zen, fwhm_deg=fw)
        for prec, tol in ((_abi.PRISIM_FP64, 1e-11), (_abi.PRISIM_FP32, 5e-6)):
            ctx.compute(precision=prec, want_grad=True)
            vis, grad = ctx.get_vis(want_grad=True)
            assert NP.max(NP.abs(vis[sel] - ref) / scale) <= tol, (taper, prec)
            for k in range(3):
                assert NP.max(NP.abs(grad[k][sel] - gref[k]) / scale) <= tol, (taper, prec, k)
            del vis, grad
