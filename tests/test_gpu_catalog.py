"""GPU parity of the device-resident catalogue path (include/prisim_hip.h, ABI 0.4): the per-snapshot sky geometry formed on the device
against the host statements it replaces (prisim_amd/geometry.py = GEOM.hadec2altaz / altaz2dircos as InterferometerArray.observe() calls
them, prisim/interferometry.py:6174-6180, 6204-6219, 6263), and the visibilities through it against the uploaded-sky path and the oracle."""
import os
import sys

import numpy as NP
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from prisim_amd import _abi, geometry as GEOM, layouts as LAY, workloads as W   # noqa: E402

pytestmark = pytest.mark.gpu

ZEN = NP.array([0.0, 0.0, 1.0])


def radec_catalogue(sky, lat, lst0):
    """The sky of a workload (local frame at lst0) as (RA, Dec), like prisim_amd.driver.build_skymodel."""
    hadec = GEOM.altaz2hadec(sky['altaz'], lat, units='degrees')
    return NP.stack(((lst0 - hadec[:, 0]) % 360.0, hadec[:, 1]), axis=1)


def host_roi(radec, lat, lst, roi_radius=90.0, coords='radec'):
    """What observe() forms on the host (prisim_amd/interferometry.py, the uploaded path): indices and direction cosines."""
    if coords == 'radec':
        altaz = GEOM.hadec2altaz(NP.stack((lst - radec[:, 0], radec[:, 1]), axis=1), lat, units='degrees')
    elif coords == 'hadec':
        altaz = GEOM.hadec2altaz(radec, lat, units='degrees')
    else:
        altaz = radec
    m2 = NP.arange(altaz.shape[0])[NP.where(altaz[:, 0] >= 90.0 - roi_radius)]
    host_roi.on_the_rim = NP.where(NP.abs(altaz[:, 0] - (90.0 - roi_radius)) < 1e-12)[0]      # see _same_indices
    return m2, GEOM.altaz2dircos(altaz[m2], 'degrees'), altaz[m2]


def _same_indices(idx, m2, rim):
    """Bit-identical index lists -- except for sources whose altitude sits within 1e-12 degrees of the rim of the region of interest,
    where one ulp of the device's and numpy's arcsin decides (HEALPix has rings at exactly 30 degrees of altitude: nside-16 at
    roi_radius = 60 and LST = the LST the map was laid out for).  Those are compared as sets without them."""
    if rim.size == 0:
        return NP.array_equal(idx, m2)
    return NP.array_equal(NP.setdiff1d(idx, rim), NP.setdiff1d(m2, rim))


def _check_dircos(dc, dc_host, altaz, what):
    """The device follows the host's chain statement by statement (FMA contraction off), so the two differ only where the device's
    sin / cos / asin / atan2 and numpy's round differently -- by one unit in the last place.  The chain carries the azimuth in [0, 2 pi)
    (GEOM.hadec2altaz), where one ulp is 8.9e-16 rad: a 1-ulp difference of atan2 alone moves (l, m) by that much, two such roundings by
    1.8e-15.  Measured on MI355X vs numpy 2.2 on the box's EPYC: about 70 % of the entries bit-identical, worst 1.7e-15 below 78 degrees of
    altitude; asserted: 2.5e-15 (3 ulp of an azimuth) -- 2e-12 cycles of phase on a 1 km baseline at 200 MHz.  Near the zenith the chain
    is ill-conditioned on the host and on the device alike (alt = arcsin(sin_alt), then cos(alt): one ulp of sin_alt comes back times
    1 / cos(alt)); there the bound is 2 ulp / cos(alt) (4.2e-15 measured at 87.1 degrees)."""
    err = NP.max(NP.abs(dc - dc_host), axis=1)
    tol = NP.maximum(2.5e-15, 4.5e-16 / NP.maximum(NP.cos(NP.radians(altaz[:, 0])), 1e-6))
    worst = int(NP.argmax(err / tol))
    assert NP.all(err <= tol), (what, float(err[worst]), float(altaz[worst, 0]))
    assert NP.mean(err == 0.0) > 0.5, (what, float(NP.mean(err == 0.0)))        # the arithmetic is the host's, statement by statement


CASES = [
    # (name, sky builder, latitude, lst0, LSTs)
    ('cfg2 nside-16 diffuse', lambda: W.diffuse_sky(16, 2), -30.7224, 30.0, [30.0, 33.7, 75.123, 211.0, 359.99]),
    ('cfg4 nside-64 diffuse', lambda: W.diffuse_sky(64, 44, f_ref=185e6), -26.701, 0.0, [0.0, 0.4679, 14.5, 181.25]),
    ('cfg5 nside-256 diffuse', lambda: W.diffuse_sky(256, 55), -30.7224, 10.0, [10.0, 15.35, 100.0]),
    ('cfg3 points + diffuse', lambda: W.concat_skies(W.point_source_sky(10000, 3), W.diffuse_sky(32, 33)), -30.7224, 55.5, [55.5, 56.0, 300.0]),
]


@pytest.mark.parametrize('case', CASES, ids=[c[0] for c in CASES])
def test_roi_indices_bit_identical_and_dircos_to_3ulp(case):
    """index arrays bit-identical and direction cosines within 3 ulp of an azimuth of prisim_amd/geometry.py on the
    configs' skies (see _check_dircos for why not 1e-15)."""
    _, mk, lat, lst0, lsts = case
    sky = mk()
    radec = radec_catalogue(sky, lat, lst0)
    with _abi.Context(0) as ctx:
        ctx.set_array(NP.array([[14.6, 0.0, 0.0]]), W.channel_grid(150e6, 1e5, 8), nt_max=1)
        ctx.set_catalog(radec, 'radec', flux_ref=sky['flux_ref'], spindex=sky['spindex'], ref_freq_hz=sky['ref_freq'], fwhm_deg=sky['fwhm_deg'])
        for roi_radius in (90.0, 60.0):
            obs = ctx.make_obs(lat, roi_radius_deg=roi_radius)
            for lst in lsts:
                idx, dc = ctx.catalog_roi(obs, lst, ZEN)
                m2, dc_host, altaz = host_roi(radec, lat, lst, roi_radius)
                rim = host_roi.on_the_rim
                assert idx.dtype == NP.int64 and _same_indices(idx, m2, rim), (lst, roi_radius, idx.size, m2.size, rim.size)
                if rim.size == 0:
                    assert dc.shape == dc_host.shape
                    if m2.size:
                        _check_dircos(dc, dc_host, altaz, (case[0], lst, roi_radius))
                else:
                    assert roi_radius == 60.0 and lst == lsts[0]            # the one constructed coincidence; never at the horizon
                    keep_d, keep_h = ~NP.isin(idx, rim), ~NP.isin(m2, rim)
                    _check_dircos(dc[keep_d], dc_host[keep_h], altaz[keep_h], (case[0], lst, roi_radius))


def test_roi_hadec_altaz_catalogues_and_pointing_centre_roi():
    sky = W.point_source_sky(5000, 7, alt_min_deg=-60.0) if False else W.point_source_sky(5000, 7)
    lat = 12.5
    rng = NP.random.default_rng(11)
    hadec = NP.stack((rng.uniform(-180, 180, 4000), NP.degrees(NP.arcsin(rng.uniform(-1, 1, 4000)))), axis=1)
    with _abi.Context(0) as ctx:
        ctx.set_array(NP.array([[14.6, 0.0, 0.0]]), W.channel_grid(150e6, 1e5, 8), nt_max=1)
        # HA-Dec catalogue: the LST plays no role
        ctx.set_catalog(hadec, 'hadec', flux_ref=NP.ones(4000), spindex=NP.zeros(4000), ref_freq_hz=150e6)
        obs = ctx.make_obs(lat, roi_radius_deg=90.0)
        idx, dc = ctx.catalog_roi(obs, 123.0, ZEN)
        m2, dc_host, _ = host_roi(hadec, lat, 0.0, 90.0, coords='hadec')
        assert NP.array_equal(idx, m2)
        _check_dircos(dc, dc_host, GEOM.hadec2altaz(hadec[m2], lat, units='degrees'), 'hadec')
        # alt-az catalogue: mask and direction cosines straight from the positions
        ctx.set_catalog(sky['altaz'], 'altaz', flux_ref=sky['flux_ref'], spindex=sky['spindex'], ref_freq_hz=sky['ref_freq'])
        obs = ctx.make_obs(lat, roi_radius_deg=40.0)
        idx, dc = ctx.catalog_roi(obs, 0.0, ZEN)
        m2, dc_host, _ = host_roi(sky['altaz'], lat, 0.0, 40.0, coords='altaz')
        assert m2.size > 0 and NP.array_equal(idx, m2) and float(NP.max(NP.abs(dc - dc_host))) <= 2.5e-16
        # region of interest about the pointing centre (:6210-6213)
        pc = GEOM.altaz2dircos(NP.array([55.0, 130.0]), 'degrees').ravel()
        obs = ctx.make_obs(lat, roi_radius_deg=25.0, roi_center='pointing_center')
        idx, dc = ctx.catalog_roi(obs, 0.0, pc)
        dc_all = GEOM.altaz2dircos(sky['altaz'], 'degrees')
        ang = NP.degrees(NP.arccos(NP.clip(NP.dot(dc_all, pc), -1.0, 1.0)))
        m2 = NP.where(ang <= 25.0)[0]
        # (the dot product is BLAS on the host: sources within 1e-12 deg of the rim may differ)
        rim = NP.abs(ang - 25.0) < 1e-12
        assert m2.size > 0 and NP.array_equal(NP.setdiff1d(idx, NP.where(rim)[0]), NP.setdiff1d(m2, NP.where(rim)[0]))


def _oracle_scale(pb):
    return NP.sum(NP.abs(pb), axis=0)[None, :]


def test_visibilities_through_the_catalogue_path_match_the_uploaded_path_and_the_oracle():
    """config-2-shaped drift scan in fp64 and fp32: set_sky_from_catalog + compute against set_sky_analytic of the host-formed sky
    (the same kernels behind both) and against the C oracle."""
    from oracle import c_oracle as CO, beams_oracle as BO
    cfg = W.config2()
    lat, lst0 = -30.7224, 40.0
    sky = cfg['sky']
    radec = radec_catalogue(sky, lat, lst0)
    ch = cfg['channels']
    with _abi.Context(0) as ctx, _abi.Context(0) as ref:
        for c in (ctx, ref):
            c.set_array(cfg['baselines'], ch, nt_max=1)
        ctx.set_catalog(radec, 'radec', flux_ref=sky['flux_ref'], spindex=sky['spindex'], ref_freq_hz=sky['ref_freq'], fwhm_deg=sky['fwhm_deg'])
        obs = ctx.make_obs(lat, beam_kind=_abi.PRISIM_BEAM_AIRY, diameter_m=14.0)
        for lst in (40.0, 47.5, 139.0):
            n = ctx.set_sky_from_catalog(obs, lst, ZEN, ZEN)
            m2, dc, altaz = host_roi(radec, lat, lst)
            assert n == m2.size
            ref.set_sky_analytic(dc, sky['flux_ref'][m2], sky['spindex'][m2], sky['ref_freq'], _abi.PRISIM_BEAM_AIRY, 14.0, ZEN, ZEN,
                                 fwhm_deg=sky['fwhm_deg'][m2])
            pb_dev, pb_ref = ctx.get_pbflux(), ref.get_pbflux()
            assert float(NP.max(NP.abs(pb_dev - pb_ref) / NP.max(NP.abs(pb_ref)))) <= 1e-13
            for prec, tol in ((_abi.PRISIM_FP64, 1e-11), (_abi.PRISIM_FP32, 5e-6)):
                ctx.compute(precision=prec)
                ref.compute(precision=prec)
                v, vr = ctx.get_vis(), ref.get_vis()
                scale = _oracle_scale(pb_ref)
                assert float(NP.max(NP.abs(v - vr) / scale)) <= (1e-13 if prec == _abi.PRISIM_FP64 else 5e-7)
                if lst == 47.5:
                    vo = CO.skyvis(cfg['baselines'], ch, dc, pb_ref, ZEN, fwhm_deg=sky['fwhm_deg'][m2])
                    assert float(NP.max(NP.abs(v - vo) / scale)) <= tol


def test_culling_order_and_table_on_the_device_config4_shape():
    """Long baselines over nside-64 pixels (config 4's shape, a 1/8 baseline shard): the catalogue path sorts every run by altitude and
    builds the cull table on the device; the result must agree with the uploaded path (host order + host table) within the fp32
    tolerance, cull about as much, and agree with the oracle that sums everything."""
    from oracle import c_oracle as CO
    cfg = W.config4(n_acc=1)
    lat, lst0 = cfg['latitude'], 0.0
    sky = cfg['sky']
    radec = radec_catalogue(sky, lat, lst0)
    bl = cfg['baselines'][3::8]
    ch = cfg['channels'][:128]
    with _abi.Context(0) as ctx, _abi.Context(0) as ref:
        for c in (ctx, ref):
            c.set_array(bl, ch, nt_max=1)
        ctx.set_catalog(radec, 'radec', flux_ref=sky['flux_ref'], spindex=sky['spindex'], ref_freq_hz=sky['ref_freq'], fwhm_deg=sky['fwhm_deg'])
        obs = ctx.make_obs(lat, beam_kind=_abi.PRISIM_BEAM_AIRY, diameter_m=4.0)
        for lst in (0.0, 3.3):
            n = ctx.set_sky_from_catalog(obs, lst, ZEN, ZEN)
            m2, dc, altaz = host_roi(radec, lat, lst)
            assert n == m2.size
            order = NP.argsort(-altaz[:, 0], kind='stable')          # what InterferometerArray._cull_order hands the uploaded path
            ref.set_sky_analytic(dc[order], sky['flux_ref'][m2][order], sky['spindex'][m2][order], sky['ref_freq'], _abi.PRISIM_BEAM_AIRY, 4.0,
                                 ZEN, ZEN, fwhm_deg=sky['fwhm_deg'][m2][order])
            for prec, tol in ((_abi.PRISIM_FP32, 5e-6), (_abi.PRISIM_FP64, 1e-11)):
                ctx.compute(precision=prec)
                ref.compute(precision=prec)
                v, vr = ctx.get_vis(), ref.get_vis()
                td, tr = ctx.timing(), ref.timing()
                assert tr['last_culled_fraction'] > 0.05, tr
                assert abs(td['last_culled_fraction'] - tr['last_culled_fraction']) <= 0.01 * tr['last_culled_fraction'] + 1e-6, (td, tr)
                pb = ref.get_pbflux()
                scale = _oracle_scale(pb)
                assert float(NP.max(NP.abs(v - vr) / scale)) <= tol
            sel = NP.arange(0, bl.shape[0], 97)
            vo = CO.skyvis(bl[sel], ch, dc[order], pb, ZEN, fwhm_deg=sky['fwhm_deg'][m2][order])
            assert float(NP.max(NP.abs(v[sel] - vo) / scale)) <= 1e-11


def test_tabulated_spectra_and_external_beam_through_the_index_list():
    """A catalogue with flux spectra (nsrc x nchan table, catalogue order) and the external HEALPix beam: both are read through the
    compacted index list on the device."""
    cfg = W.config4(n_acc=1)
    lat, lst0 = cfg['latitude'], 20.0
    sky = W.diffuse_sky(16, 5, f_ref=185e6)
    radec = radec_catalogue(sky, lat, lst0)
    bl = cfg['baselines'][::200]
    ch = cfg['channels'][:64]
    rng = NP.random.default_rng(5)
    spec = sky['flux_ref'][:, None] * (ch[None, :] / sky['ref_freq']) ** sky['spindex'][:, None] * rng.uniform(0.9, 1.1, (radec.shape[0], ch.size))
    from prisim_amd import primary_beams as PB
    m = PB.spectral_interp_matrix(cfg['beam_freqs'], ch, kind='cubic', chromatic=True, select_freq=None)
    with _abi.Context(0) as ctx, _abi.Context(0) as ref:
        for c in (ctx, ref):
            c.set_array(bl, ch, nt_max=1)
            c.set_external_beam(cfg['beam_table'], m)
        ctx.set_catalog(radec, 'radec', flux_spectrum=spec, fwhm_deg=sky['fwhm_deg'])
        obs = ctx.make_obs(lat, use_external_beam=True)
        for lst in (27.0, 41.0):
            n = ctx.set_sky_from_catalog(obs, lst, ZEN, ZEN)
            m2, dc, altaz = host_roi(radec, lat, lst)
            assert n == m2.size and 0 < n < radec.shape[0]
            # (kilometre baselines over 3.7-degree pixels: the catalogue path lists the sources by decreasing altitude, like
            # InterferometerArray._cull_order does for the uploaded path)
            order = NP.argsort(-altaz[:, 0], kind='stable')
            m2, dc = m2[order], dc[order]
            ref.set_sky_external(dc, spec[m2], ZEN, fwhm_deg=sky['fwhm_deg'][m2])
            pb_dev, pb_ref = ctx.get_pbflux(), ref.get_pbflux()
            assert float(NP.max(NP.abs(pb_dev - pb_ref) / NP.max(NP.abs(pb_ref)))) <= 1e-6      # (float32 beam storage, :4466)
            ctx.compute(precision=_abi.PRISIM_FP64)
            ref.compute(precision=_abi.PRISIM_FP64)
            scale = _oracle_scale(pb_ref)
            assert float(NP.max(NP.abs(ctx.get_vis() - ref.get_vis()) / scale)) <= 5e-6


def test_observe_catalog_many_snapshots_equals_one_by_one():
    """prisim_hip_observe_catalog (K snapshots, one geometry pass, no host synchronisation between snapshots) leaves in slots
    slot0 ... slot0 + K - 1 exactly what K calls of set_sky_from_catalog + compute leave."""
    cfg = W.config2()
    lat, lst0 = -30.7224, 10.0
    sky = cfg['sky']
    radec = radec_catalogue(sky, lat, lst0)
    lsts = 10.0 + 0.25 * NP.arange(7)
    bl = NP.vstack((cfg['baselines'], cfg['baselines'] * 1.7))[:300]          # two baseline groups: the per-snapshot loop, not wave items
    with _abi.Context(0) as ctx, _abi.Context(0) as one:
        ctx.set_array(bl, cfg['channels'], nt_max=8)
        one.set_array(bl, cfg['channels'], nt_max=1)
        for c in (ctx, one):
            c.set_catalog(radec, 'radec', flux_ref=sky['flux_ref'], spindex=sky['spindex'], ref_freq_hz=sky['ref_freq'], fwhm_deg=sky['fwhm_deg'])
        obs = ctx.make_obs(lat, beam_kind=_abi.PRISIM_BEAM_AIRY, diameter_m=14.0)
        counts = ctx.observe_catalog(obs, lsts, ZEN, precision=_abi.PRISIM_FP64, slot0=1)
        for t, lst in enumerate(lsts):
            n = one.set_sky_from_catalog(obs, lst, ZEN, ZEN)
            assert n == counts[t]
            one.compute(precision=_abi.PRISIM_FP64)
            assert NP.array_equal(ctx.get_vis(slot=1 + t), one.get_vis())


def test_catalogue_path_state_errors():
    with _abi.Context(0) as ctx:
        ctx.set_array(NP.array([[14.6, 0.0, 0.0]]), W.channel_grid(150e6, 1e5, 8), nt_max=1)
        obs = ctx.make_obs(-30.0)
        with pytest.raises(RuntimeError):
            ctx.set_sky_from_catalog(obs, 0.0, ZEN)                  # no catalogue yet
        ctx.set_catalog(NP.zeros((0, 2)), 'radec', flux_ref=NP.zeros(0), spindex=NP.zeros(0), ref_freq_hz=150e6)
        assert ctx.set_sky_from_catalog(obs, 0.0, ZEN) == 0          # an empty catalogue is a valid (empty) sky (:6378-6382)
        ctx.compute()
        assert NP.all(ctx.get_vis() == 0)
        ctx.set_catalog(NP.array([[10.0, -30.0]]), 'radec', flux_ref=NP.ones(1), spindex=NP.zeros(1), ref_freq_hz=150e6)
        ctx.set_array(NP.array([[14.6, 0.0, 0.0]]), W.channel_grid(150e6, 1e5, 8), nt_max=1)
        with pytest.raises(RuntimeError):
            ctx.set_sky_from_catalog(obs, 10.0, ZEN)                 # set_array drops the catalogue
        with pytest.raises(ValueError):
            ctx.set_catalog(NP.array([[NP.nan, 0.0]]), 'radec', flux_ref=NP.ones(1), spindex=NP.zeros(1), ref_freq_hz=150e6)


# ---- many snapshots of a small array in one launch (prisim_hip_observe_catalog, wave items over a batch) ----
def _small_array_case(nbl, nchan, nside=16, seed=2):
    cfg = W.config2()
    bl = cfg['baselines']
    if nbl > bl.shape[0]:
        bl = NP.vstack((bl, bl * 1.37))
    bl = bl[:nbl]
    ch = W.channel_grid(150e6, 390625.0, nchan)
    sky = W.diffuse_sky(nside, seed)
    return bl, ch, sky


@pytest.mark.parametrize('nbl,nchan,k', [(171, 256, 64), (3, 64, 9), (64, 40, 5), (65, 96, 7), (256, 128, 12)])
def test_batched_snapshots_bit_identical_to_single_launches(nbl, nchan, k):
    """K LSTs of a small array in ONE sky-sum launch + ONE reduction; results bit-identical to K single launches cut
    the same way (set_tuning with the batch's tile and split count), and within 1e-13 of the planner's own single launches."""
    bl, ch, sky = _small_array_case(nbl, nchan)
    lat, lst0 = -30.7224, 25.0
    radec = radec_catalogue(sky, lat, lst0)
    lsts = lst0 + 0.75 * NP.arange(k)
    with _abi.Context(0) as ctx, _abi.Context(0) as one:
        ctx.set_array(bl, ch, nt_max=k)
        one.set_array(bl, ch, nt_max=1)
        for c in (ctx, one):
            c.set_catalog(radec, 'radec', flux_ref=sky['flux_ref'], spindex=sky['spindex'], ref_freq_hz=sky['ref_freq'], fwhm_deg=sky['fwhm_deg'])
        obs = ctx.make_obs(lat, beam_kind=_abi.PRISIM_BEAM_AIRY, diameter_m=14.0)
        counts = ctx.observe_catalog(obs, lsts, ZEN, precision=_abi.PRISIM_FP64)
        tm = ctx.timing()
        assert tm['last_batch_snapshots'] == k, tm                     # the whole chunk went into one launch
        assert tm['last_chan_tile'] in (16, 32)
        batch = [ctx.get_vis(slot=t) for t in range(k)]
        # single launches, partitioned like the batch
        one.set_tuning(tm['last_chan_tile'], 0, tm['last_nsplit'])
        worst = 0.0
        for t, lst in enumerate(lsts):
            n = one.set_sky_from_catalog(obs, lst, ZEN, ZEN)
            assert n == counts[t]
            one.compute(precision=_abi.PRISIM_FP64)
            assert NP.array_equal(batch[t], one.get_vis()), (t, float(NP.max(NP.abs(batch[t] - one.get_vis()))))
        # ... and the planner's own single launches (other tiles, other splits): the same sums to rounding
        one.set_tuning(0, 0, 0)
        for t in (0, k // 2, k - 1):
            one.set_sky_from_catalog(obs, lsts[t], ZEN, ZEN)
            one.compute(precision=_abi.PRISIM_FP64)
            scale = NP.sum(NP.abs(one.get_pbflux()), axis=0)[None, :]
            worst = max(worst, float(NP.max(NP.abs(batch[t] - one.get_vis()) / scale)))
        assert worst <= 1e-13, worst


@pytest.mark.parametrize('prep_stream', [False, True])
def test_batched_chunks_queued_back_to_back(prep_stream):
    """More snapshots than one launch takes (chunks of 64): the chunks are queued without the host waiting for any of them -- the pinned
    per-chunk tables, the geometry sets and the row buffers are all reused while earlier chunks are still queued.  Every slot must be
    the single launch's result, with the chunk prepared in line (the default) and on the preparation stream."""
    bl, ch, sky = _small_array_case(65, 40)
    lat, lst0 = -30.7224, 25.0
    radec = radec_catalogue(sky, lat, lst0)
    k = 215                                                      # 64 + 64 + 64 + 23
    lsts = lst0 + 0.4 * NP.arange(k)
    os.environ['PRISIM_HIP_PREP_ASYNC_BATCH'] = '1' if prep_stream else '0'
    try:
        with _abi.Context(0) as ctx, _abi.Context(0) as one:
            ctx.set_array(bl, ch, nt_max=k)
            one.set_array(bl, ch, nt_max=1)
            for c in (ctx, one):
                c.set_catalog(radec, 'radec', flux_ref=sky['flux_ref'], spindex=sky['spindex'], ref_freq_hz=sky['ref_freq'], fwhm_deg=sky['fwhm_deg'])
            obs = ctx.make_obs(lat, beam_kind=_abi.PRISIM_BEAM_AIRY, diameter_m=14.0)
            for rep in range(2):                                     # the second call finds every buffer in use by the first
                counts = ctx.observe_catalog(obs, lsts, ZEN, precision=_abi.PRISIM_FP64)
            tm = ctx.timing()
            assert tm['last_batch_snapshots'] == k - 3 * 64, tm
            one.set_tuning(tm['last_chan_tile'], 0, tm['last_nsplit'])
            for t in list(range(0, k, 7)) + [63, 64, 127, 128, 191, 192, k - 1]:
                n = one.set_sky_from_catalog(obs, lsts[t], ZEN, ZEN)
                assert n == counts[t]
                one.compute(precision=_abi.PRISIM_FP64)
                want, got = one.get_vis(), ctx.get_vis(slot=t)
                scale = NP.sum(NP.abs(one.get_pbflux()), axis=0)[None, :]
                # (the last chunk is cut into other source splits than the full ones: same sums to rounding there, bit-identical in the full chunks
                # only if their split count equals the last chunk's)
                assert float(NP.max(NP.abs(got - want) / scale)) <= 1e-13, t
    finally:
        del os.environ['PRISIM_HIP_PREP_ASYNC_BATCH']


@pytest.mark.parametrize('spectra', [False, True])
def test_batched_snapshots_with_the_external_healpix_beam(spectra):
    """Small arrays with an external HEALPix beam (what HERA-sized runs use) go through the batched launch too: gather, per-snapshot
    column maximum and 10 ** (.) x flux of the whole chunk in four launches -- bit-identical to one launch per snapshot."""
    from prisim_amd import primary_beams as PB
    cfg4 = W.config4(n_acc=1)
    bl, ch, sky = _small_array_case(171, 64)
    lat, lst0 = -30.7224, 25.0
    radec = radec_catalogue(sky, lat, lst0)
    beam_freqs = NP.linspace(float(ch[0]) - 5e6, float(ch[-1]) + 5e6, cfg4['beam_freqs'].size)
    m = PB.spectral_interp_matrix(beam_freqs, ch, kind='cubic', chromatic=True, select_freq=None)
    rng = NP.random.default_rng(8)
    spec = sky['flux_ref'][:, None] * (ch[None, :] / sky['ref_freq']) ** sky['spindex'][:, None] * rng.uniform(0.9, 1.1, (radec.shape[0], ch.size))
    k = 11
    lsts = lst0 + 1.5 * NP.arange(k)
    lsts[4] = lst0 + 180.0                                       # only the circumpolar part of this catalogue is up: a short snapshot inside the chunk
    with _abi.Context(0) as ctx, _abi.Context(0) as one:
        ctx.set_array(bl, ch, nt_max=k)
        one.set_array(bl, ch, nt_max=1)
        for c in (ctx, one):
            c.set_external_beam(cfg4['beam_table'], m)
            if spectra:
                c.set_catalog(radec, 'radec', flux_spectrum=spec, fwhm_deg=sky['fwhm_deg'])
            else:
                c.set_catalog(radec, 'radec', flux_ref=sky['flux_ref'], spindex=sky['spindex'], ref_freq_hz=sky['ref_freq'], fwhm_deg=sky['fwhm_deg'])
        obs = ctx.make_obs(lat, use_external_beam=True)
        counts = ctx.observe_catalog(obs, lsts, ZEN, precision=_abi.PRISIM_FP64)
        tm = ctx.timing()
        assert tm['last_batch_snapshots'] == k, tm
        assert 0 < counts[4] < counts[0] // 2
        one.set_tuning(tm['last_chan_tile'], 0, tm['last_nsplit'])
        for t, lst in enumerate(lsts):
            n = one.set_sky_from_catalog(obs, lst, ZEN, ZEN)
            assert n == counts[t]
            if n == 0:
                continue
            one.compute(precision=_abi.PRISIM_FP64)
            assert NP.array_equal(ctx.get_vis(slot=t), one.get_vis()), t


def test_chunks_of_both_kinds_and_an_uploaded_sky_on_one_context():
    """65 snapshots = one batched chunk prepared in line + one single snapshot prepared on the preparation stream, right behind an
    uploaded sky's sum on the same context: every change of the preparing stream has to wait for the sums still reading what it is
    about to overwrite (buffer sets, work areas)."""
    bl, ch, sky = _small_array_case(171, 128)
    lat, lst0 = -30.7224, 25.0
    radec = radec_catalogue(sky, lat, lst0)
    k = 65
    lsts = lst0 + 0.5 * NP.arange(k)
    with _abi.Context(0) as ctx, _abi.Context(0) as one:
        ctx.set_array(bl, ch, nt_max=k)
        one.set_array(bl, ch, nt_max=1)
        for c in (ctx, one):
            c.set_catalog(radec, 'radec', flux_ref=sky['flux_ref'], spindex=sky['spindex'], ref_freq_hz=sky['ref_freq'], fwhm_deg=sky['fwhm_deg'])
        obs = ctx.make_obs(lat, beam_kind=_abi.PRISIM_BEAM_AIRY, diameter_m=14.0)
        m2, dc, altaz = host_roi(radec, lat, lst0)
        for rep in range(3):
            # an uploaded sky summed several times (long enough to still be running), then the catalogue path at once
            ctx.set_sky_analytic(dc, sky['flux_ref'][m2], sky['spindex'][m2], sky['ref_freq'], _abi.PRISIM_BEAM_AIRY, 14.0, ZEN, ZEN, fwhm_deg=sky['fwhm_deg'][m2])
            for q in range(4):
                ctx.compute(precision=_abi.PRISIM_FP64, slot=k - 1)
            counts = ctx.observe_catalog(obs, lsts, ZEN, precision=_abi.PRISIM_FP64)
            # ... and single snapshots (preparation stream) right behind the last chunk's
            n_last = ctx.set_sky_from_catalog(obs, lsts[3], ZEN, ZEN)
            ctx.compute(precision=_abi.PRISIM_FP64, slot=3)
            assert n_last == counts[3]
        tm_one = None
        for t in (0, 3, 31, 63, 64):
            one.set_tuning(0, 0, 0)
            one.set_sky_from_catalog(obs, lsts[t], ZEN, ZEN)
            one.compute(precision=_abi.PRISIM_FP64)
            scale = NP.sum(NP.abs(one.get_pbflux()), axis=0)[None, :]
            assert float(NP.max(NP.abs(ctx.get_vis(slot=t) - one.get_vis()) / scale)) <= 1e-13, t


def test_batched_launch_on_a_catalogue_without_source_shapes():
    """A sky model without src_shape (point sources, no taper) on a small array: the batched launch treats it as kappa = 0 everywhere --
    weight exactly 1 -- and gives the plain kernel's sums to rounding."""
    bl, ch, sky = _small_array_case(65, 96)
    lat, lst0 = -30.7224, 25.0
    radec = radec_catalogue(sky, lat, lst0)
    k = 9
    lsts = lst0 + 0.75 * NP.arange(k)
    with _abi.Context(0) as ctx, _abi.Context(0) as one:
        ctx.set_array(bl, ch, nt_max=k)
        one.set_array(bl, ch, nt_max=1)
        for c in (ctx, one):
            c.set_catalog(radec, 'radec', flux_ref=sky['flux_ref'], spindex=sky['spindex'], ref_freq_hz=sky['ref_freq'])
        obs = ctx.make_obs(lat, beam_kind=_abi.PRISIM_BEAM_GAUSSIAN, diameter_m=14.0)
        counts = ctx.observe_catalog(obs, lsts, ZEN, precision=_abi.PRISIM_FP64)
        assert ctx.timing()['last_batch_snapshots'] == k
        for t, lst in enumerate(lsts):
            assert one.set_sky_from_catalog(obs, lst, ZEN, ZEN) == counts[t]
            one.compute(precision=_abi.PRISIM_FP64)
            scale = NP.sum(NP.abs(one.get_pbflux()), axis=0)[None, :]
            assert float(NP.max(NP.abs(ctx.get_vis(slot=t) - one.get_vis()) / scale)) <= 1e-13, t


def test_batched_snapshots_against_the_oracle_with_empty_and_moving_pointings():
    """The batched launch against the C oracle: per-snapshot phase and beam pointing centres, a snapshot with nothing above the horizon
    (its slot must hold zeros), source sizes that vary from source to source (no runs), a flux-spectrum table read through the index list."""
    from oracle import c_oracle as CO, beams_oracle as BO
    bl, ch, _ = _small_array_case(100, 64)
    lat = -30.7224
    rng = NP.random.default_rng(21)
    n = 900
    radec = NP.stack((rng.uniform(40.0, 120.0, n), rng.uniform(-45.0, 0.0, n)), axis=1)       # a patch: below the horizon half a day later
    fwhm = rng.uniform(0.0, 1.5, n)
    spec = rng.uniform(1.0, 5.0, (n, 1)) * (ch[None, :] / 150e6) ** rng.uniform(-1.0, -0.5, (n, 1))
    lsts = NP.array([80.0, 95.0, 260.0, 110.0])                                              # third one: nothing up
    pcs = GEOM.altaz2dircos(NP.array([[90.0, 0.0], [80.0, 45.0], [90.0, 0.0], [70.0, 200.0]]), 'degrees')
    with _abi.Context(0) as ctx:
        ctx.set_array(bl, ch, nt_max=4)
        ctx.set_catalog(radec, 'radec', flux_spectrum=spec, fwhm_deg=fwhm)
        obs = ctx.make_obs(lat, beam_kind=_abi.PRISIM_BEAM_GAUSSIAN, diameter_m=14.0)
        counts = ctx.observe_catalog(obs, lsts, pcs, pcs, precision=_abi.PRISIM_FP64)
        assert ctx.timing()['last_batch_snapshots'] == 4
        assert counts[2] == 0 and NP.all(ctx.get_vis(slot=2) == 0)
        for t in (0, 1, 3):
            m2, dc, altaz = host_roi(radec, lat, lsts[t])
            assert counts[t] == m2.size > 0
            pb = BO.gaussian_beam(14.0, altaz, ch, pointing_altaz=GEOM.dircos2altaz(pcs[t]).ravel(), power=True) * spec[m2]
            ref = CO.skyvis(bl, ch, dc, pb, pcs[t], fwhm_deg=fwhm[m2])
            scale = NP.sum(NP.abs(pb), axis=0)[None, :]
            assert float(NP.max(NP.abs(ctx.get_vis(slot=t) - ref) / scale)) <= 1e-11


@pytest.mark.parametrize('external_beam', [False, True])
def test_observing_run_on_a_small_array_uses_the_batched_launch(external_beam):
    """InterferometerArray.observing_run (interferometry.py:6414-6657) on HERA-19: the batched launch against the same run with
    PRISIM_HIP_WAVE_BATCH=0 (one launch per snapshot) and against PRISIM_CATALOG=0 (the sky of every snapshot formed on the host);
    with the analytic Airy beam and with an external HEALPix beam."""
    from prisim_amd import interferometry as RI, skymodel as SM
    cfg = W.config2()
    cfg4 = W.config4(n_acc=1)
    beam_freqs = NP.linspace(float(cfg['channels'][0]) - 5e6, float(cfg['channels'][-1]) + 5e6, cfg4['beam_freqs'].size)
    lat = -30.7224
    sky = cfg['sky']
    radec = radec_catalogue(sky, lat, 15.0 * 1.0)
    n = radec.shape[0]
    tel = {'id': 'hera', 'shape': 'dish', 'size': 14.0, 'ocoords': 'altaz', 'orientation': NP.array([[90.0, 270.0]]), 'groundplane': None}

    def run():
        skymod = SM.SkyModel(location=radec, flux_ref=sky['flux_ref'], spindex=sky['spindex'], ref_freq=sky['ref_freq'],
                             src_shape=NP.stack((sky['fwhm_deg'], sky['fwhm_deg'], NP.zeros(n)), axis=1))
        ia = RI.InterferometerArray(['b%d' % i for i in range(cfg['baselines'].shape[0])], cfg['baselines'], cfg['channels'], telescope=tel,
                                    latitude=lat, skycoords='radec', pointing_coords='hadec')
        if external_beam:
            ia.set_external_beam(cfg4['beam_table'], beam_freqs)
        ia.observing_run(NP.array([0.0, lat]), skymod, 120.0, 120.0 * 24, cfg['channels'], NP.ones(cfg['channels'].size), 100.0, 1.0, mode='drift')
        tm = ia._ctx.timing()
        return ia, NP.array(ia.skyvis_freq), tm

    ia, vis, tm = run()
    assert vis.shape == (171, 256, 24) and tm['last_batch_snapshots'] == 24
    assert [e.size for e in ia.obs_catalog_indices][0] == n and len(ia.lst) == 24 and ia.n_acc == 24
    os.environ['PRISIM_HIP_WAVE_BATCH'] = '0'
    try:
        _, vis1, tm1 = run()
    finally:
        del os.environ['PRISIM_HIP_WAVE_BATCH']
    assert tm1['last_batch_snapshots'] == 1
    os.environ['PRISIM_CATALOG'] = '0'
    try:
        ia0, vis0, _ = run()
    finally:
        del os.environ['PRISIM_CATALOG']
    scale = float(NP.max(NP.abs(vis0)))
    # (external beam: the beam is stored as float32 (:4466) and the two paths' direction cosines differ by ulps, which can move a value
    # across a float32 rounding boundary: 6e-8 of one source's beam)
    assert float(NP.max(NP.abs(vis - vis1))) <= 1e-12 * scale and float(NP.max(NP.abs(vis - vis0))) <= (1e-8 if external_beam else 1e-12) * scale
    # lazy class state of the catalogue path equals the host path's
    assert NP.array_equal(NP.asarray(ia.obs_catalog_indices[5]), NP.asarray(ia0.obs_catalog_indices[5]))
    assert float(NP.max(NP.abs(NP.asarray(ia.geometric_delays[5]) - NP.asarray(ia0.geometric_delays[5])))) <= 1e-20


def test_equatorial_baselines_give_the_enu_visibilities():
    """baseline_coords='equatorial' (rotated to ENU at the array's latitude, interferometry.py:6151-6153) against the same array
    given in ENU, through observe() on the catalogue path, and against the oracle."""
    from prisim_amd import interferometry as RI, skymodel as SM
    from oracle import c_oracle as CO, beams_oracle as BO
    rng = NP.random.default_rng(8)
    lat = -30.7224
    bl = rng.uniform(-150.0, 150.0, size=(40, 3)) * NP.array([1.0, 1.0, 0.05])
    ch = W.channel_grid(150e6, 2e5, 48)
    sky = W.point_source_sky(300, 4)
    radec = radec_catalogue(sky, lat, 70.0)
    skymod = SM.SkyModel(location=radec, flux_ref=sky['flux_ref'], spindex=sky['spindex'], ref_freq=sky['ref_freq'])
    tel = {'id': 'custom', 'shape': 'gaussian', 'size': 14.0, 'ocoords': 'altaz', 'orientation': NP.array([[90.0, 270.0]]), 'groundplane': None}
    out = []
    for coords, arr in (('localenu', bl), ('equatorial', GEOM.enu2xyz(bl, lat, 'degrees'))):
        ia = RI.InterferometerArray(['b%d' % i for i in range(40)], arr, ch, telescope=tel, latitude=lat, skycoords='radec', pointing_coords='hadec',
                                    baseline_coords=coords)
        ia.observe((2457000.5, 72.5), {'Tnet': 100.0}, NP.ones(ch.size), [0.0, lat], skymod, 10.0)
        out.append(NP.array(ia.skyvis_freq[:, :, 0]))
    m2, dc, altaz = host_roi(radec, lat, 72.5)
    pb = BO.gaussian_beam(14.0, altaz, ch, power=True) * sky['flux_ref'][m2, None] * (ch[None, :] / sky['ref_freq']) ** sky['spindex'][m2, None]
    ref = CO.skyvis(bl, ch, dc, pb, ZEN)
    scale = NP.sum(NP.abs(pb), axis=0)[None, :]
    assert float(NP.max(NP.abs(out[0] - ref) / scale)) <= 1e-11
    assert float(NP.max(NP.abs(out[1] - out[0]) / scale)) <= 1e-12          # (the rotation to ENU and back rounds the baselines at 1e-16)


def test_post_actions_of_a_batch_host_staging_and_one_rank_gather():
    """What prisim_hip_observe_catalog queues behind every snapshot (prisim_post): the download into the page-locked host cube
    (reserve(host_staging=True) through observe_batch -- loop chunks and the batched launch alike) and the RCCL all-gather of the slot on
    the communication stream (a 1-rank communicator on this box)."""
    from prisim_amd import interferometry as RI, skymodel as SM
    lat = -30.7224
    tel = {'id': 'hera', 'shape': 'dish', 'size': 14.0, 'ocoords': 'altaz', 'orientation': NP.array([[90.0, 270.0]]), 'groundplane': None}
    for nbl, memsave in ((171, False), (400, True)):              # 171: one launch for all snapshots; 400: the per-snapshot loop, complex64 staging
        bl, ch, sky = _small_array_case(nbl if nbl <= 256 else 256, 64)
        if nbl > 256:
            bl = NP.vstack((bl, bl[:nbl - 256] * 2.1))
        radec = radec_catalogue(sky, lat, 20.0)
        n = radec.shape[0]
        skymod = SM.SkyModel(location=radec, flux_ref=sky['flux_ref'], spindex=sky['spindex'], ref_freq=sky['ref_freq'],
                             src_shape=NP.stack((sky['fwhm_deg'], sky['fwhm_deg'], NP.zeros(n)), axis=1))
        times = [(2457000.5 + j * 1e-3, 20.0 + 0.5 * j) for j in range(6)]
        cubes = []
        for staging in (True, False):
            ia = RI.InterferometerArray(['b%d' % i for i in range(bl.shape[0])], bl, ch, telescope=tel, latitude=lat, skycoords='radec',
                                        pointing_coords='hadec')
            ia.reserve(6, host_staging=staging)
            ia.observe_batch(times, {'Tnet': 100.0}, NP.ones(ch.size), [0.0, lat], skymod, 10.0, memsave=memsave)
            if staging:
                snaps = ia.skyvis_freq_snapshots()
                assert ia._host_cube is not None and snaps is ia._host_cube or snaps.base is ia._host_cube or NP.shares_memory(snaps, ia._host_cube)
                assert all(sn.staged for sn in ia._cube)
            cubes.append(NP.array(ia.skyvis_freq))
            assert cubes[-1].dtype == (NP.complex64 if memsave else NP.complex128) and cubes[-1].shape == (bl.shape[0], ch.size, 6)
        assert NP.array_equal(cubes[0], cubes[1])
    # the slot gathers of a batch on a 1-rank communicator
    bl, ch, sky = _small_array_case(171, 64)
    radec = radec_catalogue(sky, lat, 20.0)
    with _abi.Context(0) as ctx:
        ctx.set_array(bl, ch, nt_max=5)
        ctx.set_catalog(radec, 'radec', flux_ref=sky['flux_ref'], spindex=sky['spindex'], ref_freq_hz=sky['ref_freq'], fwhm_deg=sky['fwhm_deg'])
        ctx.comm_init(_abi.Context.comm_unique_id(), 1, 0)
        ctx.comm_selftest()
        obs = ctx.make_obs(lat, beam_kind=_abi.PRISIM_BEAM_AIRY, diameter_m=14.0)
        ctx.observe_catalog(obs, 20.0 + 0.5 * NP.arange(5), ZEN, precision=_abi.PRISIM_FP64, gather='c128')
        g = ctx.get_gathered(5, 1)                                  # (nt, nranks, nbl, nchan)
        for t in range(5):
            assert NP.array_equal(g[t, 0], ctx.get_vis(slot=t))
        assert ctx.comm_stats()['n_gathers'] == 5
        assert 'librccl' in _abi.Context.comm_version()


def test_batched_launch_on_a_sky_of_several_source_runs():
    """Point sources + a diffuse map (two runs of one source size each) on a small array: one batched launch for all snapshots (every source
    carries its own kappa); against single launches (which sum the sky run by run, the point sources through the no-taper kernel) and the
    C oracle."""
    from oracle import c_oracle as CO, beams_oracle as BO
    bl, ch, _ = _small_array_case(171, 96)
    lat, lst0 = -30.7224, 50.0
    sky = W.concat_skies(W.point_source_sky(700, 9), W.diffuse_sky(8, 10))
    radec = radec_catalogue(sky, lat, lst0)
    lsts = lst0 + 1.5 * NP.arange(6)
    with _abi.Context(0) as ctx, _abi.Context(0) as one:
        ctx.set_array(bl, ch, nt_max=6)
        one.set_array(bl, ch, nt_max=1)
        for c in (ctx, one):
            c.set_catalog(radec, 'radec', flux_ref=sky['flux_ref'], spindex=sky['spindex'], ref_freq_hz=sky['ref_freq'], fwhm_deg=sky['fwhm_deg'])
        obs = ctx.make_obs(lat, beam_kind=_abi.PRISIM_BEAM_AIRY, diameter_m=14.0)
        counts = ctx.observe_catalog(obs, lsts, ZEN, precision=_abi.PRISIM_FP64)
        assert ctx.timing()['last_batch_snapshots'] == 6
        for t in (0, 3, 5):
            assert one.set_sky_from_catalog(obs, lsts[t], ZEN, ZEN) == counts[t]
            one.compute(precision=_abi.PRISIM_FP64)
            pb = one.get_pbflux()
            scale = NP.sum(NP.abs(pb), axis=0)[None, :]
            assert float(NP.max(NP.abs(ctx.get_vis(slot=t) - one.get_vis()) / scale)) <= 1e-13
            m2, dc, altaz = host_roi(radec, lat, lsts[t])
            ref = CO.skyvis(bl, ch, dc, pb, ZEN, fwhm_deg=sky['fwhm_deg'][m2])
            assert float(NP.max(NP.abs(ctx.get_vis(slot=t) - ref) / scale)) <= 1e-11
