"""GPU parity of the device-resident catalogue path (include/prisim_hip.h, ABI 0.5): the per-snapshot sky geometry formed on the device
against the host statements it replaces (prisim_amd/geometry.py = GEOM.hadec2altaz / altaz2dircos as InterferometerArray.observe() calls
them, prisim/interferometry.py:6174-6180, 6204-6219, 6263), and the visibilities through it against the uploaded-sky path and the oracle."""
import os
import sys

import numpy as NP
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from prisim_amd import _abi, frames as FR, geometry as GEOM, layouts as LAY, workloads as W   # noqa: E402

pytestmark = pytest.mark.gpu

ZEN = NP.array([0.0, 0.0, 1.0])


def radec_catalogue(sky, lat, lst0):
    """The sky of a workload (local frame at lst0) as (RA, Dec), like prisim_amd.driver.build_skymodel."""
    hadec = GEOM.altaz2hadec(sky['altaz'], lat, units='degrees')
    return NP.stack(((lst0 - hadec[:, 0]) % 360.0, hadec[:, 1]), axis=1)


def host_roi(radec, lat, lst, roi_radius=90.0, coords='radec', frame=None):
    """What observe() forms on the host (prisim_amd/interferometry.py _upload_snapshot_sky): geometry.frame_dircos + roi_select -- the host
    statement of cat_source() in catalog_kernels.hip; frame None = the library's fall-back rotation (hour angle = LST - RA).
    Returns indices, direction cosines and (alt, az) of the region of interest."""
    if frame is None:
        frame = FR.snapshot_frame(coords, lst, lat, model='date')
    dc_all = GEOM.frame_dircos(GEOM.catalog_unitvec(radec, coords), frame[0], frame[1])
    m2 = GEOM.roi_select(dc_all, 'zenith', roi_radius)
    return m2, dc_all[m2], GEOM.dircos2altaz(dc_all[m2])


def old_chain_roi(radec, lat, lst, roi_radius=90.0):
    """The chain this package used before ABI 0.5 (HA = LST - RA -> hadec2altaz -> altaz2dircos, through degrees, asin and atan2)."""
    altaz = GEOM.hadec2altaz(NP.stack((lst - radec[:, 0], radec[:, 1]), axis=1), lat, units='degrees')
    m2 = NP.arange(altaz.shape[0])[NP.where(altaz[:, 0] >= 90.0 - roi_radius)]
    rim = NP.where(NP.abs(altaz[:, 0] - (90.0 - roi_radius)) < 1e-12)[0]
    return m2, GEOM.altaz2dircos(altaz[m2], 'degrees'), altaz[m2], rim


def _check_dircos(dc, dc_host, altaz, what):
    """Host (geometry.frame_dircos) and device (cat_source) apply the same +, *, sqrt, / in the same order with contraction off to the
    same unit vectors and the same frame: the direction cosines are bit-identical -- by construction, not by two maths libraries
    agreeing."""
    assert dc.shape == dc_host.shape, what
    assert NP.array_equal(dc, dc_host), (what, float(NP.max(NP.abs(dc - dc_host))))


CASES = [
    # (name, sky builder, latitude, lst0, LSTs)
    ('cfg2 nside-16 diffuse', lambda: W.diffuse_sky(16, 2), -30.7224, 30.0, [30.0, 33.7, 75.123, 211.0, 359.99]),
    ('cfg4 nside-64 diffuse', lambda: W.diffuse_sky(64, 44, f_ref=185e6), -26.701, 0.0, [0.0, 0.4679, 14.5, 181.25]),
    ('cfg5 nside-256 diffuse', lambda: W.diffuse_sky(256, 55), -30.7224, 10.0, [10.0, 15.35, 100.0]),
    ('cfg3 points + diffuse', lambda: W.concat_skies(W.point_source_sky(10000, 3), W.diffuse_sky(32, 33)), -30.7224, 55.5, [55.5, 56.0, 300.0]),
]


@pytest.mark.parametrize('case', CASES, ids=[c[0] for c in CASES])
def test_roi_indices_and_dircos_bit_identical_to_the_host_statement(case):
    """Index arrays and direction cosines of the device are those of geometry.frame_dircos / roi_select bit for bit on the configs' skies,
    for the library's fall-back rotation, for the same rotation handed in as a frame, and for a full apparent-place frame (precession,
    nutation, aberration: prisim_amd/frames.py).  Against the chain used before ABI 0.5 (degrees, asin, atan2 and back) the direction
    cosines agree to that chain's own rounding: 3 ulp of an azimuth, 2 ulp / cos(alt) near the zenith; the index lists agree except for
    sources within 1e-12 degrees of the rim."""
    _, mk, lat, lst0, lsts = case
    sky = mk()
    radec = radec_catalogue(sky, lat, lst0)
    jd = 2461300.5
    with _abi.Context(0) as ctx:
        ctx.set_array(NP.array([[14.6, 0.0, 0.0]]), W.channel_grid(150e6, 1e5, 8), nt_max=1)
        ctx.set_catalog(radec, 'radec', flux_ref=sky['flux_ref'], spindex=sky['spindex'], ref_freq_hz=sky['ref_freq'], fwhm_deg=sky['fwhm_deg'])
        for roi_radius in (90.0, 60.0):
            obs = ctx.make_obs(lat, roi_radius_deg=roi_radius)
            for lst in lsts:
                m2, dc_host, altaz = host_roi(radec, lat, lst, roi_radius)
                for frame in (None, FR.snapshot_frame('radec', lst, lat, model='date')):
                    idx, dc = ctx.catalog_roi(obs, lst, ZEN, frame=frame)
                    assert idx.dtype == NP.int64 and NP.array_equal(idx, m2), (lst, roi_radius, idx.size, m2.size, frame is None)
                    _check_dircos(dc, dc_host, altaz, (case[0], lst, roi_radius, frame is None))
                # the chain used before ABI 0.5
                o2, dc_old, altaz_old, rim = old_chain_roi(radec, lat, lst, roi_radius)
                assert NP.array_equal(NP.setdiff1d(m2, rim), NP.setdiff1d(o2, rim)), (lst, roi_radius)
                keep_n, keep_o = ~NP.isin(m2, rim), ~NP.isin(o2, rim)
                if keep_o.any():
                    err = NP.max(NP.abs(dc_host[keep_n] - dc_old[keep_o]), axis=1)
                    tol = NP.maximum(2.5e-15, 4.5e-16 / NP.maximum(NP.cos(NP.radians(altaz_old[keep_o, 0])), 1e-6))
                    assert NP.all(err <= tol), (case[0], lst, float(NP.max(err / tol)))
                # apparent place of a J2000 catalogue in 2026: another sky (0.37 degrees of precession), the same bit-identity
                frame = FR.snapshot_frame('radec', lst, lat, jd=jd, epoch='J2000', model='apparent')
                idx, dc = ctx.catalog_roi(obs, lst, ZEN, frame=frame)
                m2a, dca, altaza = host_roi(radec, lat, lst, roi_radius, frame=frame)
                assert NP.array_equal(idx, m2a)
                _check_dircos(dc, dca, altaza, (case[0], lst, roi_radius, 'apparent'))
                if m2a.size == m2.size and m2.size:
                    assert float(NP.max(NP.abs(dca - dc_host))) > 1e-3      # (0.37 degrees = 6e-3 rad: not the same sky)


def test_device_unit_vectors_and_the_host_rotated_catalogue():
    """(a) unitvec='device': the device forms the unit vectors from (RA, Dec) with its own sin / cos -- the same sky to a few ulp.
    (b) VERDICT r5 next #1: a catalogue rotated on the HOST into the local frame and fed as an alt-az catalogue (identity frame) equals the
    device path given the same cel2enu / aberr_beta, bit for bit, and so do the visibilities."""
    cfg = W.config2()
    lat, lst = -30.7224, 61.25
    sky = cfg['sky']
    radec = radec_catalogue(sky, lat, 40.0)
    frame = FR.snapshot_frame('radec', lst, lat, jd=2461041.5, epoch='J2000', model='apparent')
    dc_host = GEOM.frame_dircos(GEOM.catalog_unitvec(radec, 'radec'), frame[0], frame[1])
    kw = dict(flux_ref=sky['flux_ref'], spindex=sky['spindex'], ref_freq_hz=sky['ref_freq'], fwhm_deg=sky['fwhm_deg'])
    with _abi.Context(0) as ctx, _abi.Context(0) as rot:
        for c in (ctx, rot):
            c.set_array(cfg['baselines'], cfg['channels'], nt_max=1)
        obs = ctx.make_obs(lat, beam_kind=_abi.PRISIM_BEAM_AIRY, diameter_m=14.0)
        ctx.set_catalog(radec, 'radec', **kw)
        idx, dc = ctx.catalog_roi(obs, lst, ZEN, frame=frame)
        ctx.set_catalog(radec, 'radec', unitvec='device', **kw)
        idx_d, dc_d = ctx.catalog_roi(obs, lst, ZEN, frame=frame)
        both = NP.intersect1d(idx, idx_d)
        assert both.size >= idx.size - 2 and both.size >= idx_d.size - 2            # (a source within an ulp of the horizon may differ)
        assert float(NP.max(NP.abs(dc[NP.isin(idx, both)] - dc_d[NP.isin(idx_d, both)]))) <= 1e-15
        # (b): the rotated catalogue as alt-az unit vectors, no frame
        ctx.set_catalog(radec, 'radec', **kw)
        rot.set_catalog(NP.zeros((radec.shape[0], 2)), 'altaz', unitvec=dc_host, **kw)
        n1 = ctx.set_sky_from_catalog(obs, lst, ZEN, ZEN, frame=frame)
        n2 = rot.set_sky_from_catalog(obs, 0.0, ZEN, ZEN)
        idx_r, dc_r = rot.catalog_roi(obs, 0.0, ZEN)
        assert n1 == n2 == idx.size and NP.array_equal(idx_r, idx)
        assert float(NP.max(NP.abs(dc_r - dc))) <= 1e-15                             # (normalised once more: 1 ulp)
        for prec in (_abi.PRISIM_FP64, _abi.PRISIM_FP32):
            ctx.compute(precision=prec)
            rot.compute(precision=prec)
            scale = NP.sum(NP.abs(ctx.get_pbflux()), axis=0)[None, :]
            assert float(NP.max(NP.abs(ctx.get_vis() - rot.get_vis()) / scale)) <= (1e-13 if prec == _abi.PRISIM_FP64 else 5e-7)
        # a frame that is no rotation, a velocity that is no velocity, unit vectors that are none: refused
        bad = (2.0 * frame[0], frame[1])
        with pytest.raises(ValueError):
            ctx.catalog_roi(obs, lst, ZEN, frame=bad)
        with pytest.raises(ValueError):
            ctx.catalog_roi(obs, lst, ZEN, frame=(frame[0], NP.array([0.2, 0.0, 0.0])))
        with pytest.raises(ValueError):
            ctx.set_catalog(radec, 'radec', unitvec=2.0 * dc_host, **kw)
        assert ctx.set_sky_from_catalog(obs, lst, ZEN, ZEN, frame=frame) == n1      # ... and the rejected upload left the resident sky alone


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
        m2, dc_host, altaz_h = host_roi(hadec, lat, 0.0, 90.0, coords='hadec')
        assert NP.array_equal(idx, m2)
        _check_dircos(dc, dc_host, altaz_h, 'hadec')
        old = GEOM.altaz2dircos(GEOM.hadec2altaz(hadec[m2], lat, units='degrees'), 'degrees')      # GEOM.hadec2altaz, interferometry.py:6176-6177
        assert float(NP.max(NP.abs(dc - old))) <= 1e-14
        # alt-az catalogue: mask and direction cosines straight from the positions
        ctx.set_catalog(sky['altaz'], 'altaz', flux_ref=sky['flux_ref'], spindex=sky['spindex'], ref_freq_hz=sky['ref_freq'])
        obs = ctx.make_obs(lat, roi_radius_deg=40.0)
        idx, dc = ctx.catalog_roi(obs, 0.0, ZEN)
        m2, dc_host, _ = host_roi(sky['altaz'], lat, 0.0, 40.0, coords='altaz')
        assert m2.size > 0 and NP.array_equal(idx, m2) and NP.array_equal(dc, dc_host)
        assert float(NP.max(NP.abs(dc - GEOM.altaz2dircos(sky['altaz'][m2], 'degrees')))) <= 2.5e-16
        # region of interest about the pointing centre (:6210-6213)
        pc = GEOM.altaz2dircos(NP.array([55.0, 130.0]), 'degrees').ravel()
        obs = ctx.make_obs(lat, roi_radius_deg=25.0, roi_center='pointing_center')
        idx, dc = ctx.catalog_roi(obs, 0.0, pc)
        dc_all = GEOM.frame_dircos(GEOM.catalog_unitvec(sky['altaz'], 'altaz'), NP.eye(3), NP.zeros(3))
        m2 = GEOM.roi_select(dc_all, 'pointing_center', 25.0, pc)
        assert m2.size > 0 and NP.array_equal(idx, m2)                         # s . s_pc >= cos(radius), the same operations on both sides
        ang = NP.degrees(NP.arccos(NP.clip(NP.dot(dc_all, pc), -1.0, 1.0)))      # the angle itself (:6211), away from the rim
        rim = NP.where(NP.abs(ang - 25.0) < 1e-12)[0]
        assert NP.array_equal(NP.setdiff1d(idx, rim), NP.setdiff1d(NP.where(ang <= 25.0)[0], rim))


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
    os.environ['PRISIM_HIP_BATCH_CHUNK'] = '64'                 # (the default takes up to 256 snapshots per launch: see the test below)
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
            assert tm['last_batch_snapshots'] == k - 3 * 64, tm       # (PRISIM_HIP_BATCH_CHUNK = 64 below: four chunks)
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
        del os.environ['PRISIM_HIP_BATCH_CHUNK']


def test_long_runs_take_up_to_256_snapshots_per_launch():
    """A call with more than 64 snapshots of a small array cuts chunks of up to 256 (no source splits at that length: every wave item
    writes its cube slot directly, no partial cubes, no reduction pass); slots equal single launches cut the same way."""
    bl, ch, sky = _small_array_case(171, 64)
    lat, lst0 = -30.7224, 25.0
    radec = radec_catalogue(sky, lat, lst0)
    k = 300                                                      # 256 + 44
    lsts = lst0 + 0.3 * NP.arange(k)
    with _abi.Context(0) as ctx, _abi.Context(0) as one:
        ctx.set_array(bl, ch, nt_max=k)
        one.set_array(bl, ch, nt_max=1)
        for c in (ctx, one):
            c.set_catalog(radec, 'radec', flux_ref=sky['flux_ref'], spindex=sky['spindex'], ref_freq_hz=sky['ref_freq'], fwhm_deg=sky['fwhm_deg'])
        obs = ctx.make_obs(lat, beam_kind=_abi.PRISIM_BEAM_AIRY, diameter_m=14.0)
        ctx.timing(reset=True)
        counts = ctx.observe_catalog(obs, lsts, ZEN, precision=_abi.PRISIM_FP64)
        tm = ctx.timing()
        assert tm['n_kernel'] == 2 and tm['last_batch_snapshots'] == k - 256, tm
        for t in (0, 100, 255, 256, k - 1):
            n = one.set_sky_from_catalog(obs, lsts[t], ZEN, ZEN)
            assert n == counts[t]
            one.compute(precision=_abi.PRISIM_FP64)
            scale = NP.sum(NP.abs(one.get_pbflux()), axis=0)[None, :]
            assert float(NP.max(NP.abs(ctx.get_vis(slot=t) - one.get_vis()) / scale)) <= 1e-13, t


@pytest.mark.parametrize('nbl,nchan,k', [(171, 256, 64), (3, 64, 9), (65, 96, 7), (256, 128, 12), (17, 40, 1), (171, 64, 300)])
def test_batched_gradient_snapshots_equal_single_launches(nbl, nchan, k):
    """visibilities AND baseline gradients (interferometry.py:6330, 6338, 6343) of K LSTs of a small array in one launch
    (k_skyvis_grad_taper_f64_batch: wave items of 16 baselines x 4 sources over the chunk's snapshot table).  A chunk whose sources are not
    split walks every snapshot's sources in the single launch's order, four at a time from a multiple of four: bit-identical; split
    sources (few snapshots: the grid is filled by cutting the sources) are the same sums to rounding, 1e-13 of sum |beam x flux|."""
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
        ctx.timing(reset=True)
        counts = ctx.observe_catalog(obs, lsts, ZEN, precision=_abi.PRISIM_FP64, want_grad=True)
        tm = ctx.timing()
        assert tm['last_batch_snapshots'] == (k if k <= 256 else k - 256) and tm['n_kernel'] == (1 if k <= 256 else 2), tm
        assert tm['last_chan_tile'] == 32
        exact = tm['last_nsplit'] == 1
        picks = range(k) if k <= 64 else (0, 1, 100, 255, 256, k - 1)
        for t in picks:
            n = one.set_sky_from_catalog(obs, lsts[t], ZEN, ZEN)
            assert n == counts[t]
            one.compute(precision=_abi.PRISIM_FP64, want_grad=True)
            v1, g1 = one.get_vis(want_grad=True)
            vb, gb = ctx.get_vis(slot=t, want_grad=True)
            if exact:
                assert NP.array_equal(vb, v1) and NP.array_equal(gb, g1), (t, float(NP.max(NP.abs(gb - g1))))
            else:
                scale = NP.sum(NP.abs(one.get_pbflux()), axis=0)[None, :]
                assert float(NP.max(NP.abs(vb - v1) / scale)) <= 1e-13, t
                assert float(NP.max(NP.abs(gb - g1) / scale[None])) <= 1e-13, t
        # the visibilities of the gradient launch are those of the plain batched launch to rounding (other kernel, other summation order)
        plain = ctx.observe_catalog(obs, lsts[:min(k, 8)], ZEN, precision=_abi.PRISIM_FP64)
        assert NP.array_equal(plain, counts[:min(k, 8)])
        one.set_sky_from_catalog(obs, lsts[0], ZEN, ZEN)
        scale = NP.sum(NP.abs(one.get_pbflux()), axis=0)[None, :]
        one.compute(precision=_abi.PRISIM_FP64, want_grad=True)
        assert float(NP.max(NP.abs(ctx.get_vis(slot=0) - one.get_vis()) / scale)) <= 1e-13


def test_batched_gradient_against_the_oracle_with_an_empty_snapshot():
    """The batched gradient launch against the CPU oracle's four sums (tolerance 1e-11 of sum |beam x flux|, the fp64 bar), on a run whose
    middle snapshot looks at a patch of sky holding no catalogue source (zeros, and nothing read through the empty snapshot's rows)."""
    from oracle import skyvis_oracle as O, beams_oracle as BO
    bl, ch, sky = _small_array_case(40, 48)
    lat, lst0 = -30.7224, 25.0
    radec = radec_catalogue(sky, lat, lst0)
    keep = NP.where(NP.abs(((radec[:, 0] - lst0 + 180.0) % 360.0) - 180.0) < 60.0)[0]      # a catalogue around RA = lst0 only
    radec = radec[keep]
    lsts = NP.array([lst0, lst0 + 180.0, lst0 + 5.0])                                     # the second LST sees none of it
    with _abi.Context(0) as ctx:
        ctx.set_array(bl, ch, nt_max=3)
        ctx.set_catalog(radec, 'radec', flux_ref=sky['flux_ref'][keep], spindex=sky['spindex'][keep], ref_freq_hz=sky['ref_freq'],
                        fwhm_deg=sky['fwhm_deg'][keep])
        obs = ctx.make_obs(lat, beam_kind=_abi.PRISIM_BEAM_GAUSSIAN, diameter_m=14.0, roi_radius_deg=25.0)
        counts = ctx.observe_catalog(obs, lsts, ZEN, precision=_abi.PRISIM_FP64, want_grad=True)
        assert ctx.timing()['last_batch_snapshots'] == 3
        assert counts[1] == 0 and counts[0] > 0 and counts[2] > 0
        for t, lst in enumerate(lsts):
            m2, dc, altaz = host_roi(radec, lat, lst, roi_radius=25.0)
            assert m2.size == counts[t]
            v, g = ctx.get_vis(slot=t, want_grad=True)
            if m2.size == 0:
                assert not v.any() and not g.any()
                continue
            pb = BO.gaussian_beam(14.0, altaz, ch, power=True) * sky['flux_ref'][keep][m2, None] \
                * (ch[None, :] / sky['ref_freq']) ** sky['spindex'][keep][m2, None]
            ref, gref = O.skyvis(bl, ch, dc, pb, ZEN, fwhm_deg=sky['fwhm_deg'][keep][m2], gradient=True)
            scale = O.abs_flux_sum(pb)[None, :]
            assert float(NP.max(NP.abs(v - ref) / scale)) <= 1e-11
            assert float(NP.max(NP.abs(g - gref) / scale[None])) <= 1e-11


def test_fp32_requests_on_a_small_array_are_served_by_the_fp64_batch():
    """PRISim's memsave (fp32 arithmetic, interferometry.py:6320-6343) on an array this small: the cost of a snapshot is its launches, so the
    request is served by the batched fp64 launch -- the fp64 result, bit for bit (well inside the fp32 tolerance 5e-6);
    PRISIM_HIP_BATCH_FP32_AS_FP64=0 keeps the per-snapshot fp32 chain (the A/B)."""
    bl, ch, sky = _small_array_case(171, 96)
    lat, lst0 = -30.7224, 25.0
    radec = radec_catalogue(sky, lat, lst0)
    k = 10
    lsts = lst0 + 0.75 * NP.arange(k)
    with _abi.Context(0) as ctx:
        ctx.set_array(bl, ch, nt_max=k)
        ctx.set_catalog(radec, 'radec', flux_ref=sky['flux_ref'], spindex=sky['spindex'], ref_freq_hz=sky['ref_freq'], fwhm_deg=sky['fwhm_deg'])
        obs = ctx.make_obs(lat, beam_kind=_abi.PRISIM_BEAM_AIRY, diameter_m=14.0)
        ctx.observe_catalog(obs, lsts, ZEN, precision=_abi.PRISIM_FP64)
        v64 = [ctx.get_vis(slot=t) for t in range(k)]
        ctx.observe_catalog(obs, lsts, ZEN, precision=_abi.PRISIM_FP32)
        assert ctx.timing()['last_batch_snapshots'] == k
        for t in range(k):
            assert NP.array_equal(ctx.get_vis(slot=t), v64[t])
        os.environ['PRISIM_HIP_BATCH_FP32_AS_FP64'] = '0'
        try:
            ctx.observe_catalog(obs, lsts, ZEN, precision=_abi.PRISIM_FP32)
            assert ctx.timing()['last_batch_snapshots'] == 1
            for t in (0, k - 1):
                ctx.set_sky_from_catalog(obs, lsts[t], ZEN, ZEN)
                scale = NP.sum(NP.abs(ctx.get_pbflux()), axis=0)[None, :]
                err = float(NP.max(NP.abs(ctx.get_vis(slot=t) - v64[t]) / scale))
                assert 0.0 < err <= 5e-6, err
        finally:
            del os.environ['PRISIM_HIP_BATCH_FP32_AS_FP64']


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


def test_class_surface_gradient_and_memsave_runs_of_a_small_array_share_launches():
    """observe_batch / observe() with gradient_mode='baseline' and with memsave=True on HERA-19 (interferometry.py:6320-6343, all three modes):
    the whole run in one launch; visibilities and gradient cubes equal those of the host-formed sky path (PRISIM_CATALOG=0: per-snapshot
    kernels on uploaded skies) to 1e-12 of the largest visibility in fp64 and 5e-6 of it for memsave (complex64 storage, fp64 arithmetic)."""
    from prisim_amd import interferometry as RI, skymodel as SM
    cfg = W.config2()
    lat = -30.7224
    sky = cfg['sky']
    radec = radec_catalogue(sky, lat, 15.0)
    n = radec.shape[0]
    tel = {'id': 'hera', 'shape': 'dish', 'size': 14.0, 'ocoords': 'altaz', 'orientation': NP.array([[90.0, 270.0]]), 'groundplane': None}
    k = 12
    times = [(2457000.5 + j * 1e-3, 15.0 + 0.5 * j) for j in range(k)]

    def run(mode, **kw):
        skymod = SM.SkyModel(location=radec, flux_ref=sky['flux_ref'], spindex=sky['spindex'], ref_freq=sky['ref_freq'],
                             src_shape=NP.stack((sky['fwhm_deg'], sky['fwhm_deg'], NP.zeros(n)), axis=1), epoch=None)
        ia = RI.InterferometerArray(['b%d' % i for i in range(cfg['baselines'].shape[0])], cfg['baselines'], cfg['channels'], telescope=tel,
                                    latitude=lat, skycoords='radec', pointing_coords='hadec')
        if mode == 'batch':
            ia.observe_batch(times, {'Tnet': 100.0}, NP.ones(cfg['channels'].size), [0.0, lat], skymod, 10.0, **kw)
        else:
            for t in times:
                ia.observe(t, {'Tnet': 100.0}, NP.ones(cfg['channels'].size), [0.0, lat], skymod, 10.0, **kw)
        tm = ia._ctx.timing() if ia._ctx is not None else {}
        vis = NP.array(ia.skyvis_freq)
        grad = NP.array(ia.gradient['baseline']) if kw.get('gradient_mode') else None
        ia.close()
        return vis, grad, tm

    os.environ['PRISIM_CATALOG'] = '0'
    try:
        v0, g0, _ = run('observe', gradient_mode='baseline')
    finally:
        del os.environ['PRISIM_CATALOG']
    scale = float(NP.max(NP.abs(v0)))
    vb, gb, tmb = run('batch', gradient_mode='baseline')
    assert tmb['last_batch_snapshots'] == k and vb.shape == (171, 256, k) and gb.shape == g0.shape
    assert float(NP.max(NP.abs(vb - v0))) <= 1e-12 * scale and float(NP.max(NP.abs(gb - g0))) <= 1e-12 * scale
    vo, go, tmo = run('observe', gradient_mode='baseline')
    assert tmo['last_batch_snapshots'] == 1 and tmo['last_chan_tile'] == 32           # a chunk of one through the batched gradient launch
    assert float(NP.max(NP.abs(vo - v0))) <= 1e-12 * scale and float(NP.max(NP.abs(go - g0))) <= 1e-12 * scale
    vm, _, tmm = run('batch', memsave=True)
    assert tmm['last_batch_snapshots'] == k and vm.dtype == NP.complex64
    assert float(NP.max(NP.abs(vm - v0))) <= 5e-6 * scale


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
    m2, dc, altaz = host_roi(radec, lat, 72.5, frame=FR.snapshot_frame('radec', 72.5, lat, jd=2457000.5, epoch='J2000', model='apparent'))
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


# ---- the snapshot's frame through the class surface: precession + nutation + aberration, one function for host and device ----
def test_observe_radec_sky_in_the_apparent_frame_of_the_snapshot():
    """InterferometerArray.observe() on a J2000 catalogue in 2026 (interferometry.py:6174-6180: FK5(epoch) -> FK5(obstime) -> AltAz): the
    device applies the frame of prisim_amd/frames.py; region of interest and visibilities equal the oracle on the host statement of the
    same frame; the per-snapshot upload path (PRISIM_CATALOG=0) selects the same sources; 'date' (HA = LST - RA on the J2000 positions) is
    a visibly different sky; frame_provider replaces the built-in model; an in-place edit of the sky model is seen."""
    from oracle import skyvis_oracle as O, beams_oracle as BO
    from prisim_amd import interferometry as RI, skymodel as SM
    rng = NP.random.default_rng(77)
    lat, lst, jd = -30.7224, 47.0, 2461041.5                 # 2026 Jan 1
    nsrc = 600
    ra = rng.uniform(0, 360, nsrc)
    dec = NP.degrees(NP.arcsin(rng.uniform(-1, 0.6, nsrc)))
    skymod = SM.SkyModel(location=NP.stack((ra, dec), 1), flux_ref=rng.uniform(1, 5, nsrc), spindex=rng.uniform(-1, 0, nsrc), ref_freq=150e6,
                         epoch='J2000')
    ch = 150e6 + (NP.arange(24) - 12) * 2e5
    bl = rng.uniform(-300, 300, (11, 3)); bl[:, 2] = 0

    def make(model=None, provider=None):
        ia = RI.InterferometerArray(list(range(11)), bl, ch, telescope={'shape': 'dish', 'size': 14.0}, latitude=lat, skycoords='radec',
                                    pointing_coords='altaz')
        if model is not None:
            ia.frame_model = model
        ia.frame_provider = provider
        ia.observe((jd, lst), {'Tnet': 100.0}, NP.ones(24), [90.0, 270.0], skymod, 10.0)
        return ia

    def expect(model):
        frame = FR.snapshot_frame('radec', lst, lat, jd=jd, epoch='J2000', model=model)
        dc = GEOM.frame_dircos(GEOM.catalog_unitvec(skymod.location, 'radec'), frame[0], frame[1])
        sel = GEOM.roi_select(dc, 'zenith', 90.0)
        altaz = GEOM.dircos2altaz(dc[sel])
        pb = BO.airy_disk_pattern(14.0, altaz, ch, pointing_altaz=NP.array([90.0, 270.0])) * skymod.generate_spectrum(ind=sel, frequency=ch)
        return sel, O.skyvis(bl, ch, dc[sel], pb, ZEN), O.abs_flux_sum(pb)[None, :]

    ia = make()
    assert ia.frame_model == 'apparent'
    sel, ref, scale = expect('apparent')
    assert NP.array_equal(NP.asarray(ia.obs_catalog_indices[0]), sel) and 0 < sel.size < nsrc
    assert float(NP.max(NP.abs(ia.skyvis_freq[:, :, 0] - ref) / scale)) <= 1e-11
    # the sky formed on the host and uploaded: the same function, the same sources, the same visibilities
    os.environ['PRISIM_CATALOG'] = '0'
    try:
        up = make()
    finally:
        del os.environ['PRISIM_CATALOG']
    assert NP.array_equal(NP.asarray(up.obs_catalog_indices[0]), sel)
    assert float(NP.max(NP.abs(up.skyvis_freq[:, :, 0] - ia.skyvis_freq[:, :, 0]) / scale)) <= 1e-13
    # 26 years of precession are 0.36 degrees: on 300 m baselines at 150 MHz a different set of visibilities
    old = make('date')
    sel_d, ref_d, scale_d = expect('date')
    assert float(NP.max(NP.abs(old.skyvis_freq[:, :, 0] - ref_d) / scale_d)) <= 1e-11
    assert float(NP.max(NP.abs(old.skyvis_freq[:, :, 0] - ia.skyvis_freq[:, :, 0]) / scale)) > 1e-2
    mean = make('mean')
    d_mean = float(NP.max(NP.abs(mean.skyvis_freq[:, :, 0] - ia.skyvis_freq[:, :, 0]) / scale))
    assert 1e-4 < d_mean < 0.2                               # nutation + aberration: tens of arcseconds (0.1 rad of phase on 300 m)
    # a caller-supplied frame (filled from astropy on the PRISim side) replaces the built-in model
    calls = []

    def provider(jd_, lst_, skymodel_):
        calls.append((jd_, lst_))
        return FR.snapshot_frame('radec', lst_, lat, jd=jd_, epoch=skymodel_.epoch, model='mean')
    prov = make(provider=provider)
    assert calls == [(jd, lst)]
    assert NP.array_equal(prov.skyvis_freq, mean.skyvis_freq)
    # the resident catalogue is re-uploaded when the sky model is edited in place
    before = ia.skyvis_freq[:, :, 0].copy()
    skymod.flux_ref *= 2.0
    try:
        ia.observe((jd, lst), {'Tnet': 100.0}, NP.ones(24), [90.0, 270.0], skymod, 10.0)
        assert float(NP.max(NP.abs(ia.skyvis_freq[:, :, 1] - 2.0 * before) / scale)) <= 1e-12
        skymod.src_shape = None
        skymod.location[:, 0] += 1.0                      # ... and a moved sky is another sky
        ia.observe((jd, lst), {'Tnet': 100.0}, NP.ones(24), [90.0, 270.0], skymod, 10.0)
        assert float(NP.max(NP.abs(ia.skyvis_freq[:, :, 2] - 2.0 * before) / scale)) > 1e-3
    finally:
        skymod.flux_ref /= 2.0


def test_observe_batch_failure_leaves_the_instance_aligned():
    """a batch that fails (here: a bandpass of the wrong length in the third snapshot) leaves timestamp, pointing centres,
    bandpass and Tsys layers as they were; a later observe() lines up with n_acc."""
    from prisim_amd import interferometry as RI, skymodel as SM
    cfg = W.config1()
    sky = cfg['sky']
    lat = -30.7224
    radec = radec_catalogue(sky, lat, 40.0)
    skymod = SM.SkyModel(location=radec, flux_ref=sky['flux_ref'], spindex=sky['spindex'], ref_freq=sky['ref_freq'])
    nchan = cfg['channels'].size
    ia = RI.InterferometerArray(['a', 'b', 'c'], cfg['baselines'], cfg['channels'], telescope={'shape': 'gaussian', 'size': cfg['diameter']},
                                latitude=lat, skycoords='radec', pointing_coords='hadec')
    ia.reserve(6)
    ia.observe((2457000.5, 40.0), {'Tnet': 100.0}, NP.ones(nchan), [0.0, lat], skymod, 10.0)
    times = [(2457000.5 + 0.001 * t, 40.0 + t) for t in range(1, 4)]
    with pytest.raises(ValueError):
        ia.observe_batch(times, {'Tnet': 100.0}, [NP.ones(nchan), NP.ones(nchan), NP.ones(nchan + 1)], [0.0, lat], skymod, 10.0)
    assert ia.n_acc == 1 and len(ia.timestamp) == 1 and ia.pointing_center.shape == (1, 2) and len(ia.Tsysinfo) == 1
    assert ia.bp.shape[2:] in ((), (1,)) and len(ia.lst) == 1
    ia.observe_batch(times, {'Tnet': 100.0}, NP.ones(nchan), [0.0, lat], skymod, 10.0)
    assert ia.n_acc == 4 and len(ia.timestamp) == 4 and ia.pointing_center.shape == (4, 2) and ia.bp.shape == (3, nchan, 4)
    assert ia.skyvis_freq.shape == (3, nchan, 4) and ia.Tsys.shape == (3, nchan, 4)


def test_external_beam_at_a_source_due_north_to_the_last_bit():
    """A source whose East direction cosine is -1e-17 has azimuth 2 pi - 1e-17, which rounds to 2 pi itself: the HEALPix ring
    interpolation must treat it as azimuth 0 (found when the snapshot frame replaced the asin / atan2 chain: 11 pixels of config 4's
    nside-64 sky sit due North and their beam values came from one pixel past the ring -- 4e-4 of the peak)."""
    from oracle import healpix_oracle as H
    from prisim_amd import primary_beams as PB
    cfg = W.config4(n_acc=1)
    ch = cfg['channels'][:16]
    m = PB.spectral_interp_matrix(cfg['beam_freqs'], ch, kind='cubic', chromatic=True, select_freq=None)
    alts = NP.radians(NP.array([5.379379, 18.839405, 32.089951, 37.921651, 60.0, 85.0]))
    variants = []
    for east in (0.0, -1e-17, 1e-17):
        dc = NP.stack((NP.full(alts.size, east), NP.cos(alts), NP.sin(alts)), axis=1)
        with _abi.Context(0) as ctx:
            ctx.set_array(cfg['baselines'][:4], ch, nt_max=1)
            ctx.set_external_beam(cfg['beam_table'], m)
            ctx.set_sky_external(dc, NP.ones((alts.size, ch.size)), ZEN)
            variants.append(ctx.get_pbflux())
    scale = NP.max(NP.abs(variants[0]))
    assert float(NP.max(NP.abs(variants[1] - variants[0])) / scale) <= 1e-7          # (float32 beam storage: one rounding may flip)
    assert float(NP.max(NP.abs(variants[2] - variants[0])) / scale) <= 1e-7
    # the checker has the same property
    for ph in (-1e-17, 2 * NP.pi - 1e-17):
        a = H.get_interp_val(NP.log10(cfg['beam_table'][:, 0]), NP.pi / 2 - alts, NP.full(alts.size, ph))
        b = H.get_interp_val(NP.log10(cfg['beam_table'][:, 0]), NP.pi / 2 - alts, NP.zeros(alts.size))
        assert float(NP.max(NP.abs(a - b))) <= 1e-12


def test_device_external_beam_normalisation_against_the_reference_statements():
    """scripts/run_prisim.py:2099-2103 + interferometry.py:4466 on the device, against the golden vector of those statements executed
    (tests/golden/golden_aux.npz): a nside-2 beam table holds 10 ** logbeam_in at 23 pixels and the sources sit exactly on those pixel
    centres, where the HEALPix interpolation returns the pixel itself; unit fluxes.  The one NaN of the vector is replaced by a low finite
    value (the API wants a positive table) -- the reference's nanmax skipped it, so every other number is unchanged.  Tolerance: 1.5
    float32 ulp (log10(10 ** x) rounds x in the last place before the float32 store)."""
    from conftest import GOLDEN
    g = NP.load(os.path.join(GOLDEN, 'golden_aux.npz'))
    logb = g['logbeam_in'].copy()
    want = g['pbeam_f32'].astype(NP.float64)
    nsrc, nchan = logb.shape
    nanpos = NP.argwhere(NP.isnan(logb))
    assert nanpos.shape == (1, 2)
    logb[tuple(nanpos[0])] = -6.0
    nside = 2
    theta, phi = GEOM.healpix_pix2ang_ring(nside)
    pix = NP.arange(nsrc)                                   # rings 1-3 (20 pixels) and three of the equatorial ring: altitude >= 0
    table = NP.full((12 * nside * nside, nchan), 1e-9)
    table[pix] = 10.0 ** logb
    st = NP.sin(theta[pix])
    dc = NP.stack((st * NP.sin(phi[pix]), st * NP.cos(phi[pix]), NP.cos(theta[pix])), axis=1)      # azimuth from North through East
    ch = 150e6 + 1e5 * NP.arange(nchan)
    with _abi.Context(0) as ctx:
        ctx.set_array(NP.array([[14.6, 0.0, 0.0]]), ch, nt_max=1)
        ctx.set_external_beam(table, NP.eye(nchan))
        ctx.set_sky_external(dc, NP.ones((nsrc, nchan)), ZEN)
        pb = ctx.get_pbflux()
    keep = NP.ones(logb.shape, dtype=bool)
    keep[tuple(nanpos[0])] = False
    assert float(NP.max(NP.abs(pb[keep] / want[keep] - 1.0))) <= 1.5 * 2.0 ** -23
    assert NP.max(pb[:, 3]) < 1.0 and abs(NP.max(NP.delete(pb, 3, axis=1)) - 1.0) <= 2.0 ** -23     # the clamp of :2100 on the device
