"""CPU: host-side logic -- layouts, channel grid, HEALPix, geometry, workloads, sharding, sky model,
beam dispatch, argument validation of the reference-mirroring interface."""
import os
import numpy as NP
import pytest

from prisim_amd import geometry as GEOM, layouts as LAY, workloads as W, skymodel as SM, primary_beams as PB
from prisim_amd import baseline_delay_horizon as DLY, _abi
import bench


def test_hexagon_generator_hera19():
    xy, labels = LAY.hexagon_generator(14.6, n_total=19)
    assert xy.shape == (19, 2) and labels == [str(i) for i in range(19)]
    assert NP.allclose(xy.mean(axis=0), 0.0, atol=1e-12)
    d = NP.sqrt(((xy[:, None, :] - xy[None, :, :]) ** 2).sum(-1))
    d[NP.diag_indices(19)] = NP.inf
    assert NP.allclose(d.min(axis=1), 14.6)              # nearest neighbour = spacing
    assert NP.isclose(d[NP.isfinite(d)].max(), 4 * 14.6)  # corner to corner of a side-3 hexagon
    with pytest.raises(ValueError):
        LAY.hexagon_generator(14.6, n_total=20)
    with pytest.raises(ValueError):
        LAY.hexagon_generator(14.6, n_total=19, n_side=3)
    with pytest.raises(NameError):
        LAY.hexagon_generator(14.6)
    with pytest.raises(TypeError):
        LAY.hexagon_generator('a', n_total=19)
    xy2, _ = LAY.hexagon_generator(14.6, n_side=3)
    assert NP.array_equal(xy, xy2)


def test_baselines_fold_and_sort():
    bl, ids = LAY.layout_baselines('HERA-19')
    assert bl.shape == (171, 3) and ids.shape == (171, 2)
    length = NP.sqrt((bl ** 2).sum(1))
    assert NP.all(NP.diff(length) >= -1e-12)
    ang = NP.degrees(NP.angle(bl[:, 0] + 1j * bl[:, 1]))
    assert NP.all(ang >= -67.5 - 1e-9) and NP.all(ang <= 112.5 + 1e-9)
    pos = LAY.array_layout('HERA-19')
    assert NP.allclose(pos[ids[:, 0]] - pos[ids[:, 1]], bl)    # ids stay consistent with the fold
    assert NP.isclose(length[0], 14.6)
    assert LAY.layout_baselines('HERA-350')[0].shape == (61075, 3)
    bla, _ = LAY.baseline_generator(pos, auto=True)
    assert bla.shape[0] == 171 + 19
    blc, _ = LAY.baseline_generator(pos, conjugate=True)
    assert blc.shape[0] == 2 * 171


def test_channel_grid_centre():
    ch = W.channel_grid(150e6, 390625.0, 256)
    assert ch[128] == 150e6 and NP.allclose(NP.diff(ch), 390625.0)


def test_healpix_ring_pixel_centres():
    th, ph = GEOM.healpix_pix2ang_ring(1)
    assert NP.allclose(th[:4], NP.arccos(2 / 3.0)) and NP.allclose(ph[:4], NP.pi / 4 + NP.arange(4) * NP.pi / 2)
    assert NP.allclose(th[4:8], NP.pi / 2) and NP.allclose(th[8:], NP.pi - NP.arccos(2 / 3.0))
    for nside in (2, 8, 32):
        th, ph = GEOM.healpix_pix2ang_ring(nside)
        assert th.size == 12 * nside * nside
        assert NP.all(NP.diff(th) >= -1e-15)                   # RING order: colatitude never decreases
        v = NP.stack((NP.sin(th) * NP.cos(ph), NP.sin(th) * NP.sin(ph), NP.cos(th)), 1)
        assert NP.abs(v.mean(0)).max() < 1e-12                 # equal-area pixels: centroid at the origin
        assert NP.allclose(th + th[::-1], NP.pi)               # north/south mirror symmetry
        assert NP.isclose(NP.cos(th[0]), 1 - 1.0 / (3 * nside ** 2))
    assert NP.isclose(GEOM.nside2resol(16), NP.sqrt(4 * NP.pi / 3072))
    with pytest.raises(ValueError):
        GEOM.healpix_pix2ang_ring(4, ipix=[192])


def test_geometry_conventions():
    assert NP.allclose(GEOM.altaz2dircos([0.0, 90.0]), [[1, 0, 0]], atol=1e-15)      # East
    assert NP.allclose(GEOM.altaz2dircos([0.0, 0.0]), [[0, 1, 0]], atol=1e-15)       # North
    assert NP.allclose(GEOM.altaz2dircos([90.0, 270.0]), [[0, 0, 1]], atol=1e-15)    # zenith
    aa = NP.array([[10.0, 20.0], [80.0, 200.0], [45.0, 359.0]])
    assert NP.allclose(GEOM.dircos2altaz(GEOM.altaz2dircos(aa)), aa)
    # a source on the meridian (HA=0) at Dec = latitude is at the zenith
    assert NP.isclose(GEOM.hadec2altaz([0.0, -30.72], -30.72)[0], 90.0)
    hd = NP.array([[10.0, -30.0], [-60.0, 20.0]])
    assert NP.allclose(GEOM.altaz2hadec(GEOM.hadec2altaz(hd, -30.7), -30.7), hd)
    with pytest.raises(ValueError):
        GEOM.altaz2dircos([1.0, 2.0, 3.0])


def test_geometric_delay_api():
    bl = NP.array([[100.0, 0.0, 0.0], [0.0, 50.0, 0.0]])
    tau = DLY.geometric_delay(bl, NP.array([[0.0, 90.0]]), altaz=True, hadec=False)
    assert tau.shape == (1, 2) and NP.isclose(tau[0, 0], 100.0 / 299792458.0) and abs(tau[0, 1]) < 1e-20
    with pytest.raises(ValueError):
        DLY.geometric_delay(bl, NP.zeros((1, 2)), altaz=True, hadec=True)
    with pytest.raises(ValueError):
        DLY.geometric_delay(bl, NP.zeros((1, 2)))            # hadec without latitude
    with pytest.raises(TypeError):
        DLY.geometric_delay([[1, 2, 3]], NP.zeros((1, 3)), dircos=True, hadec=False)


def test_workload_shapes():
    c1, c2 = W.config1(), W.config2()
    assert c1['baselines'].shape == (3, 3) and c1['channels'].size == 64 and c1['sky']['dircos'].shape == (100, 3)
    assert c2['baselines'].shape == (171, 3) and c2['channels'].size == 256
    n2 = c2['sky']['dircos'].shape[0]
    assert n2 == 1504 and NP.all(c2['sky']['dircos'][:, 2] > 0)
    assert NP.allclose(c2['sky']['fwhm_deg'], NP.degrees(NP.sqrt(4 * NP.pi / 3072)))
    assert NP.allclose(NP.linalg.norm(c1['sky']['dircos'], axis=1), 1.0)
    sub = W.subsample(c2, bl_stride=10, ch_count=16, src_stride=7)
    assert sub['baselines'].shape[0] == 18 and sub['channels'].size == 16 and sub['sky']['flux_ref'].size == 215


@pytest.mark.parametrize('nbl,world', [(61075, 1), (61075, 2), (61075, 3), (61075, 8), (8128, 8), (171, 2), (171, 4), (3, 8), (1, 2)])
def test_baseline_sharding_covers_everything_once(nbl, world):
    """prisim_amd.sharding: groups of baselines dealt round-robin, equal padded shard sizes, every baseline exactly once, and the
    gathered rank-major cube goes back into the global order."""
    from prisim_amd import sharding as SH
    bl = NP.arange(nbl * 3, dtype=float).reshape(nbl, 3)
    per = SH.shard_size(nbl, world)
    group = SH.group_size(nbl, world)
    assert 1 <= group <= 256 and (nbl < 1024 * world or group == 256)
    seen, gathered = [], []
    for rank in range(world):
        mine, n_real = bench.shard_baselines(bl, world, rank)
        rows, idx, n2 = SH.shard_rows(bl, world, rank)
        assert mine.shape == (per, 3) and n_real == n2 == idx.size and NP.array_equal(mine, rows)
        assert NP.array_equal(mine[:n_real], bl[idx]) and NP.all(NP.diff(idx) > 0)
        assert NP.all(mine[n_real:] == bl[-1])                                  # padding repeats the last baseline
        assert per - n_real <= group                                            # shares differ by at most one group
        seen.append(idx)
        gathered.append(mine * 10.0 + rank)
    assert NP.array_equal(NP.sort(NP.concatenate(seen)), NP.arange(nbl))
    back = SH.unshard_rows(NP.concatenate(gathered), nbl, world)
    owner = NP.empty(nbl)
    for rank, idx in enumerate(seen):
        owner[idx] = rank
    assert NP.array_equal(back, bl * 10.0 + owner[:, None])
    with pytest.raises(ValueError):
        SH.unshard_rows(NP.zeros((world * per + 1, 3)), nbl, world)


def test_round_robin_shards_spread_the_long_baselines():
    """The headline array lists baselines by length; contiguous shards would give the last of 8 ranks every group that cannot use the
    lifting rotation, the dealt-out ones give each rank 1 or 2 of them."""
    from prisim_amd import sharding as SH
    cfg = W.config3(nsrc=10)
    length = NP.sqrt(NP.sum(cfg['baselines'] ** 2, axis=1))
    assert NP.all(NP.diff(length) >= -1e-9)
    long_ones = length > 298.5
    per_rank = [int(NP.sum(long_ones[SH.shard_index(length.size, 8, r)])) for r in range(8)]
    assert max(per_rank) - min(per_rank) <= 256 and min(per_rank) > 0
    lo = 7 * ((length.size + 7) // 8)
    assert int(NP.sum(long_ones[lo:])) == int(NP.sum(long_ones))               # contiguous: all of them on the last rank


def test_skymodel_spectra():
    loc = NP.array([[10.0, 20.0], [30.0, 40.0], [50.0, 60.0]])
    sm = SM.SkyModel(location=loc, flux_ref=[1.0, 2.0, 3.0], spindex=-0.8, ref_freq=150e6)
    f = NP.array([100e6, 150e6, 200e6])
    sp = sm.generate_spectrum(ind=[2, 0], frequency=f, interp_method='pchip')
    assert sp.shape == (2, 3) and NP.allclose(sp[:, 1], [3.0, 1.0]) and NP.isclose(sp[0, 0], 3.0 * (100 / 150.0) ** -0.8)
    tab = SM.SkyModel(location=loc, frequency=f, spectrum=NP.arange(9.0).reshape(3, 3))
    assert NP.allclose(tab.generate_spectrum(ind=[1], frequency=f), [[3.0, 4.0, 5.0]])
    assert NP.allclose(tab.generate_spectrum(ind=[1], frequency=[125e6]), [[3.5]])
    sub = sm.subset([1])
    assert sub.location.shape == (1, 2) and sub.flux_ref[0] == 2.0


def test_device_beam_spec_dispatch():
    k, d, p, x = PB.device_beam_spec({'id': 'hera', 'orientation': [90.0, 270.0], 'ocoords': 'altaz'})
    assert k == _abi.PRISIM_BEAM_AIRY and d == 14.0 and NP.allclose(p, [0, 0, 1], atol=1e-12) and x is None
    assert PB.device_beam_spec({'id': 'hirax'})[1] == 6.0
    assert PB.device_beam_spec({'shape': 'gaussian', 'size': 5.0})[:2] == (_abi.PRISIM_BEAM_GAUSSIAN, 5.0)
    assert PB.device_beam_spec({'shape': 'dish', 'size': 25.0})[0] == _abi.PRISIM_BEAM_AIRY
    assert PB.device_beam_spec({})[0] == _abi.PRISIM_BEAM_DELTA
    k, d, p, x = PB.device_beam_spec({'shape': 'dish', 'size': 14.0}, pointing_center=[0.0, 90.0])
    assert NP.allclose(p, [1, 0, 0], atol=1e-12)
    k, d, p, x = PB.device_beam_spec({'id': 'mwa'}, east2ax1=15.0)
    assert k == _abi.PRISIM_BEAM_DIPOLE and d == 0.74 and x['array']['nax1'] == 4 and x['array']['sep1'] == 1.1
    assert x['array']['east2ax1'] == 15.0 and NP.allclose(x['dipole_dircos'], [1, 0, 0])
    assert PB.device_beam_spec({'id': 'paper'})[1] == 2.0 and PB.device_beam_spec({'id': 'mwa_dipole'})[1] == 0.74
    k, d, p, x = PB.device_beam_spec({'shape': 'dipole', 'size': 1.5, 'ocoords': 'altaz', 'orientation': [0.0, 0.0],
                                      'groundplane': 0.3, 'ground_modify': {'scale': 2.0, 'max': 3.0}}, half_wave_dipole_approx=True)
    assert k == _abi.PRISIM_BEAM_DIPOLE and NP.allclose(x['dipole_dircos'], [0, 1, 0], atol=1e-12)
    assert x['dipole_mode'] == _abi.PRISIM_DIPOLE_HALFWAVE and x['ground']['height'] == 0.3
    ext = _abi.make_beam_ext(x)
    assert ext.ground_modify == 7 and ext.ground_scale == 2.0 and ext.ground_max == 3.0 and ext.array_nax1 == 0
    # a dish ignores the ground plane (primary_beams.py:422)
    assert PB.device_beam_spec({'shape': 'dish', 'size': 14.0, 'groundplane': 0.3})[3] is None
    with pytest.raises(TypeError):
        PB.device_beam_spec(None)
    with pytest.raises(ValueError):
        PB.device_beam_spec({'shape': 'banana'})
    with pytest.raises(ValueError):
        PB.device_beam_spec({'id': 'paper'}, short_dipole_approx=True, half_wave_dipole_approx=True)
    with pytest.raises(KeyError):
        PB.device_beam_spec({'id': 'paper', 'orientation': [0.0, 90.0]})
    with pytest.raises(ValueError):                                              # the band is chosen from the first frequency
        PB.device_beam_spec({'id': 'vla'})
    kind, size, bpc, ext = PB.device_beam_spec({'id': 'vla'}, first_frequency_hz=1.4e9)
    assert kind == _abi.PRISIM_BEAM_POLY and NP.allclose(ext['poly'], [-1.343, 6.579, -1.186, 0.0])      # 1.465 GHz band (:494)
    kind, size, bpc, ext = PB.device_beam_spec({'id': 'ugmrt'}, first_frequency_hz=0.61e9)
    assert NP.allclose(ext['poly'], [-3.190, 38.642, -20.471, 3.964])
    assert NP.all(NP.isnan(PB.poly_beam_coefficients('ugmrt', 0.2e9)))           # no uGMRT polynomial at 235 MHz (:786)
    with pytest.raises(KeyError):
        PB.device_beam_spec({'id': 'gmrt_x'}, first_frequency_hz=0.61e9)
    with pytest.raises(NotImplementedError):
        PB.device_beam_spec({'shape': 'rect', 'size': [3.0, 4.0]})
    with pytest.raises(TypeError):                                               # the reference wants a numpy array (:1616-1617)
        PB.device_beam_spec({'id': 'mwa'}, pointing_info={'delays': [0] * 16})


def test_error_code_mapping():
    with pytest.raises(ValueError):
        _abi._raise(_abi.PRISIM_EINVAL, 'x')
    with pytest.raises(MemoryError):
        _abi._raise(_abi.PRISIM_ENOMEM, 'x')
    with pytest.raises(RuntimeError):
        _abi._raise(_abi.PRISIM_ESTATE, 'x')
    with pytest.raises(_abi.PrisimHipError):
        _abi._raise(_abi.PRISIM_ENODEV, 'x')


def test_beamformer_settings_follow_the_reference_draw_order_and_checks():
    """Host side of the phased-array beamformer (primary_beams.py:1595-1668): delays from a pointing centre, explicit delays and
    gains, jitter drawn delays-first from numpy's global generator (the oracle restates the same rule independently)."""
    from oracle import beams_oracle as BO
    from prisim_amd import primary_beams as PB
    tile = PB.mwa_tile_element_locs()
    assert tile.shape == (16, 3) and NP.allclose(tile[0], [-1.65, 1.65, 0.0]) and NP.allclose(tile[-1], [1.65, -1.65, 0.0])
    pc = NP.array([0.3, -0.2, NP.sqrt(1 - 0.13)])
    info = {'pointing_center': pc, 'pointing_coords': 'dircos', 'delayerr': 0.25e-9, 'gainerr': 0.4, 'nrand': 5}
    NP.random.seed(11)
    d1, g1 = PB.beamformer_settings(tile, info)
    NP.random.seed(11)
    d2, g2 = BO.beamformer_settings(tile, info)
    assert d1.shape == (16, 5) and NP.array_equal(d1, d2) and NP.array_equal(g1, g2)
    d, g = PB.beamformer_settings(tile, {'pointing_center': NP.array([90.0, 0.0]), 'pointing_coords': 'altaz'})
    assert d.shape == (16, 1) and NP.allclose(d, 0.0, atol=1e-24) and NP.all(g == 1.0)      # zenith: no compensation needed
    d, g = PB.beamformer_settings(tile, None)
    assert NP.all(d == 0.0) and NP.all(g == 1.0)
    with pytest.raises(KeyError):
        PB.beamformer_settings(tile, {'pointing_center': pc})
    with pytest.raises(ValueError):
        PB.beamformer_settings(tile, {'delays': NP.zeros(3)})
    with pytest.raises(TypeError):
        PB.beamformer_settings(tile, {'delays': [0.0] * 16})
    with pytest.raises(ValueError):
        PB.beamformer_settings(tile, {'delayerr': -1.0})
    with pytest.raises(TypeError):
        PB.beamformer_settings(tile, {'nrand': 2.5})
    # telescope dictionary -> device beam spec
    kind, size, bpc, ext = PB.device_beam_spec({'id': 'mwa'}, pointing_info={'delays': NP.zeros(16)})
    assert 'beamformer' in ext and 'array' not in ext and ext['beamformer']['positions'].shape == (16, 3)
    kind, size, bpc, ext = PB.device_beam_spec({'id': 'mwa'})
    assert 'array' in ext and 'beamformer' not in ext
    kind, size, bpc, ext = PB.device_beam_spec({'shape': 'delta'}, pointing_info={'delays': NP.zeros(16)})     # no element_locs: factor 1 (:387-389)
    assert ext is None


def test_bench_quotes_profiled_traffic_only_for_the_running_sources(tmp_path, monkeypatch):
    """bench.roofline.traffic comes from profiles/*/pmc_summary.json only when that summary carries the hash of the kernel sources that
    are running (bench.csrc_hash); a summary of other sources is ignored (null beats stale)."""
    import json
    import bench
    h = bench.csrc_hash()
    assert len(h) == 40 and h == bench.csrc_hash()
    prof = tmp_path / 'profiles' / 'rXX'
    prof.mkdir(parents=True)
    (tmp_path / 'prisim_amd').mkdir()
    os.symlink(os.path.join(bench.ROOT, 'prisim_amd', 'csrc'), str(tmp_path / 'prisim_amd' / 'csrc'))
    summary = {'_kernel': {'Kernel_Name': 'void prisim::k_skyvis_rec_f32pk<64, false>(prisim::SkyvisParams)'},
               'FETCH_SIZE': {'mean_per_launch': 1000.0}, 'WRITE_SIZE': {'mean_per_launch': 500.0}, '_csrc_hash': 'not-this-build'}
    (prof / 'pmc_summary.json').write_text(json.dumps(summary))
    monkeypatch.setattr(bench, 'ROOT', str(tmp_path))
    assert bench.profiled_traffic('f32pk<64, false>') is None
    summary['_csrc_hash'] = h
    (prof / 'pmc_summary.json').write_text(json.dumps(summary))
    got = bench.profiled_traffic('f32pk<64, false>')
    assert got is not None and got[0] == (2 * 1000.0 + 500.0) * 1024.0      # 2 x FETCH_SIZE + WRITE_SIZE, KiB -> B (gfx950 FETCH correction)
    assert bench.profiled_traffic('k_skyvis_rec<double') is None


def test_committed_round_profiles_match_the_committed_sources():
    """The round's rocprofv3 summaries named in profiles/r06_MANIFEST.json (written by tools/profile_all_round6.sh after the last kernel
    change of the round; r04 / r05_MANIFEST.json name the earlier rounds', taken with those rounds' sources) were taken with the kernel sources that are
    committed beside them.  Summaries of earlier states of the round
    (A/B evidence) are not in the manifest; bench.py only quotes a summary as this build's traffic when its hash matches anyway."""
    import glob
    import json
    import bench
    manifest = os.path.join(bench.ROOT, 'profiles', 'r06_MANIFEST.json')
    if not os.path.exists(manifest):
        pytest.skip('no profile manifest yet (kernels still changing this round)')
    with open(manifest) as f:
        m = json.load(f)
    assert m['csrc_hash'] == bench.csrc_hash()
    assert len(m['dirs']) >= 6
    for d in m['dirs']:
        with open(os.path.join(bench.ROOT, 'profiles', d, 'pmc_summary.json')) as f:
            assert json.load(f).get('_csrc_hash') == bench.csrc_hash(), d


# ---- apply_gradients (interferometry.py:6726-6819): the consumer of the fused baseline gradient ----
def _oracle_array(monkeypatch, bl, ch):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import fake_context
    from prisim_amd import interferometry as RI
    monkeypatch.setattr(_abi, 'Context', fake_context.OracleContext)
    return RI.InterferometerArray(['b%d' % i for i in range(bl.shape[0])], bl, ch, telescope={'id': 'hera', 'orientation': [90.0, 270.0], 'ocoords': 'altaz'},
                                  latitude=-30.7, skycoords='altaz', pointing_coords='hadec')


def test_device_resident_gradients_and_catalogue_state_survive_conjugate(monkeypatch):
    """with reserve() the gradient blocks of observe() live only in the device gradient cube, which prisim_hip_set_array
    releases -- and conjugate() calls set_array.  They (and the catalogue-resident index lists) must be fetched first."""
    from prisim_amd import skymodel as SM
    rng = NP.random.default_rng(3)
    bl = rng.uniform(-60.0, 60.0, size=(5, 3)) * NP.array([1.0, 1.0, 0.01])
    ch = 150e6 + 1e5 * NP.arange(16)
    ia = _oracle_array(monkeypatch, bl, ch)
    skymod = SM.SkyModel(location=NP.stack((rng.uniform(20.0, 89.0, 30), rng.uniform(0.0, 360.0, 30)), axis=1), flux_ref=rng.uniform(0.5, 5.0, 30),
                         spindex=NP.zeros(30), ref_freq=150e6)
    ia.reserve(2)
    for j in range(2):
        ia.observe((2457000.5 + j, 10.0 + j), {'Tnet': 100.0}, NP.ones(16), [0.0, -30.7], skymod, 10.0, gradient_mode='baseline')
    assert type(ia.obs_catalog_indices[0]).__name__ == '_CatalogROI' and ia.obs_catalog_indices[0].size == 30       # the catalogue path ran
    want = NP.array(ia._ctx._grad[1])
    ia.conjugate(ind=[0, 3])
    got = ia.gradient['baseline']
    assert got.shape == (3, 5, 16, 2) and NP.array_equal(got[:, :, :, 1], want)
    assert NP.array_equal(NP.asarray(ia.obs_catalog_indices[1]), NP.arange(30))                                      # fetched before set_array dropped it
    # the next snapshot uploads the catalogue again and goes on
    ia2 = _oracle_array(monkeypatch, bl, ch)
    ia2.reserve(3)
    for j in range(2):
        ia2.observe((2457000.5 + j, 10.0 + j), {'Tnet': 100.0}, NP.ones(16), [0.0, -30.7], skymod, 10.0)
    ia2.conjugate(ind=[1])
    ia2.observe((2457002.5, 12.0), {'Tnet': 100.0}, NP.ones(16), [0.0, -30.7], skymod, 10.0)
    assert ia2.skyvis_freq.shape == (5, 16, 3)


def test_apply_gradients_matches_the_reference_method(monkeypatch):
    """tests/golden/golden_apply_gradients.npz holds what the reference's own method returned (make_golden.make_apply_gradients)."""
    g = NP.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'golden_apply_gradients.npz'))
    grad, ch = g['gradient'], g['channels']
    ia = _oracle_array(monkeypatch, NP.ones((grad.shape[1], 3)), ch)
    ia.gradient_mode, ia.gradient = 'baseline', {'baseline': grad}
    for name in ('seeds3', 'plain2d', 'grid5d', 'xy_only', 'x_only', 'four_axes'):
        pert = g['pert_' + name]
        given = {'baseline': pert.copy()}
        if name in ('xy_only', 'x_only', 'four_axes'):
            with pytest.warns(UserWarning):
                res = ia.apply_gradients(gradient_mode='baseline', perturbations=given)
        else:
            res = ia.apply_gradients(perturbations=given)
        ref = g['delta_' + name]
        assert res.shape == ref.shape and res.dtype == ref.dtype, name
        assert NP.max(NP.abs(res - ref)) <= 1e-13 * NP.max(NP.abs(ref)), name
        assert NP.array_equal(given['baseline'], pert)                      # the caller's array is left alone
    with pytest.raises(ValueError):                                         # the default stand-in is (1, 1, 1): one baseline only,
        ia.apply_gradients()                                                # as in the reference (:6762-6763, 6798-6799)


def test_apply_gradients_predicts_perturbed_visibilities(monkeypatch):
    """V(b + db) = V(b) + apply_gradients(db) + O(|db|^2): observe() with the gradient, then displace every baseline by a few mm.
    The reference's gradient (:6338, 6343) differentiates the geometric delays only, so the phase-centre delays stay those of the
    unperturbed baselines."""
    from oracle import skyvis_oracle as O, beams_oracle as BO
    rng = NP.random.default_rng(11)
    ch = 150e6 + 2e5 * NP.arange(6)
    bl = rng.uniform(-60.0, 60.0, size=(5, 3)) * NP.array([1.0, 1.0, 0.02])
    alt, az = rng.uniform(25.0, 89.0, 40), rng.uniform(0.0, 360.0, 40)
    skymod = SM.SkyModel(location=NP.stack((alt, az), axis=1), flux_ref=rng.uniform(0.5, 5.0, 40), spindex=rng.uniform(-1.0, 0.0, 40), ref_freq=150e6)
    ia = _oracle_array(monkeypatch, bl, ch)
    for j in range(2):
        ia.observe((2457000.5 + j, 10.0), {'Tnet': 100.0}, NP.ones(ch.size), [0.0, -30.7], skymod, 10.0, gradient_mode='baseline')
    assert ia.gradient['baseline'].shape == (3, 5, 6, 2)
    db = rng.normal(scale=2e-3, size=(3, 3, 5))                              # 3 realisations of mm-level position errors
    delta = ia.apply_gradients(perturbations={'baseline': db})
    assert delta.shape == (3, 5, 6, 2)
    pb = BO.airy_disk_pattern(14.0, NP.stack((alt, az), axis=1), ch, pointing_altaz=[90.0, 270.0]) * skymod.generate_spectrum(frequency=ch)
    dc = O.altaz2dircos(NP.stack((alt, az), axis=1))
    zen = NP.array([0.0, 0.0, 1.0])
    v0 = O.skyvis(bl, ch, dc, pb, zen)
    assert NP.max(NP.abs(ia.skyvis_freq[:, :, 0] - v0)) <= 1e-12 * NP.max(NP.abs(v0))
    scale = NP.sum(NP.abs(pb), axis=0)[None, :]
    for s in range(3):
        v1 = O.skyvis(bl + db[s].T, ch, dc, pb, zen) * NP.exp(-2j * NP.pi * ch[None, :] * (db[s].T @ zen)[:, None] / 299792458.0)
        first_order = NP.max(NP.abs(v1 - v0) / scale)
        resid = NP.max(NP.abs(v1 - v0 - delta[s, :, :, 0]) / scale)
        assert first_order > 1e-4 and resid < 2e-2 * first_order            # second order: (2 pi |db| / lambda) / 2 ~ 3e-3 of the first
        assert NP.array_equal(delta[s, :, :, 0], delta[s, :, :, 1])          # same altaz sky in both snapshots


def test_apply_gradients_argument_checks(monkeypatch):
    ch = 150e6 + 1e5 * NP.arange(3)
    ia = _oracle_array(monkeypatch, NP.ones((4, 3)), ch)
    with pytest.raises(AttributeError):
        ia.apply_gradients(perturbations={'baseline': NP.zeros((3, 4))})     # nothing observed with a gradient yet
    ia.gradient_mode, ia.gradient = 'baseline', {'baseline': NP.zeros((3, 4, 3, 1), dtype=NP.complex128)}
    with pytest.raises(TypeError):
        ia.apply_gradients(perturbations=NP.zeros((3, 4)))
    with pytest.raises(TypeError):
        ia.apply_gradients(gradient_mode=1, perturbations={'baseline': NP.zeros((3, 4))})
    with pytest.raises(KeyError):
        ia.apply_gradients(gradient_mode='skypos', perturbations={'skypos': NP.zeros((3, 4))})
    with pytest.raises(KeyError):
        ia.apply_gradients(gradient_mode='baseline', perturbations={'frequency': NP.zeros((3, 4))})
    with pytest.raises(TypeError):
        ia.apply_gradients(perturbations={'baseline': [[0.0] * 4] * 3})
    with pytest.raises(ValueError):
        ia.apply_gradients(perturbations={'baseline': NP.zeros(4)})
    with pytest.raises(ValueError):
        ia.apply_gradients(perturbations={'baseline': NP.zeros((3, 5))})      # five baselines against four


# ---- the device cube and the host cube must not drift apart (delay_transform() used stale device slots) ----
def _observed_oracle_array(monkeypatch, n_acc, reserve=None, host_staging=False):
    from prisim_amd import skymodel as SM
    rng = NP.random.default_rng(11)
    bl = rng.normal(size=(5, 3)) * 30.0
    ch = 150e6 + NP.arange(16) * 1e6
    ia = _oracle_array(monkeypatch, bl, ch)
    if reserve:
        ia.reserve(reserve, host_staging=host_staging)
    altaz = NP.stack((rng.uniform(30, 89, 12), rng.uniform(0, 360, 12)), axis=1)
    skymod = SM.SkyModel(location=altaz, flux_ref=rng.uniform(1, 5, 12), spindex=NP.full(12, -0.7), ref_freq=150e6)
    for j in range(n_acc):
        ia.observe((2457000.5 + j * 1e-3, 10.0 + j), {'Tnet': 100.0}, NP.ones(ch.size), [0.0, -30.7], skymod, 10.0)
    return ia, ch


@pytest.mark.parametrize('n_acc,reserve', [(1, None), (3, 3)])
def test_delay_transform_follows_an_assigned_skyvis_freq(monkeypatch, n_acc, reserve):
    from oracle import delay_oracle as DO
    ia, ch = _observed_oracle_array(monkeypatch, n_acc, reserve)
    ia.delay_transform(pad=1.0, verbose=False)
    lag0 = NP.array(ia.skyvis_lag)
    ia.skyvis_freq = 2.0 * NP.array(ia.skyvis_freq)            # the host cube changes; the device slots still hold the old one
    ia.delay_transform(pad=1.0, verbose=False)
    lag1 = NP.array(ia.skyvis_lag)
    assert NP.max(NP.abs(lag1 - 2.0 * lag0)) <= 1e-12 * NP.max(NP.abs(lag0))
    ref, _ = DO.delay_transform(ia.skyvis_freq, ia.bp, ia.bp_wts, ch[1] - ch[0], pad=1.0)
    assert NP.max(NP.abs(lag1 - ref)) <= 1e-12 * NP.max(NP.abs(ref))


def test_host_staging_gives_the_same_cube_as_the_lazy_download(monkeypatch):
    monkeypatch.setattr(_abi, 'host_empty', lambda shape, dtype: NP.empty(shape, dtype=dtype))
    from prisim_amd import interferometry as RI
    staged, _ = _observed_oracle_array(monkeypatch, 3, reserve=3, host_staging=True)
    lazy, _ = _observed_oracle_array(monkeypatch, 3, reserve=3)
    assert all(isinstance(s, RI._DeviceSlot) and s.staged for s in staged._cube)
    assert all(isinstance(s, RI._DeviceSlot) and not s.staged for s in lazy._cube)
    a, b = staged.skyvis_freq, lazy.skyvis_freq
    staged3, _ = _observed_oracle_array(monkeypatch, 3, reserve=3, host_staging=True)
    snaps = staged3.skyvis_freq_snapshots()                    # snapshot-major: the pinned cube itself, no copy
    assert snaps.shape == (3, 5, 16) and NP.shares_memory(snaps, staged3._host_cube)
    assert a.shape == b.shape == (5, 16, 3) and NP.array_equal(a, b) and a.flags['C_CONTIGUOUS']      # the reference's layout
    assert NP.array_equal(NP.moveaxis(snaps, 0, 2), b) and NP.array_equal(lazy.skyvis_freq_snapshots(), snaps)
    # a re-centring on the device invalidates staged copies: they are queued again, and the host sees the rotated cube
    pc = NP.array([[80.0, 200.0]])
    staged2, _ = _observed_oracle_array(monkeypatch, 3, reserve=3, host_staging=True)
    staged2.phase_centering(phase_center=pc, phase_center_coords='altaz', verbose=False)
    lazy.phase_centering(phase_center=pc, phase_center_coords='altaz', verbose=False)
    assert NP.max(NP.abs(staged2.skyvis_freq - lazy.skyvis_freq)) <= 1e-12 * NP.max(NP.abs(b))
    assert NP.max(NP.abs(staged2.skyvis_freq - b)) > 1e-3 * NP.max(NP.abs(b))


def test_host_staging_falls_back_when_pinned_memory_is_unavailable(monkeypatch):
    def refuse(shape, dtype):
        raise MemoryError('no pinned memory')
    monkeypatch.setattr(_abi, 'host_empty', refuse)
    with pytest.warns(UserWarning, match='host staging switched off'):
        ia, _ = _observed_oracle_array(monkeypatch, 2, reserve=2, host_staging=True)
    assert ia.skyvis_freq.shape == (5, 16, 2) and not ia._stage


def test_host_staging_is_refused_when_the_pinned_cube_would_take_most_of_the_free_memory(monkeypatch):
    from prisim_amd import interferometry as RI_
    assert RI_._available_host_bytes() > 0                                     # MemAvailable of /proc/meminfo on this box
    monkeypatch.setattr(RI_, '_available_host_bytes', lambda: 1000)             # (the budget is 2 x the cube: pinned + its stacked copy)
    monkeypatch.setattr(_abi, 'host_empty', lambda shape, dtype: (_ for _ in ()).throw(AssertionError('must not be reached')))
    with pytest.warns(UserWarning, match='host staging switched off: the pinned host cube and its'):
        ia, _ = _observed_oracle_array(monkeypatch, 2, reserve=2, host_staging=True)
    assert ia.skyvis_freq.shape == (5, 16, 2) and not ia._stage


def test_equatorial_baselines_are_rotated_to_the_local_frame(monkeypatch):
    """baseline_coords='equatorial' (interferometry.py:6151-6153: GEOM.xyz2enu at the array's latitude): the same array given in the
    equatorial frame observes what it observes given in ENU; the rotation itself against a hand calculation."""
    import fake_context
    from prisim_amd import interferometry as RI, skymodel as SM
    lat = -30.7224
    s, c = NP.sin(NP.radians(lat)), NP.cos(NP.radians(lat))
    assert NP.allclose(GEOM.enu2xyz([[0.0, 1.0, 0.0]], lat), [[-s, 0.0, c]])              # North: towards the pole, tilted by the latitude
    assert NP.allclose(GEOM.enu2xyz([[1.0, 0.0, 0.0]], lat), [[0.0, 1.0, 0.0]])           # East is Y
    assert NP.allclose(GEOM.enu2xyz([[0.0, 0.0, 1.0]], lat), [[c, 0.0, s]])               # Up: hour angle 0, declination = latitude
    rng = NP.random.default_rng(3)
    bl = rng.uniform(-100.0, 100.0, size=(6, 3)) * NP.array([1.0, 1.0, 0.05])
    assert NP.allclose(GEOM.xyz2enu(GEOM.enu2xyz(bl, lat), lat), bl, rtol=0, atol=1e-12)
    monkeypatch.setattr(_abi, 'Context', fake_context.OracleContext)
    ch = 150e6 + 1e5 * NP.arange(8)
    skymod = SM.SkyModel(location=NP.stack((rng.uniform(20, 89, 30), rng.uniform(0, 360, 30)), axis=1), flux_ref=rng.uniform(1, 5, 30),
                         spindex=NP.full(30, -0.7), ref_freq=150e6)
    kw = dict(telescope={'id': 'hera'}, latitude=lat, skycoords='altaz', pointing_coords='hadec')
    ia_enu = RI.InterferometerArray(['b%d' % i for i in range(6)], bl, ch, baseline_coords='localenu', **kw)
    ia_eq = RI.InterferometerArray(['b%d' % i for i in range(6)], GEOM.enu2xyz(bl, lat), ch, baseline_coords='equatorial', **kw)
    for ia in (ia_enu, ia_eq):
        ia.observe((2457000.5, 10.0), {'Tnet': 100.0}, NP.ones(ch.size), [0.0, lat], skymod, 10.0)
    assert ia_eq.baseline_coords == 'equatorial' and NP.allclose(ia_eq.baselines, GEOM.enu2xyz(bl, lat))      # stored as given
    assert NP.max(NP.abs(ia_eq.skyvis_freq - ia_enu.skyvis_freq)) <= 1e-10 * NP.max(NP.abs(ia_enu.skyvis_freq))
    assert NP.allclose(NP.asarray(ia_eq.geometric_delays[0]), NP.asarray(ia_enu.geometric_delays[0]), rtol=0, atol=1e-18)
    with pytest.raises(ValueError):
        RI.InterferometerArray(['b0'], bl[:1], ch, baseline_coords='galactic', **kw)


def test_gradient_blocks_stay_on_the_device_until_read(monkeypatch):
    """With reserve() the baseline-gradient blocks of every snapshot stay in the device gradient cube (observe() downloads nothing); the
    `gradient` dictionary fetches and stacks them when it is first read, and equals what an unreserved run (synchronous downloads) holds."""
    import fake_context
    from prisim_amd import interferometry as RI, skymodel as SM
    downloads = {'n': 0}

    class Spy(fake_context.OracleContext):
        def get_vis(self, slot=0, want_grad=False, complex64=False):
            downloads['n'] += 1
            return fake_context.OracleContext.get_vis(self, slot=slot, want_grad=want_grad, complex64=complex64)

    monkeypatch.setattr(_abi, 'Context', Spy)
    rng = NP.random.default_rng(8)
    ch = 150e6 + 2e5 * NP.arange(6)
    bl = rng.uniform(-60.0, 60.0, size=(5, 3)) * NP.array([1.0, 1.0, 0.02])
    skymod = SM.SkyModel(location=NP.stack((rng.uniform(25, 89, 30), rng.uniform(0, 360, 30)), axis=1), flux_ref=rng.uniform(0.5, 5.0, 30),
                         spindex=rng.uniform(-1.0, 0.0, 30), ref_freq=150e6)
    kw = dict(telescope={'id': 'hera'}, latitude=-30.7, skycoords='altaz', pointing_coords='hadec')
    res = {}
    for reserve in (True, False):
        ia = RI.InterferometerArray(['b%d' % i for i in range(5)], bl, ch, **kw)
        if reserve:
            ia.reserve(3)
        downloads['n'] = 0
        for j in range(3):
            ia.observe((2457000.5 + j, 10.0 + 5 * j), {'Tnet': 100.0}, NP.ones(ch.size), [0.0, -30.7], skymod, 10.0, gradient_mode='baseline')
        if reserve:
            assert downloads['n'] == 0 and all(isinstance(g, RI._DeviceSlot) for g in ia._grad)
            assert 'baseline' in ia.gradient and bool(ia.gradient) and list(ia.gradient.keys()) == ['baseline']
        res[reserve] = NP.array(ia.gradient['baseline'])
        assert res[reserve].shape == (3, 5, 6, 3)
        if reserve:
            assert downloads['n'] == 3 and not any(isinstance(g, RI._DeviceSlot) for g in ia._grad)
            n = downloads['n']
            assert ia.gradient['baseline'] is ia.gradient['baseline'] and downloads['n'] == n      # stacked once
            ia.observe((2457003.5, 25.0), {'Tnet': 100.0}, NP.ones(ch.size), [0.0, -30.7], skymod, 10.0, gradient_mode='baseline')
            assert ia.gradient['baseline'].shape == (3, 5, 6, 4)                                   # a fourth snapshot (slot 0 reused)
            assert NP.array_equal(ia.gradient['baseline'][..., :3], res[True])
    assert NP.max(NP.abs(res[True] - res[False])) <= 1e-12 * NP.max(NP.abs(res[False]))


# ---- the last executable statements next to the path, against tests/golden/golden_aux.npz (the reference's lines executed) ----
def _golden_aux():
    from conftest import GOLDEN
    return NP.load(os.path.join(GOLDEN, 'golden_aux.npz'))


def test_project_baselines_matches_reference_statements(monkeypatch):
    """interferometry.py:7980-7985: the uvw rotation matrix per snapshot and the projection NP.dot(eq_baselines, rot_matrix).  The inputs
    of the golden vector are equatorial baselines and (HA, Dec) in radians; they reach the class method through GEOM.xyz2enu / hadec
    reference points (astroutils conventions, unpinned), whose round trips cost a few ulp: tolerance 1e-12 of the baseline length."""
    from prisim_amd import geometry as GEOM
    g = _golden_aux()
    lat = -30.7
    bl = GEOM.xyz2enu(g['proj_eq_baselines'], lat, 'degrees')
    ia = _oracle_array(monkeypatch, bl, 150e6 + 1e5 * NP.arange(4))
    ia.latitude = lat
    nt = g['proj_ha'].size
    ia.n_acc, ia.lst = nt, [0.0] * nt
    ia.project_baselines({'location': NP.degrees(NP.stack((g['proj_ha'], g['proj_dec']), axis=1)), 'coords': 'hadec'})
    scale = NP.max(NP.abs(g['proj_eq_baselines']))
    assert ia.projected_baselines.shape == g['projected_baselines'].shape == (6, 3, nt)
    assert NP.max(NP.abs(ia.projected_baselines - g['projected_baselines'])) <= 1e-12 * scale
    ia.n_acc, ia.lst = 1, [0.0]
    ia.project_baselines({'location': NP.degrees(NP.array([g['proj_ha'][0], g['proj_dec'][0]])), 'coords': 'hadec'})
    assert NP.max(NP.abs(ia.projected_baselines - g['projected_baselines_one'])) <= 1e-12 * scale


def test_conjugate_matches_reference_statements(monkeypatch):
    """interferometry.py:8035-8045: flipped baselines, conjugated cubes, reversed label pairs, negated projected baselines -- exactly."""
    g = _golden_aux()
    ia = _oracle_array(monkeypatch, g['conj_baselines_in'].copy(), 150e6 + 1e5 * NP.arange(5))
    ia.labels = [(i, 100 + i) for i in range(6)]
    ia.n_acc = 4
    ia.skyvis_freq = g['conj_skyvis_in'].copy()
    ia.vis_freq, ia.vis_noise_freq = g['conj_vis_in'].copy(), g['conj_noise_in'].copy()
    ia.projected_baselines = g['conj_proj_in'].copy()
    ia.conjugate(ind=g['conj_ind'], verbose=False)
    assert NP.array_equal(ia.baselines, g['conj_baselines']) and NP.array_equal(ia.baseline_orientations, g['conj_orientations'])
    assert NP.array_equal(ia.skyvis_freq, g['conj_skyvis']) and NP.array_equal(ia.vis_freq, g['conj_vis'])
    assert NP.array_equal(ia.vis_noise_freq, g['conj_noise']) and NP.array_equal(ia.projected_baselines, g['conj_proj'])
    assert NP.array_equal(NP.array([list(l) for l in ia.labels]), g['conj_labels'])


@pytest.mark.parametrize('unit', ['JY', 'K'])
def test_vis_rms_freq_matches_reference_statements(monkeypatch, unit):
    """interferometry.py:6676-6691: vis_rms_freq = 2 k / sqrt(t_acc df) Tsys / (A_eff eff_Q) / Jy, or Tsys / eff_Q / sqrt(t_acc df) in K
    (CNST.Jy of the un-vendored astroutils is 1e-26).  Tolerance 4 ulp (the order of the products differs)."""
    g = _golden_aux()
    ia = _oracle_array(monkeypatch, NP.zeros((6, 3)) + NP.arange(6)[:, None], 150e6 + float(g['rms_df']) * NP.arange(7))
    ia.freq_resolution = float(g['rms_df'])
    ia.eff_Q, ia.A_eff = g['rms_effQ_' + unit], g['rms_Aeff_' + unit]
    ia.Tsys = g['rms_Tsys_' + unit]
    ia.t_acc = list(g['rms_tacc_' + unit])
    ia.timestamp = [2457000.5 + t for t in range(4)]
    ia.flux_unit = unit
    ia.generate_noise(seed=7)
    want = g['rms_out_' + unit]
    assert ia.vis_rms_freq.shape == want.shape and NP.max(NP.abs(ia.vis_rms_freq / want - 1.0)) <= 1e-15
    assert ia.vis_noise_freq.shape == want.shape and NP.iscomplexobj(ia.vis_noise_freq)


# ---- round 6: the resident-catalogue path of the class under failure, edits and frames (CPU: the oracle stand-in for the context) ----
def _radec_case(monkeypatch, ctx_cls=None, nsrc=60, nbl=4, nchan=8):
    import fake_context
    from prisim_amd import interferometry as RI, skymodel as SM
    monkeypatch.setattr(_abi, 'Context', ctx_cls or fake_context.OracleContext)
    rng = NP.random.default_rng(19)
    lat = -30.7224
    ch = 150e6 + 2e5 * NP.arange(nchan)
    bl = rng.uniform(-120.0, 120.0, size=(nbl, 3)) * NP.array([1.0, 1.0, 0.02])
    skymod = SM.SkyModel(location=NP.stack((rng.uniform(0, 360, nsrc), NP.degrees(NP.arcsin(rng.uniform(-1, 0.5, nsrc)))), axis=1),
                         flux_ref=rng.uniform(1, 5, nsrc), spindex=rng.uniform(-1, 0, nsrc), ref_freq=150e6, epoch='J2000')
    ia = RI.InterferometerArray(['b%d' % i for i in range(nbl)], bl, ch, telescope={'id': 'hera'}, latitude=lat, skycoords='radec',
                                pointing_coords='hadec')
    return ia, skymod, ch, lat


def test_out_of_device_memory_on_the_resident_path_falls_back_to_the_upload_path(monkeypatch):
    """the resident-catalogue path sizes buffers for whole chunks; when the device says MemoryError -- in set_catalog, in a
    single observe(), or in a batch -- the sky model continues on the per-snapshot upload path (ROI rows only), with a warning, and the
    results are those of that path; the instance stays aligned."""
    import warnings
    import fake_context

    class Tight(fake_context.OracleContext):
        refuse = 'observe'

        def set_catalog(self, *a, **k):
            if Tight.refuse == 'catalog':
                raise MemoryError('hipMalloc: out of memory (stand-in)')
            return fake_context.OracleContext.set_catalog(self, *a, **k)

        def observe_catalog(self, *a, **k):
            if Tight.refuse in ('observe', 'batch'):
                raise MemoryError('hipMalloc: out of memory (stand-in)')
            return fake_context.OracleContext.observe_catalog(self, *a, **k)

    want, _, _, _ = _radec_case(monkeypatch)
    _, skymod, ch, lat = _radec_case(monkeypatch)
    times = [(2461041.5 + 0.001 * t, 40.0 + 0.5 * t) for t in range(3)]
    for t in times:
        want.observe(t, {'Tnet': 100.0}, NP.ones(ch.size), [0.0, lat], skymod, 10.0)
    assert type(want.obs_catalog_indices[0]).__name__ == '_CatalogROI'
    for mode in ('observe', 'catalog', 'batch'):
        Tight.refuse = mode
        ia, _, _, _ = _radec_case(monkeypatch, Tight)
        ia.reserve(3)
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter('always')
            if mode == 'batch':
                ia.observe_batch(times, {'Tnet': 100.0}, NP.ones(ch.size), [0.0, lat], skymod, 10.0)
            else:
                for t in times:
                    ia.observe(t, {'Tnet': 100.0}, NP.ones(ch.size), [0.0, lat], skymod, 10.0)
        assert sum('per-snapshot upload path' in str(x.message) for x in w) == 1, [str(x.message) for x in w]     # said once, then remembered
        assert ia.n_acc == 3 and len(ia.timestamp) == 3 and ia.pointing_center.shape == (3, 2) and ia.bp.shape == (4, ch.size, 3)
        assert isinstance(ia.obs_catalog_indices[0], NP.ndarray)                       # the upload path's index arrays
        for t in range(3):
            assert NP.array_equal(NP.asarray(ia.obs_catalog_indices[t]), NP.asarray(want.obs_catalog_indices[t]))
        assert NP.max(NP.abs(ia.skyvis_freq - want.skyvis_freq)) <= 1e-12 * NP.max(NP.abs(want.skyvis_freq))


def test_in_place_edits_of_the_sky_model_and_epoch_of_date(monkeypatch):
    """The resident catalogue follows the sky model's CONTENT (flux, positions, sizes, epoch) -- an edit in place between two observe() calls
    is seen; invalidate_catalog() forces an upload; epoch None / 'date' means coordinates of date (nothing is precessed); PRISIM_FRAME_MODEL
    / frame_model pick the astrometric model."""
    from prisim_amd import frames as FR, geometry as GEOM
    ia, skymod, ch, lat = _radec_case(monkeypatch)
    uploads = {'n': 0}
    orig = ia._ctx.set_catalog

    def counting(*a, **k):
        uploads['n'] += 1
        return orig(*a, **k)
    ia._ctx.set_catalog = counting
    args = ({'Tnet': 100.0}, NP.ones(ch.size), [0.0, lat], skymod, 10.0)
    ia.observe((2461041.5, 40.0), *args)
    ia.observe((2461041.6, 41.0), *args)
    assert uploads['n'] == 1                                             # unchanged model: stays resident
    skymod.flux_ref[7] *= 3.0                                            # one element, in place
    ia.observe((2461041.5, 40.0), *args)
    assert uploads['n'] == 2 and NP.max(NP.abs(ia.skyvis_freq[:, :, 2] - ia.skyvis_freq[:, :, 0])) > 0
    skymod.flux_ref[7] /= 3.0
    skymod.location[3, 1] += 0.5
    ia.observe((2461041.5, 40.0), *args)
    assert uploads['n'] == 3
    ia.invalidate_catalog()
    ia.observe((2461041.5, 40.0), *args)
    assert uploads['n'] == 4
    # epoch: J2000 -> precessed; None / 'date' -> hour angle = LST - RA on the coordinates as given
    sel_app = NP.asarray(ia.obs_catalog_indices[-1])
    rot, beta = FR.snapshot_frame('radec', 40.0, lat, jd=2461041.5, epoch='J2000', model='apparent')
    assert NP.array_equal(sel_app, GEOM.roi_select(GEOM.frame_dircos(GEOM.catalog_unitvec(skymod.location, 'radec'), rot, beta), 'zenith', 90.0))
    for epoch in (None, 'date'):
        skymod.epoch = epoch
        ia.observe((2461041.5, 40.0), *args)
        rot, beta = FR.snapshot_frame('radec', 40.0, lat, model='date')
        dc = GEOM.frame_dircos(GEOM.catalog_unitvec(skymod.location, 'radec'), rot, beta)
        assert NP.array_equal(NP.asarray(ia.obs_catalog_indices[-1]), GEOM.roi_select(dc, 'zenith', 90.0))
        assert NP.max(NP.abs(NP.asarray(ia.geometric_delays[-1]) - dc[GEOM.roi_select(dc, 'zenith', 90.0)].dot(ia.baselines.T) / 299792458.0)) <= 1e-18
    skymod.epoch = 'J2000'
    ia.frame_model = 'mean'
    ia.observe((2461041.5, 40.0), *args)
    rot, beta = FR.snapshot_frame('radec', 40.0, lat, jd=2461041.5, epoch='J2000', model='mean')
    assert NP.all(beta == 0.0)
    assert NP.array_equal(NP.asarray(ia.obs_catalog_indices[-1]),
                          GEOM.roi_select(GEOM.frame_dircos(GEOM.catalog_unitvec(skymod.location, 'radec'), rot, beta), 'zenith', 90.0))
    with pytest.raises(ValueError):
        ia.frame_model = 'ptolemaic'
        ia.observe((2461041.5, 40.0), *args)


def test_frame_provider_in_a_batch_and_close(monkeypatch):
    """A caller-supplied frame is asked once per snapshot of a batch, with that snapshot's (jd, lst); close() fetches
    what still lives on the device (index lists, snapshots) before the context goes."""
    from prisim_amd import frames as FR
    ia, skymod, ch, lat = _radec_case(monkeypatch)
    calls = []

    def provider(jd, lst, sm):
        calls.append((jd, lst))
        return FR.snapshot_frame('radec', lst, lat, jd=jd, epoch=sm.epoch, model='apparent')
    ia.frame_provider = provider
    times = [(2461041.5 + 0.002 * t, 10.0 + t) for t in range(4)]
    ia.observe_batch(times, {'Tnet': 100.0}, NP.ones(ch.size), [0.0, lat], skymod, 10.0)
    assert calls == times
    builtin, _, _, _ = _radec_case(monkeypatch)
    builtin.observe_batch(times, {'Tnet': 100.0}, NP.ones(ch.size), [0.0, lat], skymod, 10.0)
    assert NP.array_equal(ia.skyvis_freq, builtin.skyvis_freq)
    closed = {'n': 0}
    real_close = ia._ctx.close

    def closing():
        closed['n'] += 1
        real_close()
    ia._ctx.close = closing
    assert type(ia.obs_catalog_indices[2]).__name__ == '_CatalogROI' and ia.obs_catalog_indices[2]._idx is None      # still lazy
    ia.close()
    assert closed['n'] == 1 and ia.obs_catalog_indices[2]._idx is not None                                            # fetched first
    assert NP.array_equal(NP.asarray(ia.obs_catalog_indices[2]), NP.asarray(builtin.obs_catalog_indices[2]))


def test_frozen_sky_model_is_recognised_by_identity_and_edits_are_still_seen(monkeypatch):
    """SkyModel.freeze() (what driver.run does to the model it builds): the resident catalogue is recognised by the identity of the model's
    private read-only arrays -- no pass over their contents per observe() -- while everything that can still change is still seen: a new
    array assigned to an attribute, an array made writeable again, another model."""
    import fake_context
    from prisim_amd import _abi, interferometry as RI, skymodel as SM
    monkeypatch.setattr(_abi, 'Context', fake_context.OracleContext)
    rng = NP.random.default_rng(8)
    n = 40
    loc = NP.stack((rng.uniform(0, 360, n), rng.uniform(-60, 0, n)), axis=1)
    flux = rng.uniform(1, 2, n)
    sm = SM.SkyModel(location=loc, flux_ref=flux, spindex=NP.zeros(n), ref_freq=150e6, epoch=None)
    bl = NP.array([[14.6, 0.0, 0.0], [0.0, 14.6, 0.0]])
    ch = 150e6 + 1e6 * NP.arange(8)
    ia = RI.InterferometerArray(['a', 'b'], bl, ch, telescope={'id': 'hera', 'orientation': [90.0, 270.0], 'ocoords': 'altaz'}, latitude=-30.7,
                                skycoords='radec', pointing_coords='hadec')
    k_content = ia._catalog_fingerprint(sm)
    assert sm.freeze() is sm and not sm.flux_ref.flags.writeable and sm.flux_ref is not flux and sm.location.flags.owndata
    flux[:] = 7.0                                           # the caller's own array is no longer the model's
    assert NP.all(sm.flux_ref < 3.0)
    k1, k2 = ia._catalog_fingerprint(sm), ia._catalog_fingerprint(sm)
    assert k1 == k2 and k1 != k_content and len(k1) < len(k_content)          # the identity form, stable from call to call
    # an attribute assigned after the freeze: back to the content pass, and the key differs
    sm.flux_ref = NP.array(sm.flux_ref) * 2.0
    k3 = ia._catalog_fingerprint(sm)
    assert k3 != k1 and len(k3) == len(k_content)
    sm.freeze()
    k4 = ia._catalog_fingerprint(sm)
    assert k4 != k1 and len(k4) == len(k1)                   # frozen again: new private arrays, a new identity
    # an array made writeable again is not trusted
    sm.spindex.setflags(write=True)
    assert len(ia._catalog_fingerprint(sm)) == len(k_content)
    sm.spindex.setflags(write=False)
    assert ia._catalog_fingerprint(sm) == k4
    # another frozen model with the same content is another catalogue (identity), and observing works through the frozen model
    sm2 = SM.SkyModel(location=sm.location, flux_ref=sm.flux_ref, spindex=sm.spindex, ref_freq=150e6, epoch=None).freeze()
    assert ia._catalog_fingerprint(sm2) != k4
    for j in range(2):
        ia.observe((2457000.5 + j, 30.0 + j), {'Tnet': 100.0}, NP.ones(ch.size), [0.0, -30.7], sm, 10.0)
    assert ia._catalog_key == ia._catalog_fingerprint(sm)
    v = NP.array(ia.skyvis_freq)
    assert v.shape == (2, 8, 2) and NP.all(NP.isfinite(v))
