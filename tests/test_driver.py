"""YAML driver (prisim_amd/driver.py, scripts/run_prisim.py): host logic on CPU, end-to-end runs on the GPU."""
import copy
import os
import subprocess
import sys

import numpy as NP
import pytest
import yaml

from prisim_amd import driver, workloads as W

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EX = os.path.join(ROOT, 'examples')


def test_load_parms_defaults_and_template_merge(tmp_path):
    tmpl = tmp_path / 'template.yaml'
    tmpl.write_text(yaml.safe_dump({'bandpass': {'nchan': 32, 'freq': 120e6}, 'telescope': {'id': 'hera'}}))
    parm = tmp_path / 'parms.yaml'
    parm.write_text(yaml.safe_dump({'preload': {'template': 'template.yaml'}, 'bandpass': {'nchan': 48}}))
    p = driver.load_parms(str(parm))
    assert p['bandpass']['nchan'] == 48 and p['bandpass']['freq'] == 120e6          # file overrides template overrides defaults
    assert p['bandpass']['freq_resolution'] == 390625.0 and p['telescope']['id'] == 'hera'
    assert p['pp']['key'] == 'bl' and p['diagnosis']['wait_after_run'] is False       # never drops into a debugger


def test_baseline_info_selection_and_redundancy():
    p = driver.load_parms(os.path.join(EX, 'config2.yaml'))
    bl, labels, pos, groups = driver.baseline_info(p)                                 # array.redundant: false -> every pair
    assert bl.shape == (171, 3) and len(labels) == 171 and pos.shape == (19, 3) and all(len(v) == 1 for v in groups.values())
    p['baseline']['max'] = 15.0
    assert driver.baseline_info(p)[0].shape[0] == 42                                  # 14.6 m spacings of HERA-19
    p['baseline']['max'] = None
    p['array']['redundant'] = True                                                    # the reference's default: unique baselines only
    blu, lblu, _, grp = driver.baseline_info(p)
    assert blu.shape[0] == 30 and len(grp) == 30                                      # unique baselines of a 19-element hexagon
    assert sum(len(v) for v in grp.values()) == 171 and max(len(v) for v in grp.values()) == 14
    assert NP.all(NP.diff(NP.sqrt((blu ** 2).sum(1))) >= -1e-9) and all(lbl in grp[lbl] for lbl in lblu)
    p1 = driver.load_parms(os.path.join(EX, 'config1.yaml'))
    p1['array']['file'] = os.path.join(EX, 'config1_layout.txt')
    bl1 = driver.baseline_info(p1)[0]
    assert bl1.shape == (6, 3) and NP.isclose(NP.sqrt((bl1 ** 2).sum(1)).min(), 14.6, atol=1e-3)


def test_uniq_baselines_modes():
    """interferometry.py:1373-1461: None = all distinct, True = those seen more than once, False = those seen once."""
    from prisim_amd import layouts as LAY
    bl = NP.array([[14.6, 0, 0], [0, 14.6, 0], [14.6, 0, 0], [29.2, 0, 0], [-14.6, 0, 0], [7.3, 12.644, 0.0]])
    ub, first, counts, occ = LAY.uniq_baselines(bl)
    assert ub.shape == (4, 3) and sorted(counts.tolist()) == [1, 1, 1, 3]            # -b folds onto b (orientation modulo 180 deg)
    assert sorted(map(sorted, occ)) == [[0, 2, 4], [1], [3], [5]]
    assert all(first[i] == min(occ[i]) for i in range(4))
    rb, rfirst, rcounts, rocc = LAY.uniq_baselines(bl, redundant=True)
    assert rb.shape == (1, 3) and rcounts.tolist() == [3] and sorted(rocc[0]) == [0, 2, 4]
    nb, _, ncounts, _ = LAY.uniq_baselines(bl, redundant=False)
    assert nb.shape == (3, 3) and ncounts.tolist() == [1, 1, 1]
    assert LAY.uniq_baselines(bl[:, :2])[0].shape == (4, 3)                           # 2-column input is padded with zeros
    with pytest.raises(TypeError):
        LAY.uniq_baselines(bl.tolist())
    with pytest.raises(TypeError):
        LAY.uniq_baselines(bl, redundant='yes')


def test_schedule_drift_and_track():
    p = driver.load_parms(os.path.join(EX, 'config2.yaml'))
    jd, lst, hadec, t_acc, n_acc = driver.schedule(p)
    assert n_acc == 2 and t_acc == 1080.0 and NP.allclose(hadec, [[0.0, -30.7224]] * 2)
    assert NP.isclose(lst[1] - lst[0], 1080.0 * 360.0 * 1.00273790935 / 86400.0)
    assert NP.isclose(jd[1] - jd[0], 1080.0 / 86400.0) and NP.isclose(jd[0], 2457349.5)           # 2015/11/23 0h UT
    p['obsparm']['obs_mode'] = 'track'
    p['pointing']['lst_init'] = 1.0
    p['pointing']['track_init'] = {'ra': 10.0, 'dec': -25.0, 'ha': 0.0, 'epoch': '2000'}
    jd, lst, hadec, _, _ = driver.schedule(p)
    assert NP.isclose(hadec[0, 0], 5.0) and NP.allclose(hadec[:, 1], -25.0) and hadec[1, 0] > hadec[0, 0]
    p['obsparm']['obs_mode'] = 'dns'
    with pytest.raises(ValueError):
        driver.schedule(p)


def test_custom_catalog_and_flux_cut():
    p = driver.load_parms(os.path.join(EX, 'config1.yaml'))
    sm = driver.build_skymodel(p, EX)
    assert sm.location.shape == (100, 2) and NP.all(sm.spindex == -0.83) and NP.all(sm.src_shape == 0)
    sky = W.point_source_sky(100, 1)
    assert NP.allclose(sm.flux_ref, sky['flux_ref'], rtol=0, atol=1e-8)
    p['skyparm']['flux_min'] = 5.0
    sm2 = driver.build_skymodel(p, EX)
    assert 0 < sm2.flux_ref.size < 100 and NP.all(sm2.flux_ref >= 5.0)
    p['skyparm']['flux_min'] = 1e9
    with pytest.raises(IndexError):
        driver.build_skymodel(p, EX)
    p['skyparm']['model'] = 'nvss'
    with pytest.raises(NotImplementedError):
        driver.build_skymodel(p, EX)


def test_flux_cut_uses_each_sources_own_catalog_spectral_index(tmp_path):
    """run_prisim.py:1649-1660: the thresholds given at fluxcut_reffreq are moved to the catalog frequency with the catalog's
    SPINDEX column (assigned at :1649, before the cut), not with the global skyparm.spindex."""
    cat = tmp_path / 'cat.txt'
    rows = ['RA DEC F_INT SPINDEX MAJAX MINAX PA',
            '10.0 -30.0 1.00 -2.0 0 0 0',        # steep: 1 Jy at 150 MHz is 0.32 Jy... at 200 MHz threshold scale (150/200)^-2 = 1.78
            '20.0 -30.0 1.00  0.0 0 0 0',        # flat: threshold scale 1
            '30.0 -30.0 1.50 -2.0 0 0 0',
            '40.0 -30.0 2.50 -2.0 0 0 0']
    cat.write_text('\n'.join(rows) + '\n')
    p = driver.deep_merge(driver.DEFAULTS, {'catalog': {'custom_file': str(cat)}, 'bandpass': {'freq': 150e6},
                                            'skyparm': {'model': 'custom', 'custom_reffreq': 0.150, 'flux_min': 1.0, 'flux_max': 2.0,
                                                        'fluxcut_reffreq': 200e6, 'spindex': 0.0}})
    sm = driver.build_skymodel(p, str(tmp_path))
    # per-source thresholds at 150 MHz: steep sources [1.78, 3.56], flat source [1, 2] -> rows 2 (flat, 1.0) and 4 (steep, 2.5) pass
    assert sorted(sm.location[:, 0].tolist()) == [20.0, 40.0]
    # with the global spindex (0.0) rows 1, 2, 3 would have passed instead
    assert sm.flux_ref.tolist() == [1.0, 2.5] and sm.spindex.tolist() == [0.0, -2.0]


def test_external_beam_hdf5_gain_info_layout(tmp_path):
    """scripts/FEKO_beam_to_healpix.py:161-198 writes gain_info/<pol> (nfreq x npix) + spectral_info/freqs; run_prisim.py:489-494 reads it."""
    from prisim_amd import hdf5io
    try:
        f = hdf5io.File(str(tmp_path / 'beam.hdf5'), 'w')
    except hdf5io.HDF5Unavailable:
        pytest.skip('libhdf5 not available')
    rng = NP.random.default_rng(5)
    gains = {'P1': rng.uniform(0.1, 1.0, (3, 48)), 'P2': rng.uniform(0.1, 1.0, (3, 48))}
    with f:
        f.write('header/npol', 2)
        f.write('spectral_info/freqs', NP.array([100e6, 150e6, 200e6]))
        for pol, g in gains.items():
            f.write('gain_info/' + pol, g)
    p = driver.deep_merge(driver.DEFAULTS, {'beam': {'use_external': True, 'file': str(tmp_path / 'beam.hdf5'), 'filefmt': 'HDF5'}})
    beam, freqs = driver.load_external_beam(p, '.')
    assert beam.shape == (48, 3) and NP.array_equal(beam, gains['P1'].T) and NP.array_equal(freqs, [100e6, 150e6, 200e6])
    p['beam']['pol'] = 'P2'
    assert NP.array_equal(driver.load_external_beam(p, '.')[0], gains['P2'].T)
    p['beam']['pol'] = 'P9'
    with pytest.raises(KeyError):
        driver.load_external_beam(p, '.')


def test_window_shapes():
    for shape in ('rect', 'bhw', 'bnw'):
        w = driver.window(64, shape)
        assert w.shape == (64,) and NP.isclose(w.mean(), 1.0) and NP.allclose(w, w[::-1])
    assert driver.window(64, 'bhw').max() > 2.0 and driver.window(64, 'bhw')[0] < 1e-3
    with pytest.raises(ValueError):
        driver.window(64, 'hann')


def test_unsupported_modes_fail_loudly():
    p = driver.load_parms(os.path.join(EX, 'config1.yaml'))
    p['pp']['key'] = 'pixels'                   # ('freq' and 'src' are accepted: test_stock_partition_keys_run_on_baseline_shards)
    with pytest.raises(ValueError):
        driver.run(p)
    p['pp']['key'] = 'bl'
    p['beam']['use_external'] = True
    p['beam']['filefmt'] = 'uvbeam'
    with pytest.raises(NotImplementedError):
        driver.run(p)


def test_shipped_yaml_files_describe_the_baseline_configs():
    """examples/config3/4/5.yaml: the arrays, bands, skies and schedules of BASELINE.json's configs 3-5; config 4's committed
    layout file is what its generator writes and gives the workload's 8128 baselines."""
    p3 = driver.load_parms(os.path.join(EX, 'config3.yaml'))
    assert driver.baseline_info(p3)[0].shape == (61075, 3) and p3['bandpass']['nchan'] == 1024 and p3['obsparm']['n_acc'] == 1
    sm = driver.build_skymodel(p3, EX)
    cfg = W.config3(with_diffuse=True)
    assert sm.location.shape == (cfg['sky']['dircos'].shape[0], 2) and NP.array_equal(sm.flux_ref, cfg['sky']['flux_ref'])
    assert NP.count_nonzero(sm.src_shape[:, 0] == 0.0) == 10000                      # the point sources come first: runs of one source size
    p4 = driver.load_parms(os.path.join(EX, 'config4.yaml'))
    p4['array']['file'] = os.path.join(ROOT, p4['array']['file'])
    bl4 = driver.baseline_info(p4)[0]
    c4 = W.config4()
    assert bl4.shape == (8128, 3) and NP.max(NP.abs(bl4 - c4['baselines'])) == 0.0
    assert p4['obsparm']['n_acc'] == 32 and p4['beam']['use_external'] and p4['beam']['filefmt'] == 'hdf5' and p4['pp']['gather'] == 'root'
    assert NP.array_equal(W.channel_grid(p4['bandpass']['freq'], p4['bandpass']['freq_resolution'], p4['bandpass']['nchan']), c4['channels'])
    p5 = driver.load_parms(os.path.join(EX, 'config5.yaml'))
    assert p5['obsparm']['n_acc'] == 120 and p5['skyparm']['nside'] == 256 and p5['processing']['delay_transform'] is True
    with pytest.raises(ValueError):
        bad = driver.load_parms(os.path.join(EX, 'config3.yaml'))
        bad['skyparm']['components'] = []
        driver.build_skymodel(bad, EX)


def test_stock_partition_keys_run_on_baseline_shards(monkeypatch):
    """defaultparms.yaml:939 ships pp.key 'freq' (loop at run_prisim.py:1858-1995; 'src' :1996-2080).  The key only chooses how the sum is
    cut: the run proceeds on baseline shards with a warning naming the substitution, and gives the visibilities of pp.key 'bl' exactly.
    When the reference tree is at hand (the build container; never on the GPU box) its own defaultparms.yaml is what gets loaded."""
    import warnings
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import fake_context
    from prisim_amd import _abi
    monkeypatch.setattr(_abi, 'Context', fake_context.OracleContext)
    stock = '/root/reference/prisim/examples/simparms/defaultparms.yaml'
    if os.path.exists(stock):
        ref = driver.load_parms(stock)
        assert ref['pp']['key'] == 'freq'
    base = driver.load_parms(os.path.join(EX, 'config1.yaml'))
    base['array']['file'] = os.path.join(EX, 'config1_layout.txt')
    base['catalog']['custom_file'] = os.path.join(EX, 'config1_catalog.txt')
    base['processing']['add_noise'] = False
    base['processing']['delay_transform'] = False
    out = {}
    for key in ('bl', 'freq', 'src'):
        p = copy.deepcopy(base)
        p['pp']['key'] = key
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter('always')
            out[key] = driver.run(p, infile_dir=EX, verbose=False)['skyvis_freq']
        named = [str(x.message) for x in w if 'pp.key' in str(x.message)]
        assert (len(named) == 1 and repr(key) in named[0] and 'baselines' in named[0]) if key != 'bl' else not named
    assert NP.array_equal(out['freq'], out['bl']) and NP.array_equal(out['src'], out['bl']) and NP.any(out['bl'] != 0)


def test_sharded_run_stops_on_every_rank_when_any_self_test_fails(monkeypatch):
    """driver.run builds the communicator and self-tests it before the first snapshot; the outcomes are combined over the rendezvous:
    a rank whose own test passed still exits 3 when a peer's failed, and without a rendezvous the error itself propagates."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import fake_context
    from prisim_amd import _abi
    observed = {'n': 0}

    class Ctx(fake_context.OracleContext):
        fail = False

        def comm_init(self, uid, nranks, rank):
            self.nranks = nranks

        def comm_selftest(self, nbytes=1 << 20):
            if Ctx.fail:
                raise _abi.PrisimHipError('self-test pattern mismatch')

        def compute(self, *a, **k):
            observed['n'] += 1
            return fake_context.OracleContext.compute(self, *a, **k)

    class Rdzv(object):
        def __init__(self, peer):
            self.peer = peer

        def allgather(self, obj):
            return [obj, self.peer]

    monkeypatch.setattr(_abi, 'Context', Ctx)
    p = driver.load_parms(os.path.join(EX, 'config2.yaml'))
    uid = b'x' * 128
    with pytest.raises(SystemExit) as e:                                      # my test passed, the peer's did not
        driver.run(p, infile_dir=EX, rank=0, world=2, comm_uid=uid, verbose=False, host_copy='root', rdzv=Rdzv([False, 'peer: no data moved']))
    assert e.value.code == 3 and observed['n'] == 0                           # before any snapshot
    Ctx.fail = True
    with pytest.raises(SystemExit) as e:                                      # mine failed: reported to the peers, exit 3
        driver.run(p, infile_dir=EX, rank=1, world=2, comm_uid=uid, verbose=False, host_copy='root', rdzv=Rdzv([True, '']))
    assert e.value.code == 3 and observed['n'] == 0
    with pytest.raises(_abi.PrisimHipError):                                  # no rendezvous to report to: the error itself
        driver.run(p, infile_dir=EX, rank=0, world=2, comm_uid=uid, verbose=False)
    with pytest.raises(ValueError):
        driver.run(p, infile_dir=EX, rank=0, world=2, comm_uid=None, verbose=False)
    # comm_init is a collective -- a rank whose init fails must END (the launcher then stops its peers, who are inside
    # ncclCommInitRank), not go into a rendezvous vote its peers never reach.  The error propagates even with a rendezvous at hand.
    Ctx.fail = False

    class NeverCalled(object):
        def allgather(self, obj):
            raise AssertionError('a failed comm_init must not be voted on')

    def failing_init(self, uid, nranks, rank):
        raise _abi.PrisimHipError('ncclCommInitRank: unhandled system error')
    monkeypatch.setattr(Ctx, 'comm_init', failing_init)
    with pytest.raises(_abi.PrisimHipError):
        driver.run(p, infile_dir=EX, rank=1, world=2, comm_uid=uid, verbose=False, host_copy='root', rdzv=NeverCalled())


def _catalogue_in_the_snapshot_frame(parms, skymod, lat):
    """(direction cosines, alt-az) of every source of config 1's catalogue at the run's one snapshot, with all sources above the horizon."""
    from prisim_amd import frames as FR, geometry as GEOM
    jd, lst, _, _, _ = driver.schedule(parms)
    rot, beta = FR.snapshot_frame('radec', float(lst[0]), lat, jd=float(jd[0]), epoch=skymod.epoch, model='apparent')
    dc = GEOM.frame_dircos(GEOM.catalog_unitvec(skymod.location, 'radec'), rot, beta)
    assert NP.all(dc[:, 2] > 0.0)
    return dc, GEOM.dircos2altaz(dc)


@pytest.mark.gpu
def test_config1_yaml_end_to_end_matches_oracle(tmp_path):
    """BASELINE config 1 through scripts/run_prisim.py -i examples/config1.yaml, checked against the oracle."""
    from oracle import skyvis_oracle as O, beams_oracle as BO, delay_oracle as DO
    parms = yaml.safe_load(open(os.path.join(EX, 'config1.yaml')))
    parms['dirstruct']['rootdir'] = str(tmp_path) + '/'
    parms['array']['file'] = os.path.join(EX, 'config1_layout.txt')
    parms['catalog']['custom_file'] = os.path.join(EX, 'config1_catalog.txt')
    infile = tmp_path / 'cfg1.yaml'
    infile.write_text(yaml.safe_dump(parms))
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', 'run_prisim.py'), '-i', str(infile)], cwd=ROOT,
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    out = NP.load(os.path.join(str(tmp_path), 'prisim_amd_cfg1', 'cfg1', 'simdata', 'simvis.npz'))
    # the reference's npz keys (interferometry.py:8860-8861, noise added as in run_prisim.py:2278-2279) + labels and the delay spectra
    assert sorted(out.files) == sorted(['skyvis_freq', 'vis_freq', 'vis_noise_freq', 'lst', 'freq', 'timestamp', 'bl', 'bl_length', 'labels',
                                        'skyvis_lag', 'lags'])
    assert NP.allclose(out['vis_freq'], out['skyvis_freq'] + out['vis_noise_freq'], rtol=1e-12, atol=0) and NP.std(out['vis_noise_freq'].real) > 0
    vis = out['skyvis_freq']
    assert vis.shape == (6, 64, 1)
    # oracle: the same catalogue (equinox J2000, skyparm.epoch) in the local frame of the snapshot -- precession to the date of the run,
    # nutation, aberration, rotation by the LST (prisim_amd/frames.py; run_prisim.py:1690-1692 + interferometry.py:6174-6180)
    p = driver.load_parms(str(infile))
    sm = driver.build_skymodel(p, EX)
    lat = p['telescope']['latitude']
    dc, altaz = _catalogue_in_the_snapshot_frame(p, sm, lat)
    ch = out['freq']
    pb = BO.gaussian_beam(14.0, altaz, ch, pointing_altaz=O.hadec2altaz([[0.0, lat]], lat)[0]) * sm.generate_spectrum(frequency=ch)
    pc = O.altaz2dircos(O.hadec2altaz([[0.0, lat]], lat))[0]
    ref = O.skyvis(out['bl'], ch, dc, pb, pc, fwhm_deg=NP.zeros(altaz.shape[0]))
    assert NP.max(NP.abs(vis[:, :, 0] - ref) / O.abs_flux_sum(pb)[None, :]) <= 1e-11
    w = driver.window(64, 'bhw')
    lag, lags = DO.delay_transform(ref[:, :, None], NP.ones((6, 64, 1)), NP.broadcast_to(w[None, :, None], (6, 64, 1)), ch[1] - ch[0], pad=1.0)
    assert NP.max(NP.abs(out['skyvis_lag'] - lag)) <= 1e-10 * NP.max(NP.abs(lag))


@pytest.mark.gpu
def test_phasing_center_of_the_yaml_is_where_the_visibilities_end_up():
    """run_prisim.py:2281-2282: after the snapshots the visibilities are re-centred on phasing.center.  A run with the phase centre 10 degrees
    off zenith must equal the oracle's sky-sum phased there directly."""
    from oracle import skyvis_oracle as O, beams_oracle as BO
    p = driver.load_parms(os.path.join(EX, 'config1.yaml'))
    p['array']['file'] = os.path.join(EX, 'config1_layout.txt')
    p['catalog']['custom_file'] = os.path.join(EX, 'config1_catalog.txt')
    p['phasing'] = {'center': [80.0, 30.0], 'coords': 'altaz'}
    p['processing']['add_noise'] = False
    p['processing']['delay_transform'] = False
    out = driver.run(p, infile_dir=EX, verbose=False)
    assert 'vis_freq' not in out
    sm = driver.build_skymodel(p, EX)
    lat = p['telescope']['latitude']
    dc, altaz = _catalogue_in_the_snapshot_frame(p, sm, lat)
    ch = out['freq']
    pb = BO.gaussian_beam(14.0, altaz, ch, pointing_altaz=O.hadec2altaz([[0.0, lat]], lat)[0]) * sm.generate_spectrum(frequency=ch)
    pc_new = O.altaz2dircos(NP.array([[80.0, 30.0]]))[0]
    ref = O.skyvis(out['bl'], ch, dc, pb, pc_new, fwhm_deg=NP.zeros(altaz.shape[0]))
    assert NP.max(NP.abs(out['skyvis_freq'][:, :, 0] - ref) / O.abs_flux_sum(pb)[None, :]) <= 1e-11
    assert NP.allclose(out['ia'].phase_center, [[80.0, 30.0]] if out['ia'].phase_center_coords == 'altaz' else out['ia'].phase_center)


@pytest.mark.gpu
def test_config2_yaml_runs_and_reserves_device_cube():
    p = driver.load_parms(os.path.join(EX, 'config2.yaml'))
    out = driver.run(p, infile_dir=EX, verbose=False)
    assert out['skyvis_freq'].shape == (171, 256, 2) and NP.all(NP.isfinite(out['skyvis_freq'].view(NP.float64)))
    ia = out['ia']
    # the device-resident cube holds the same snapshots (slot t) as the host attribute
    for t in range(2):
        assert NP.array_equal(ia._ctx.get_vis(slot=t), out['skyvis_freq'][:, :, t])
    # device-resident delay transform (no re-upload) equals the per-snapshot host path
    from oracle import delay_oracle as DO
    ia.delay_transform(pad=1.0, verbose=False)
    ref_lag, ref_lags = DO.delay_transform(out['skyvis_freq'], ia.bp, ia.bp_wts, ia.freq_resolution, pad=1.0)
    assert NP.max(NP.abs(ia.skyvis_lag - ref_lag)) <= 1e-10 * NP.max(NP.abs(ref_lag)) and NP.allclose(ia.lags, ref_lags)
    # single-rank "gather" through the same code path the multi-GPU driver uses
    ia._ctx.allgather(2)
    g = ia._ctx.get_gathered(2, 1)
    assert NP.array_equal(NP.transpose(g[:, 0], (1, 2, 0)), out['skyvis_freq'])


@pytest.mark.gpu
def test_config2_unique_baselines_then_save_redundant(tmp_path):
    """array.redundant: true (the reference's default) simulates the 30 unique baselines of HERA-19; save_redundant re-creates all
    171 at save time (run_prisim.py:2325-2326).  The expanded file equals the all-baselines run to the 0.01 m / 0.001 arcsec at which
    the reference calls baselines redundant."""
    p_all = driver.load_parms(os.path.join(EX, 'config2.yaml'))
    p_all['dirstruct']['rootdir'] = str(tmp_path) + '/'
    p_all['dirstruct']['simid'] = 'all'
    out_all = driver.run(p_all, infile_dir=EX, verbose=False)
    f_all = NP.load(driver.save(out_all, p_all))
    p_u = driver.load_parms(os.path.join(EX, 'config2.yaml'))
    p_u['array']['redundant'] = True
    p_u['dirstruct']['rootdir'] = str(tmp_path) + '/'
    p_u['dirstruct']['simid'] = 'uniq'
    out_u = driver.run(p_u, infile_dir=EX, verbose=False)
    assert out_u['skyvis_freq'].shape == (30, 256, 2)
    f_u = NP.load(driver.save(out_u, p_u))
    assert f_u['skyvis_freq'].shape == (171, 256, 2) and f_u['bl'].shape == (171, 3) and f_u['labels'].shape == (171,)
    assert sorted(f_u['labels'].tolist()) == sorted(f_all['labels'].tolist())
    order_u = NP.argsort(f_u['labels'])
    order_a = NP.argsort(f_all['labels'])
    assert NP.allclose(f_u['bl'][order_u], f_all['bl'][order_a], atol=1e-2)
    scale = NP.max(NP.abs(f_all['skyvis_freq']))
    assert NP.max(NP.abs(f_u['skyvis_freq'][order_u] - f_all['skyvis_freq'][order_a])) <= 1e-9 * scale
    p_u['save_redundant'] = False
    p_u['dirstruct']['simid'] = 'uniq_only'
    assert NP.load(driver.save(out_u, p_u))['skyvis_freq'].shape == (30, 256, 2)
    # PRISim's HDF5 file of the same run, redundant baselines re-created by InterferometerArray.duplicate_measurements
    from prisim_amd import hdf5io
    try:
        hdf5io._load()
    except hdf5io.HDF5Unavailable:
        return
    p_u['save_redundant'] = True
    p_u['save_formats']['hdf5'] = True
    p_u['dirstruct']['simid'] = 'uniq_hdf5'
    with pytest.warns(UserWarning):
        npz = driver.save(out_u, p_u)
    with hdf5io.File(npz[:-4] + '.hdf5', 'r') as f:
        cube = f.read('visibilities/freq_spectrum/skyvis')
        assert cube.shape == (171, 256, 2) and f.read('array/baselines').shape == (171, 3) and f.read('array/labels').shape == (171,)
        assert f.exists('blgroupinfo/groups') and f.read('visibilities/freq_spectrum/noise').shape == (171, 256, 2)
    assert NP.array_equal(NP.sort_complex(cube[:, 0, 0]), NP.sort_complex(f_u['skyvis_freq'][:, 0, 0]))
