"""GPU (-m gpu): BASELINE configs 3, 4 and 5 through the reference's entry for the path -- `run_prisim.py -i parms.yaml`
(scripts/run_prisim.py:60-65, 2165-2207 of the reference; here prisim_amd.driver.main) -- from the shipped examples/config{3,4,5}.yaml at
full array size and reduced n_acc, the saved NPZ compared with the C oracle on a baseline sample; config 4 (sharded over GPUs in
BASELINE.json) also as a 2-rank run on the one GPU of the test box with only librccl replaced (tests/fake_rccl)."""
import os
import shutil
import subprocess
import sys

import numpy as NP
import pytest
import yaml

from oracle import c_oracle as CO, beams_oracle as BO, delay_oracle as DO
from prisim_amd import driver, workloads as W

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EX = os.path.join(ROOT, 'examples')
SIDEREAL_DEG_PER_SEC = 360.0 * 1.00273790935 / 86400.0
ZEN = NP.array([0.0, 0.0, 1.0])


def _yaml(tmp_path, name, override):
    with open(os.path.join(EX, name + '.yaml')) as f:
        p = yaml.safe_load(f)
    p = driver.deep_merge(p, override)
    p['dirstruct']['rootdir'] = str(tmp_path) + '/'
    p.setdefault('save_formats', {})
    p['save_formats'].update({'npz': True, 'hdf5': False, 'npz_compress': False})
    if p.get('array', {}).get('file'):
        p['array']['file'] = os.path.join(ROOT, p['array']['file'])
    path = tmp_path / (name + '.yaml')
    path.write_text(yaml.safe_dump(p))
    npz = os.path.join(str(tmp_path), p['dirstruct']['project'], p['dirstruct']['simid'], 'simdata', 'simvis.npz')
    return str(path), npz, p


def _spot(nbl, n=4):
    return NP.unique(NP.linspace(0, nbl - 1, n).astype(int))


def _fake_rccl(tmp_path):
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    fake = tmp_path / 'libfake_rccl.so'
    res = subprocess.run([hipcc, '-O2', '-std=c++17', '-fPIC', '-shared', '-x', 'hip', '--offload-arch=gfx950', '-I/opt/rocm/include',
                          os.path.join(ROOT, 'tests', 'fake_rccl', 'fake_rccl.cpp'), '-o', str(fake)], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    return str(fake)


def _run_two_ranks(path, fake):
    env = dict(os.environ, PRISIM_RCCL_LIB=fake, PRISIM_DEVICE='0', OMP_NUM_THREADS='2')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'PRISIM_RDZV_FILE', 'MASTER_PORT'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', 'run_prisim.py'), '-n', '2', '-i', path], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]


def test_config3_yaml(tmp_path):
    """examples/config3.yaml: HERA-350 x 1024 channels x (1e4 point sources + nside-128 diffuse), one LST, fp32."""
    path, npz, p = _yaml(tmp_path, 'config3', {})
    assert driver.main(['-i', path]) == 0
    _check_config3(npz)


def test_config3_yaml_two_ranks_on_one_gpu(tmp_path):
    """The same YAML as `run_prisim.py -n 2`: every rank's half of the baselines needs its sources split, and a sky of two runs (point
    sources + diffuse map) is then summed run by run into one set of partial cubes per run (capi.cpp run_pass) -- the product path of
    profiles/r04_shard_balance.json's taper rows.  Both ranks on the one GPU of the test box, only librccl replaced."""
    fake = _fake_rccl(tmp_path)
    path, npz, p = _yaml(tmp_path, 'config3', {'dirstruct': {'simid': 'cfg3_2ranks'}})
    _run_two_ranks(path, fake)
    _check_config3(npz)


def _check_config3(npz):
    out = NP.load(npz)
    cfg = W.config3(with_diffuse=True)
    bl, ch, sky = cfg['baselines'], cfg['channels'], cfg['sky']
    assert out['skyvis_freq'].shape == (61075, 1024, 1) and out['skyvis_freq'].dtype == NP.complex64
    assert NP.array_equal(out['freq'], ch) and NP.max(NP.abs(out['bl'] - bl)) == 0.0
    sel = _spot(bl.shape[0], 5)
    pb = BO.airy_disk_pattern(14.0, sky['altaz'], ch) * (sky['flux_ref'][:, None] * (ch[None, :] / sky['ref_freq']) ** sky['spindex'][:, None])
    ref = CO.skyvis(bl[sel], ch, sky['dircos'], pb, ZEN, fwhm_deg=sky['fwhm_deg'])
    err = float(NP.max(NP.abs(out['skyvis_freq'][sel, :, 0] - ref) / NP.sum(NP.abs(pb), axis=0)[None, :]))
    print('config 3 YAML: max err / sum|pbflux| = %.3e' % err)
    assert err <= 5e-6


def _config4_inputs(tmp_path):
    sys.path.insert(0, EX)
    import make_config4_inputs
    d = tmp_path / 'cfg4_inputs'
    d.mkdir(exist_ok=True)
    beam = make_config4_inputs.main(str(d))
    with open(os.path.join(str(d), 'config4_mwa128_layout.txt')) as f, open(os.path.join(EX, 'config4_mwa128_layout.txt')) as g:
        assert f.read() == g.read()                                       # the committed layout is what the generator writes
    return beam


def _check_config4(npz, n_acc, snapshots=None):
    from oracle import healpix_oracle as H
    out = NP.load(npz)
    cfg = W.config4(n_acc=n_acc)
    bl, ch, sky, lat = cfg['baselines'], cfg['channels'], cfg['sky'], cfg['latitude']
    cube = out['skyvis_freq']
    assert cube.shape == (8128, 768, n_acc) and cube.dtype == NP.complex64 and NP.max(NP.abs(out['bl'] - bl)) == 0.0
    sel = _spot(bl.shape[0])
    for j in (range(n_acc) if snapshots is None else snapshots):
        dc, altaz, keep = W.drift_snapshot_directions(sky, lat, j * cfg['t_acc'] * SIDEREAL_DEG_PER_SEC)
        flux = sky['flux_ref'][keep, None] * (ch[None, :] / sky['ref_freq']) ** sky['spindex'][keep, None]
        beam = H.external_beam(cfg['beam_table'], cfg['beam_freqs'], NP.pi / 2 - NP.radians(altaz[:, 0]), NP.radians(altaz[:, 1]), ch)
        pb = beam.astype(NP.float32).astype(NP.float64) * flux            # supplied beams are stored float32 (interferometry.py:4466)
        ref = CO.skyvis(bl[sel], ch, dc, pb, ZEN, fwhm_deg=sky['fwhm_deg'][keep])
        err = float(NP.max(NP.abs(cube[sel, :, j] - ref) / NP.sum(NP.abs(pb), axis=0)[None, :]))
        print('config 4 YAML snapshot %d: max err / sum|pbflux| = %.3e' % (j, err))
        assert err <= 5e-6, j
    assert NP.max(NP.abs(cube[:, :, 0] - cube[:, :, n_acc - 1])) > 0


def test_config4_yaml(tmp_path):
    """examples/config4.yaml: MWA-128T (committed synthetic layout file) x 768 channels, drift scan, nside-64 diffuse sky, external beam
    read from the HDF5 gain_info/<pol> layout (examples/make_config4_inputs.py), 3 of its 32 accumulations."""
    beam = _config4_inputs(tmp_path)
    path, npz, p = _yaml(tmp_path, 'config4', {'obsparm': {'n_acc': 3}, 'beam': {'file': beam}})
    assert driver.main(['-i', path]) == 0
    _check_config4(npz, 3)


def test_config4_yaml_all_32_accumulations(tmp_path):
    """examples/config4.yaml as shipped: the whole drift scan, 32 accumulations of 112 s (the sky moves by 15 degrees), every snapshot in
    its own slot of the device cube with the downloads overlapped; first, middle and last snapshot against the oracle."""
    beam = _config4_inputs(tmp_path)
    path, npz, p = _yaml(tmp_path, 'config4', {'beam': {'file': beam}, 'dirstruct': {'simid': 'cfg4_full'}})
    assert p['obsparm']['n_acc'] == 32
    assert driver.main(['-i', path]) == 0
    _check_config4(npz, 32, snapshots=(0, 15, 31))
    out = NP.load(npz)
    assert out['lst'].shape == (32,) and abs((out['lst'][31] - out['lst'][0]) - 31 * 112.0 * SIDEREAL_DEG_PER_SEC) <= 1e-9
    assert NP.all(NP.isfinite(out['skyvis_freq'].view(NP.float32)))


def test_config4_yaml_two_ranks_on_one_gpu(tmp_path):
    """The same YAML as `run_prisim.py -n 2` (baselines sharded, communicator + self-test, gather to rank 0, pp.gather: root), both ranks
    on the one GPU of the test box with only librccl replaced."""
    fake = _fake_rccl(tmp_path)
    beam = _config4_inputs(tmp_path)
    path, npz, p = _yaml(tmp_path, 'config4', {'obsparm': {'n_acc': 2}, 'beam': {'file': beam}, 'dirstruct': {'simid': 'cfg4_2ranks'}})
    assert p['pp']['gather'] == 'root'
    _run_two_ranks(path, fake)
    _check_config4(npz, 2)


def test_config5_yaml(tmp_path):
    """examples/config5.yaml: HERA-350 x 1024 channels x nside-256 diffuse sky, 2 of its 120 LSTs, then the delay transform
    (processing.delay_transform, Blackman-Harris window, f_pad = 1): visibilities and delay spectra against the oracle."""
    path, npz, p = _yaml(tmp_path, 'config5', {'obsparm': {'n_acc': 2}})
    assert driver.main(['-i', path]) == 0
    out = NP.load(npz)
    cfg = W.config5(n_acc=2)
    bl, ch, sky, lat = cfg['baselines'], cfg['channels'], cfg['sky'], cfg['latitude']
    cube, lag = out['skyvis_freq'], out['skyvis_lag']
    assert cube.shape == (61075, 1024, 2) and lag.shape == (61075, 1024, 2) and out['lags'].shape == (1024,)
    sel = _spot(bl.shape[0], 3)
    ref_cube = NP.empty((sel.size, ch.size, 2), dtype=NP.complex128)
    for j in range(2):
        dc, altaz, keep = W.drift_snapshot_directions(sky, lat, j * cfg['t_acc'] * SIDEREAL_DEG_PER_SEC)
        pb = BO.airy_disk_pattern(14.0, altaz, ch) * (sky['flux_ref'][keep, None] * (ch[None, :] / sky['ref_freq']) ** sky['spindex'][keep, None])
        ref = CO.skyvis(bl[sel], ch, dc, pb, ZEN, fwhm_deg=sky['fwhm_deg'][keep])
        err = float(NP.max(NP.abs(cube[sel, :, j] - ref) / NP.sum(NP.abs(pb), axis=0)[None, :]))
        print('config 5 YAML LST %d: max err / sum|pbflux| = %.3e' % (j, err))
        assert err <= 5e-6, j
        ref_cube[:, :, j] = ref
    w = driver.window(ch.size, 'bhw')
    ref_lag, ref_lags = DO.delay_transform(ref_cube, NP.ones(ref_cube.shape), NP.broadcast_to(w[None, :, None], ref_cube.shape), ch[1] - ch[0], pad=1.0)
    # the spectra were formed from the fp32 visibilities: the same tolerance relative to the largest lag amplitude of the row
    assert NP.max(NP.abs(lag[sel] - ref_lag)) <= 2e-5 * NP.max(NP.abs(ref_lag))
    assert NP.allclose(out['lags'], ref_lags, rtol=0, atol=1e-18)
