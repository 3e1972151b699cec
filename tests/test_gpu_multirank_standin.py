"""GPU (-m gpu): the product's multi-rank path END TO END on the one GPU of a test box -- bench.py --gpus N and run_prisim.py -n N launched
bare (prisim_amd.launch), the real rendezvous, the real libprisim_hip.so communicator code (comm_init, self-test, per-snapshot overlapped
gathers with their event timing, complex64 send buffers, lag / noise / gradient gathers, gather-to-root, the gathered-cube layout and
checksums) and the real kernels on every rank.  The ONLY stand-in is librccl itself (tests/fake_rccl/fake_rccl.cpp, loaded through
PRISIM_RCCL_LIB): RCCL refuses two ranks on one GPU, so it moves the device buffers between the rank processes through files.  What this
cannot cover is RCCL's own transport -- that runs for the first time in the driver's 8-GPU scaling run."""
import json
import os
import shutil
import subprocess
import sys

import numpy as NP
import pytest
import yaml

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def fake_rccl(tmp_path_factory):
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    out = tmp_path_factory.mktemp('fake_rccl') / 'libfake_rccl.so'
    res = subprocess.run([hipcc, '-O2', '-std=c++17', '-fPIC', '-shared', '-x', 'hip', '--offload-arch=gfx950', '-I/opt/rocm/include',
                          os.path.join(ROOT, 'tests', 'fake_rccl', 'fake_rccl.cpp'), '-o', str(out)], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    return str(out)


def _env(fake):
    env = dict(os.environ, PRISIM_RCCL_LIB=fake, PRISIM_BENCH_DEVICE='0', PRISIM_DEVICE='0', OMP_NUM_THREADS='2')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'PRISIM_RDZV_FILE', 'MASTER_PORT'):
        env.pop(k, None)
    return env


@pytest.mark.parametrize('nranks,precision', [(2, 'fp32'), (3, 'fp64')])
def test_bench_gpus_n_runs_end_to_end_on_one_gpu(fake_rccl, nranks, precision):
    """`python bench.py --gpus N` bare: self-spawned ranks, communicator + self-test, timed loop with per-snapshot gathers (complex64 on the
    wire for fp32), gather verification on every rank, ONE contract line with the N > 1 keys.  3 ranks: the last shard is padded."""
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(nranks), '--steps', '2', '--warmup', '1', '--nsrc', '1500',
                          '--precision', precision, '--no-cpu-baseline'], env=_env(fake_rccl), cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, res.stdout
    d = json.loads(lines[0])
    assert d['n_gpus'] == nranks and d['steps'] == 2 and d['gather_ok'] is True and d['launcher'] == 'prisim_amd.launch'
    assert d['config']['nbl'] == 61075 and d['config']['nsrc'] == 1500 and d['dtype'] == ('f32' if precision == 'fp32' else 'f64')
    sys.path.insert(0, ROOT)
    from prisim_amd import sharding
    per = sharding.shard_size(61075, nranks)                       # groups of 256 baselines dealt round-robin, padded to the largest share
    assert d['roofline']['terms_per_launch'] == float(per) * 1024 * 1500
    g = d['gather']
    assert g['wire_dtype'] == ('complex64' if precision == 'fp32' else 'complex128')
    assert g['bytes_per_peer'] == per * 1024 * (8 if precision == 'fp32' else 16) and g['gathers_measured'] == 2
    assert g['comm_stream_priority'] <= 0 <= g['lowest_priority'] and g['per_snapshot_ms'] >= 0 and g['exposed_ms'] >= 0
    assert g['order'] == 'global' and 0.0 < g['undeal_ms_per_snapshot'] < g['per_snapshot_ms']      # un-dealt on the device, behind each gather
    assert len(d['kernel_ms_per_rank']['all']) == nranks and d['kernel_ms_per_rank']['min'] > 0
    assert abs(d['value_n1_equiv'] * nranks - d['value']) <= 1e-9 * d['value']
    assert abs(d['value'] * d['ms_per_step'] * 1e-3 * 2 - 61075.0 * 1024 * 1500 * 2) <= 1e-6 * 61075.0 * 1024 * 1500 * 2


def _parms(tmp_path, simid, extra=None):
    p = {'dirstruct': {'rootdir': str(tmp_path) + '/', 'project': 'p', 'simid': simid},
         'array': {'layout': 'HERA-19', 'redundant': False},
         'telescope': {'id': 'hera', 'latitude': -30.7224},
         'bandpass': {'freq': 150e6, 'freq_resolution': 1e6, 'nchan': 32},
         'obsparm': {'n_acc': 3, 't_acc': 600.0, 'obs_mode': 'drift'},
         'pointing': {'lst_init': 1.0, 'drift_init': {'ha': 0.0, 'dec': -30.7224}},
         'skyparm': {'model': 'ptsrc_random', 'n_src': 300, 'seed': 7, 'custom_reffreq': 0.150, 'spindex': -0.8},
         'processing': {'delay_transform': True, 'f_pad': 1.0, 'bpass_shape': 'bhw', 'noise_seed': 20261004},
         'phasing': {'center': [75.0, 40.0], 'coords': 'altaz'},
         'save_formats': {'npz': True, 'hdf5': False}}
    if extra:
        from prisim_amd import driver
        p = driver.deep_merge(p, extra)
    path = tmp_path / (simid + '.yaml')
    path.write_text(yaml.safe_dump(p))
    return str(path), os.path.join(str(tmp_path), 'p', simid, 'simdata', 'simvis.npz')


@pytest.mark.parametrize('gather', ['all', 'root'])
def test_run_prisim_n_ranks_writes_what_one_rank_writes(fake_rccl, tmp_path, gather):
    """scripts/run_prisim.py -n 3 -i parms.yaml (the `mpirun -n 3` of the reference) against the same YAML on one rank: visibilities,
    noisy visibilities, delay spectra -- every array of the NPZ -- agree; with pp.gather = root only rank 0 ever holds the whole cube."""
    script = os.path.join(ROOT, 'scripts', 'run_prisim.py')
    y1, npz1 = _parms(tmp_path, 'one')
    yn, npzn = _parms(tmp_path, 'many_' + gather, {'pp': {'gather': gather}})
    r1 = subprocess.run([sys.executable, script, '-i', y1], env=_env(fake_rccl), cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r1.returncode == 0, r1.stderr[-3000:]
    rn = subprocess.run([sys.executable, script, '-n', '3', '-i', yn], env=_env(fake_rccl), cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert rn.returncode == 0, rn.stderr[-3000:]
    a, b = NP.load(npz1), NP.load(npzn)
    assert sorted(a.files) == sorted(b.files) and 'skyvis_lag' in a.files and 'vis_noise_freq' in a.files
    assert a['skyvis_freq'].shape == (171, 32, 3)
    for key in a.files:
        if a[key].dtype.kind in 'fc':
            scale = float(NP.max(NP.abs(a[key]))) or 1.0
            assert NP.max(NP.abs(a[key] - b[key])) <= 1e-11 * scale, key
        else:
            assert NP.array_equal(a[key], b[key]), key


def test_run_prisim_n_ranks_gathers_the_baseline_gradients(fake_rccl, tmp_path):
    from prisim_amd import hdf5io
    try:
        hdf5io._load()
    except hdf5io.HDF5Unavailable:
        pytest.skip('libhdf5 not available')
    script = os.path.join(ROOT, 'scripts', 'run_prisim.py')
    extra = {'processing': {'gradient_mode': 'baseline', 'delay_transform': False, 'add_noise': False}, 'obsparm': {'n_acc': 2},
             'save_formats': {'npz': True, 'hdf5': True}}
    y1, npz1 = _parms(tmp_path, 'gone', extra)
    yn, npzn = _parms(tmp_path, 'gmany', extra)
    for cmd in ([sys.executable, script, '-i', y1], [sys.executable, script, '-n', '2', '-i', yn]):
        r = subprocess.run(cmd, env=_env(fake_rccl), cwd=ROOT, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
    grads = []
    for npz in (npz1, npzn):
        with hdf5io.File(npz.replace('.npz', '.hdf5'), 'r') as f:
            grads.append(NP.asarray(f.read('gradients/baseline')))
    assert grads[0].shape == (3, 171, 32, 2) and grads[1].shape == grads[0].shape
    assert NP.max(NP.abs(grads[0] - grads[1])) <= 1e-11 * NP.max(NP.abs(grads[0])) and NP.max(NP.abs(grads[0])) > 0
