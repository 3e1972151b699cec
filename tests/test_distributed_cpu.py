"""CPU: world_size-2 / -3 rehearsal of the PRODUCT's multi-rank path (prisim_amd.launch + prisim_amd.rendezvous + prisim_amd.driver.run
with baseline shards, padded last shard, all-gather of the cube, of the delay spectra and of the gradient cube), a host exchange over the
rendezvous sockets standing in for RCCL at the Context seam.  No torch anywhere."""
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run_workers(mode, nproc=2):
    env = dict(os.environ, OMP_NUM_THREADS='2')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'PRISIM_RDZV_FILE'):
        env.pop(k, None)
    cmd = [sys.executable, '-m', 'prisim_amd.launch', '-n', str(nproc), os.path.join(ROOT, 'tests', 'dist_worker.py'), mode]
    res = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    out = res.stdout + res.stderr
    assert res.returncode == 0, out[-3000:]
    for r in range(nproc):
        assert 'RANK %d OK' % r in out, out[-3000:]


def test_two_rank_driver_run_shards_gathers_and_delay_transforms():
    _run_workers('oracle', 2)


def test_three_rank_driver_run_uneven_shards():
    _run_workers('oracle', 3)


def test_rendezvous_single_rank_is_trivial():
    sys.path.insert(0, ROOT)
    from prisim_amd import rendezvous
    r = rendezvous.Rendezvous(0, 1)
    assert r.broadcast_bytes(b'abc') == b'abc' and r.allreduce_max(2.5) == 2.5 and r.allgather('x') == ['x']
    r.barrier()
    r.close()


import pytest  # noqa: E402


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_real_kernels_host_exchange():
    """Both ranks on device 0 with the real HIP context (sky-sum kernels, device-resident cube, gradient cube and delay spectra); only the
    communicator is the host stand-in (RCCL refuses two ranks on one GPU)."""
    _run_workers('gpu', 2)


# ---- bench.py itself at world size > 1 (the driver's scaling runs are the first time it meets more than one GPU) ----
def _bench_ranks(tmp_path, nproc, mode='ok', extra=()):
    import json
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()), WORLD_SIZE=str(nproc), OMP_NUM_THREADS='1',
               BENCH_REHEARSAL_DIR=str(tmp_path), BENCH_REHEARSAL_MODE=mode, PRISIM_RDZV_FILE=str(tmp_path / 'rdzv'))
    procs = []
    for r in range(nproc):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, 'tests', 'bench_worker.py'), '--gpus', str(nproc), '--steps', '2',
                                       '--warmup', '1', '--nsrc', '16', '--no-cpu-baseline'] + list(extra),
                                      env=e, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    codes = [p.returncode for p in procs]
    lines = [[ln for ln in o[0].splitlines() if ln.strip()] for o in outs]
    return codes, lines, [o[1] for o in outs], json


def test_bench_main_at_world_size_three_prints_one_contract_line(tmp_path):
    codes, lines, errs, json = _bench_ranks(tmp_path, 3)
    assert codes == [0, 0, 0], errs
    assert len(lines[0]) == 1 and lines[1] == [] and lines[2] == []            # rank 0 prints ONE line, the others nothing
    d = json.loads(lines[0][0])
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
                'dtype', 'data', 'config', 'roofline'):
        assert key in d, key
    assert d['n_gpus'] == 3 and d['steps'] == 2 and d['warmup'] == 1 and d['gather_ok'] is True
    assert d['metric'] == 'visibility-terms/sec' and d['scaling'] == 'strong' and d['vs_baseline'] is None
    assert d['config']['nbl'] == 61075 and d['config']['nsrc'] == 16 and d['config']['sharding'].startswith('baselines/3')
    # whole-job value: all 61 075 baselines (not the padded 3 x 20 480), both steps, over the slowest rank's time
    assert abs(d['value'] * d['ms_per_step'] * 1e-3 * 2 - 61075.0 * 1024 * 16 * 2) <= 1e-6 * 61075.0 * 1024 * 16 * 2
    sys.path.insert(0, ROOT)
    from prisim_amd import sharding
    assert sharding.shard_size(61075, 3) == 20480                              # 239 groups of 256 dealt round-robin: 80, 80, 79
    assert d['roofline']['terms_per_launch'] == 20480.0 * 1024 * 16            # per launch: this rank's padded shard


def test_bench_launched_bare_spawns_its_own_ranks_and_reports_the_gather(tmp_path):
    """`python bench.py --gpus 3` with no launcher: bench.main() becomes the launcher (prisim_amd.launch.spawn_ranks), its three children
    run the rank code, rank 0's ONE line comes through, and the N > 1 keys of the contract are there."""
    import json
    env = dict(os.environ, OMP_NUM_THREADS='1', BENCH_REHEARSAL_DIR=str(tmp_path), BENCH_REHEARSAL_MODE='ok')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'PRISIM_RDZV_FILE', 'MASTER_PORT'):
        env.pop(k, None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'bench_worker.py'), '--gpus', '3', '--steps', '2', '--warmup', '1', '--nsrc', '16',
                          '--no-cpu-baseline'], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, res.stdout
    d = json.loads(lines[0])
    assert d['n_gpus'] == 3 and d['gather_ok'] is True and d['launcher'] == 'prisim_amd.launch'
    g = d['gather']
    for key in ('bytes_per_peer', 'per_snapshot_ms', 'exposed_ms', 'GBps_per_link', 'link_peak_GBps', 'comm_stream_priority'):
        assert key in g, key
    assert g['link_peak_GBps'] == 153.0 and g['bytes_per_peer'] > 0 and g['per_snapshot_ms'] > 0
    assert abs(g['GBps_per_link'] - g['bytes_per_peer'] / (g['per_snapshot_ms'] * 1e-3) / 1e9) < 1e-9 * g['GBps_per_link'] + 1e-12
    assert len(d['kernel_ms_per_rank']['all']) == 3 and d['kernel_ms_per_rank']['max'] >= d['kernel_ms_per_rank']['min']
    assert abs(d['value_n1_equiv'] * 3 - d['value']) <= 1e-9 * d['value']


def test_bench_launched_bare_exits_nonzero_when_a_rank_fails(tmp_path):
    env = dict(os.environ, OMP_NUM_THREADS='1', BENCH_REHEARSAL_DIR=str(tmp_path), BENCH_REHEARSAL_MODE='selftest_fails_on_0')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'PRISIM_RDZV_FILE', 'MASTER_PORT'):
        env.pop(k, None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'bench_worker.py'), '--gpus', '2', '--steps', '1', '--warmup', '0', '--nsrc', '16',
                          '--no-cpu-baseline'], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert res.returncode == 3, (res.returncode, res.stderr[-2000:])        # the RCCL self-test failed: exit 3, before any timed step
    assert [ln for ln in res.stdout.splitlines() if ln.strip()] == []       # and no value was printed
    assert 'self-test' in res.stderr


def test_bench_main_fails_loudly_when_the_communicator_cannot_be_made(tmp_path):
    for mode in ('init_fails_on_1', 'no_uid'):
        sub = tmp_path / mode
        sub.mkdir()
        codes, lines, errs, _ = _bench_ranks(sub, 2, mode=mode)
        assert codes == [3, 3], (mode, codes, errs)                             # every rank stops; no rank hangs in a collective
        assert lines == [[], []]                                                # and no value is printed


def test_bench_main_reports_a_gather_that_differs_between_ranks(tmp_path):
    codes, lines, errs, json = _bench_ranks(tmp_path, 2, mode='rank1_differs')
    assert codes == [0, 0], errs
    assert json.loads(lines[0][0])['gather_ok'] is False
