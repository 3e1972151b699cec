"""CPU: world_size-2 rehearsal of the PRODUCT's multi-rank path (prisim_amd.rendezvous + prisim_amd.driver.run with baseline
shards, padded last shard, all-gather of the cube and of the delay spectra), gloo standing in for RCCL at the Context seam."""
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
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', OMP_NUM_THREADS='2')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(nproc), '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(ROOT, 'tests', 'dist_worker.py'), mode]
    res = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
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
def test_two_ranks_on_one_gpu_real_kernels_gloo_exchange():
    """Both ranks on device 0 with the real HIP context (sky-sum kernels, device-resident cube and delay spectra); only the
    communicator is the gloo stand-in (RCCL refuses two ranks on one GPU)."""
    _run_workers('gpu', 2)
