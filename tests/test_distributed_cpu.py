"""CPU: world_size-2 gloo rehearsal of the multi-process path (sharding + id broadcast + gather)."""
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


def test_two_rank_gloo_shard_and_gather():
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', OMP_NUM_THREADS='2')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(ROOT, 'tests', 'dist_worker.py')]
    res = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    out = res.stdout + res.stderr
    assert res.returncode == 0, out[-3000:]
    assert 'RANK 0 OK' in out and 'RANK 1 OK' in out, out[-3000:]
