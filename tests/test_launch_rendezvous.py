"""CPU: the torch-free launcher (prisim_amd.launch) and the hardened socket rendezvous (prisim_amd.rendezvous) -- what stands where
the reference has `mpirun` + mpi4py (README.rst:93-99; scripts/run_prisim.py:864-880, 2211, 2233-2242)."""
import os
import socket
import stat
import subprocess
import sys
import textwrap
import threading
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from prisim_amd import launch, rendezvous  # noqa: E402


def _clean_env():
    env = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'PRISIM_RDZV_FILE', 'MASTER_PORT', 'XDG_RUNTIME_DIR'):
        env.pop(k, None)
    return env


def _script(tmp_path, body):
    p = tmp_path / 'rank_script.py'
    p.write_text('import os, sys\nsys.path.insert(0, %r)\n' % ROOT + textwrap.dedent(body))
    return str(p)


def test_launcher_sets_the_rank_environment_and_returns_zero(tmp_path):
    script = _script(tmp_path, '''
        from prisim_amd import rendezvous
        r = rendezvous.Rendezvous()
        got = r.allgather({'rank': r.rank, 'local': int(os.environ['LOCAL_RANK']), 'world': int(os.environ['WORLD_SIZE'])})
        blob = r.allgather_bytes(bytes([r.rank]) * (r.rank + 1))
        assert blob == [bytes([k]) * (k + 1) for k in range(r.world)]
        uid = r.broadcast_bytes(b'u' * 128 if r.rank == 0 else b'')
        assert uid == b'u' * 128 and r.allreduce_max(r.rank) == r.world - 1 and r.allreduce_min(r.rank) == 0
        assert os.environ['MASTER_ADDR'] == '127.0.0.1' and os.environ['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'
        st = os.stat(os.environ['PRISIM_RDZV_FILE']) if r.rank != 0 else None
        r.barrier()
        if r.rank == 0:
            print('GOT', sorted((g['rank'], g['local'], g['world']) for g in got))
        r.close()
    ''')
    res = subprocess.run([sys.executable, '-m', 'prisim_amd.launch', '-n', '3', script], env=_clean_env(), cwd=ROOT, capture_output=True, text=True,
                         timeout=120)
    assert res.returncode == 0, res.stderr[-2000:]
    assert 'GOT [(0, 0, 3), (1, 1, 3), (2, 2, 3)]' in res.stdout


def test_launcher_stops_the_job_when_one_rank_fails(tmp_path):
    script = _script(tmp_path, '''
        import time
        from prisim_amd import rendezvous
        r = rendezvous.Rendezvous()
        if r.rank == 1:
            sys.exit(7)
        r.barrier()            # the others would wait here for ever: the launcher must end them
        time.sleep(60)
    ''')
    t0 = time.time()
    res = subprocess.run([sys.executable, '-m', 'prisim_amd.launch', '-n', '3', script], env=_clean_env(), cwd=ROOT, capture_output=True, text=True,
                         timeout=120)
    assert res.returncode == 7, (res.returncode, res.stderr[-2000:])
    assert time.time() - t0 < 40.0 and 'rank 1 exited with 7' in res.stderr


def test_rendezvous_file_is_private_and_foreign_or_planted_files_are_refused(tmp_path, monkeypatch):
    monkeypatch.delenv('XDG_RUNTIME_DIR', raising=False)
    monkeypatch.setenv('TMPDIR', str(tmp_path))
    import tempfile
    tempfile.tempdir = None                                    # re-read TMPDIR
    try:
        d = rendezvous.private_dir()
        st = os.lstat(d)
        assert stat.S_ISDIR(st.st_mode) and st.st_uid == os.getuid() and not (st.st_mode & 0o077)
        os.chmod(d, 0o755)                                     # somebody loosened it: refused, not silently used
        with pytest.raises(PermissionError):
            rendezvous.private_dir()
        os.chmod(d, 0o700)
        f = os.path.join(d, 'rdzv_test')
        rendezvous._publish(f, '1234 abcd\n')
        assert stat.S_IMODE(os.lstat(f).st_mode) == 0o600 and rendezvous._read_published(f) == '1234 abcd\n'
        # a symlink planted at the name is not followed by readers ...
        target = tmp_path / 'elsewhere'
        target.write_text('9 evil\n')
        link = os.path.join(d, 'rdzv_link')
        os.symlink(str(target), link)
        with pytest.raises(OSError):
            rendezvous._read_published(link)
        # ... and the writer replaces it instead of writing through it
        rendezvous._publish(link, '5678 good\n')
        assert not os.path.islink(link) and target.read_text() == '9 evil\n' and rendezvous._read_published(link) == '5678 good\n'
    finally:
        tempfile.tempdir = None


def test_rank0_survives_stray_connections_and_duplicate_ranks(tmp_path):
    """Anything can find an open loopback port: garbage, a silent peer, a wrong nonce, an out-of-range or duplicate rank are dropped and
    the accept loop goes on (they used to take rank 0 down or block it for the full timeout)."""
    path = str(tmp_path / 'rdzv')
    os.environ['PRISIM_RDZV_FILE'] = path
    result = {}

    def rank0():
        try:
            r = rendezvous.Rendezvous(0, 2, timeout=60.0)
            result['gather'] = r.allgather('zero')
            r.close()
        except Exception as exc:                                # noqa
            result['error'] = repr(exc)

    th = threading.Thread(target=rank0)
    try:
        th.start()
        t0 = time.time()
        while not os.path.exists(path) and time.time() - t0 < 10:
            time.sleep(0.01)
        port, nonce = rendezvous._read_published(path).split()
        import struct
        for payload in (b'\x00\x01garbage', None,
                        struct.pack('<Q', 12 + 4) + b'PRSM' + struct.pack('<I', 1) + b'wrongnonce12',
                        struct.pack('<Q', 8 + len(nonce)) + b'PRSM' + struct.pack('<I', 9) + nonce.encode(),      # rank out of range
                        struct.pack('<Q', 1 << 40)):                                                            # absurd length word
            s = socket.create_connection(('127.0.0.1', int(port)), timeout=5)
            if payload is not None:
                s.sendall(payload)
            s.close()
        r1 = rendezvous.Rendezvous(1, 2, timeout=60.0)
        assert r1.allgather('one') == ['zero', 'one']
        r1.close()
        th.join(30)
        assert result.get('gather') == ['zero', 'one'], result
    finally:
        os.environ.pop('PRISIM_RDZV_FILE', None)
        th.join(1)


def test_collectives_do_not_time_out_while_a_rank_is_busy(tmp_path):
    """Rank 0 writes its files while the others already wait in the barrier (driver.main): the wait is blocking, not a 10-minute timer."""
    path = str(tmp_path / 'rdzv')
    os.environ['PRISIM_RDZV_FILE'] = path
    out = {}

    def rank1():
        r = rendezvous.Rendezvous(1, 2, timeout=30.0)
        out['timeout'] = r._sock.gettimeout()
        r.barrier()
        r.close()

    th = threading.Thread(target=rank1)
    try:
        th.start()
        r0 = rendezvous.Rendezvous(0, 2, timeout=30.0)
        assert all(c.gettimeout() is None for c in r0._peers)
        time.sleep(0.3)
        r0.barrier()
        r0.close()
        th.join(30)
        assert out['timeout'] is None
    finally:
        os.environ.pop('PRISIM_RDZV_FILE', None)
        th.join(1)


def test_launcher_takes_its_ranks_along_when_it_is_terminated(tmp_path):
    """SIGTERM to the launcher (a batch system's timeout) must not leave rank processes behind on the GPUs."""
    import signal
    marker = tmp_path / 'pids'
    script = _script(tmp_path, '''
        import time
        with open(%r, 'a') as f:
            f.write('%%d\\n' %% os.getpid())
        time.sleep(120)
    ''' % str(marker))
    p = subprocess.Popen([sys.executable, '-m', 'prisim_amd.launch', '-n', '2', script], env=_clean_env(), cwd=ROOT, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, text=True)
    t0 = time.time()
    while time.time() - t0 < 30 and (not marker.exists() or len(marker.read_text().split()) < 2):
        time.sleep(0.05)
    pids = [int(x) for x in marker.read_text().split()]
    assert len(pids) == 2
    p.send_signal(signal.SIGTERM)
    assert p.wait(timeout=30) == 130
    time.sleep(0.2)
    for pid in pids:
        with pytest.raises(OSError):
            os.kill(pid, 0)                                   # gone


def test_a_rank_stuck_in_comm_init_ends_the_job_with_a_message(tmp_path):
    """VERDICT r5 next #3a: ncclCommInitRank waits for every rank.  Rank 1 never arrives (it sleeps); rank 0 sits inside its comm_init (a
    blocking call standing in for the collective).  PRISIM_COMM_TIMEOUT_S = 2: rank 0 prints who and where it is -- rank, device, the
    step, librccl's last error text -- and exits 124; the launcher reaps rank 1; the job is over in seconds, not at a driver's timeout."""
    script = _script(tmp_path, '''
        import threading, time
        from prisim_amd import rendezvous, watchdog
        r = rendezvous.Rendezvous()
        r.barrier()
        if r.rank == 1:
            time.sleep(600)                       # "died before its comm_init"
        never = threading.Event()
        with watchdog.CommDeadline(r.rank, 0, describe=lambda: '0000:05:00.0', last_error=lambda: 'unhandled system error (stand-in)') as dl:
            dl.step('ncclCommInitRank (prisim_hip_comm_init)')
            never.wait()                          # a C call that never returns, GIL released
        print('NOT REACHED')
    ''')
    env = _clean_env()
    env['PRISIM_COMM_TIMEOUT_S'] = '2'
    t0 = time.time()
    res = subprocess.run([sys.executable, '-m', 'prisim_amd.launch', '-n', '2', script], env=env, cwd=ROOT, capture_output=True, text=True, timeout=120)
    took = time.time() - t0
    assert res.returncode != 0 and 'NOT REACHED' not in res.stdout
    assert took < 40.0, took
    err = res.stderr
    assert 'rank 0 (device 0, PCI 0000:05:00.0)' in err and 'PRISIM_COMM_TIMEOUT_S = 2 s' in err
    assert 'stuck in step "ncclCommInitRank (prisim_hip_comm_init)"' in err
    assert 'last RCCL error: unhandled system error (stand-in)' in err and 'exiting with code 124' in err


def test_comm_deadline_is_silent_when_the_block_finishes_and_can_be_switched_off(monkeypatch):
    from prisim_amd import watchdog
    fired = []
    with watchdog.CommDeadline(3, 1, seconds=0.2, _exit=lambda code: fired.append(code)) as dl:
        dl.step('quick')
    time.sleep(0.4)
    assert fired == []
    with watchdog.CommDeadline(3, 1, seconds=0.1, describe=lambda: 1 / 0, _exit=lambda code: fired.append(code)) as dl:   # a failing diagnostic is no failure
        dl.step('slow')
        time.sleep(0.5)
    assert fired == [watchdog.EXIT_CODE]
    monkeypatch.setenv('PRISIM_COMM_TIMEOUT_S', '0')
    assert watchdog.timeout_seconds() is None
    with watchdog.CommDeadline(0, 0) as dl:
        assert dl._timer is None
    monkeypatch.setenv('PRISIM_COMM_TIMEOUT_S', 'soon')
    assert watchdog.timeout_seconds() == 120.0
