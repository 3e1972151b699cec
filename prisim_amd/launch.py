"""One process per GPU without torch: the `mpirun -n N` of the reference's README.rst:93-99 for a single node.

    python -m prisim_amd.launch -n 8 scripts/run_prisim.py -i parms.yaml
    python bench.py --gpus 8                 (bench.py and scripts/run_prisim.py -n N call spawn_ranks themselves)

The parent starts N children with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set -- what the ranks read, and what
torch.distributed.run would have provided -- plus PRISIM_RDZV_FILE inside a private directory (mkdtemp, 0700) for
prisim_amd.rendezvous.  It never touches HIP: children are started with subprocess (fork + exec of a fresh interpreter) before any
GPU call exists in this process, so no GPU-initialised process is ever replaced or forked.  stdout / stderr are inherited: rank 0's
single JSON line (bench.py) reaches the caller unchanged.  The first child to fail ends the job: the others get SIGTERM (then
SIGKILL), addressed by the exact PIDs started here, and the parent exits with that child's code -- a rank stuck in a collective
with a dead peer cannot hang the run.
"""
import os
import shutil
import signal
import socket
import subprocess
import sys
import tempfile
import time


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def rank_env(rank, world, base=None, port=None, rdzv_file=None):
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1')
    if port is not None:
        env['MASTER_PORT'] = str(port)
    if rdzv_file is not None:
        env['PRISIM_RDZV_FILE'] = rdzv_file
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')       # dmabuf IPC: what RCCL's peer-to-peer setup needs on this driver
    env.setdefault('NCCL_SOCKET_IFNAME', 'lo')              # single node: RCCL bootstraps over loopback
    return env


def spawn_ranks(nranks, argv, env=None, grace=10.0):
    """Run `argv` (a full command line, argv[0] = the interpreter) as `nranks` processes; returns the job's exit code:
    0 when every rank returned 0, else the code of the first rank that failed (the rest are terminated)."""
    nranks = int(nranks)
    if nranks < 1:
        raise ValueError('nranks must be >= 1')
    tmpdir = tempfile.mkdtemp(prefix='prisim_launch_')      # 0700 by construction
    rdzv_file = os.path.join(tmpdir, 'rdzv')
    port = _free_port()
    procs = []

    def _on_term(signum, frame):            # the launcher itself is being stopped: take the ranks along (no orphans on the GPUs)
        raise KeyboardInterrupt

    old_term = None
    try:
        old_term = signal.signal(signal.SIGTERM, _on_term)
    except ValueError:                      # not the main thread: the caller's signal handling stands
        pass
    try:
        for r in range(nranks):
            procs.append(subprocess.Popen(list(argv), env=rank_env(r, nranks, env, port, rdzv_file)))
        code = 0
        alive = {p.pid: r for r, p in enumerate(procs)}
        def reap_one():
            """(pid, status) of one of OUR ranks that has ended, or None: only the PIDs started here are polled (os.waitpid(pid, WNOHANG)),
            so an exit status of any other child of the calling process -- another Popen, a pool, a test harness -- is never consumed."""
            for pid in list(alive):
                try:
                    got, status = os.waitpid(pid, os.WNOHANG)
                except ChildProcessError:
                    alive.pop(pid, None)
                    continue
                if got == pid:
                    return pid, status
            return None

        while alive:
            # ranks are reaped in the order they end (polled every few ms), so the rank reported -- and whose code the job returns -- is
            # the one that failed FIRST, not a peer that died of the broken connection a moment later
            got = reap_one()
            if got is None:
                time.sleep(0.005)
                continue
            pid, status = got
            r = alive.pop(pid)
            rc = os.waitstatus_to_exitcode(status)
            procs[r].returncode = rc
            if rc != 0:
                # A rank that dies takes its peers' sockets with it, and a peer can finish dying of the broken connection (exit 1, a
                # traceback) BEFORE the culprit's interpreter has finished shutting down.  So look at who else has gone within a
                # moment and report the rank with the most specific code: one that is not the generic 1, else the first reaped.
                failed = [(r, rc)]
                t_end = time.time() + 0.5
                while alive and time.time() < t_end:
                    got2 = reap_one()
                    if got2 is None:
                        time.sleep(0.01)
                        continue
                    pid2, status2 = got2
                    r2 = alive.pop(pid2)
                    rc2 = os.waitstatus_to_exitcode(status2)
                    procs[r2].returncode = rc2
                    if rc2 != 0:
                        failed.append((r2, rc2))
                r, rc = next(((a, b) for a, b in failed if b != 1), failed[0])
                code = rc if rc > 0 else 128 - rc          # killed by signal s: 128 + s, as a shell reports it
                sys.stderr.write('prisim_amd.launch: rank %d exited with %d; stopping the other ranks\n' % (r, rc))
                _stop(procs, set(alive.values()), grace)
                alive.clear()
        return code
    except KeyboardInterrupt:
        _stop(procs, set(range(len(procs))), grace)
        return 130
    finally:
        if old_term is not None:
            signal.signal(signal.SIGTERM, old_term)
        shutil.rmtree(tmpdir, ignore_errors=True)


def _stop(procs, which, grace):
    for r in which:
        if procs[r].poll() is None:
            try:
                procs[r].send_signal(signal.SIGTERM)
            except OSError:
                pass
    t0 = time.time()
    for r in which:
        try:
            procs[r].wait(timeout=max(0.1, grace - (time.time() - t0)))
        except subprocess.TimeoutExpired:
            try:
                procs[r].kill()
            except OSError:
                pass
            procs[r].wait()


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(prog='python -m prisim_amd.launch', description='start N ranks of a script, one per GPU (no torch, no MPI)')
    ap.add_argument('-n', '--nranks', type=int, required=True)
    ap.add_argument('script')
    ap.add_argument('args', nargs=argparse.REMAINDER)
    a = ap.parse_args(argv)
    return spawn_ranks(a.nranks, [sys.executable, a.script] + a.args)


if __name__ == '__main__':
    sys.exit(main())
