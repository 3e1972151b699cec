"""A deadline around the steps of a multi-GPU run that every rank must reach together: ncclCommInitRank, the communicator self-test,
and the rendezvous collectives right after them.  Those calls block for as long as a peer stays away -- a rank that died before its
comm_init, a GPU that never came up, a fabric that does not connect -- and the reference's MPI runs (scripts/run_prisim.py:99-102) hang
the same way.  Here the first N > 1 run on new hardware must be diagnosable from its record: when PRISIM_COMM_TIMEOUT_S (default 120)
pass without the guarded block finishing, the rank prints who it is (rank, device, PCI bus id), the step it is in, how long it has
been there and librccl's last error text, and EXITS non-zero -- a fresh exit, never a re-exec of a process that has touched the GPU.
The launcher (prisim_amd.launch, torchrun) sees the exit and stops the peers.

    with CommDeadline(rank, device) as dl:
        dl.step('ncclCommInitRank')
        ctx.comm_init(uid, world, rank)
        dl.step('self-test all-gather')
        ctx.comm_selftest()
        dl.step('rendezvous: self-test outcomes')
        outcomes = rdzv.allgather(ok)
"""
import os
import sys
import threading
import time

EXIT_CODE = 124                 # what timeout(1) uses


def timeout_seconds():
    try:
        t = float(os.environ.get('PRISIM_COMM_TIMEOUT_S', '120'))
    except ValueError:
        t = 120.0
    return t if t > 0 else None          # <= 0 switches the deadline off


class CommDeadline(object):
    def __init__(self, rank, device, seconds=None, describe=None, last_error=None, _exit=os._exit):
        self.rank, self.device = int(rank), int(device)
        self.seconds = timeout_seconds() if seconds is None else seconds
        self._describe = describe            # () -> str: PCI bus id etc.; evaluated NOW, on the caller's thread
        self._last_error = last_error        # () -> str: librccl's last error; evaluated at expiry, on the timer thread
        self._exit = _exit
        self._where, self._since = 'start', time.time()
        self._timer = None
        self._what = ''

    def __enter__(self):
        if self.seconds is None:
            return self
        if self._describe is not None:
            try:
                self._what = str(self._describe())
            except Exception as exc:                      # a diagnostic must never be the failure
                self._what = 'unavailable ({0})'.format(exc)
        self._t0 = time.time()
        self._timer = threading.Timer(self.seconds, self._expire)
        self._timer.daemon = True
        self._timer.start()
        return self

    def step(self, name):
        self._where, self._since = str(name), time.time()

    def __exit__(self, *exc):
        if self._timer is not None:
            self._timer.cancel()
        return False

    def _expire(self):
        now = time.time()
        msg = ('[prisim_amd] rank {0} (device {1}, PCI {2}): communicator setup made no progress within PRISIM_COMM_TIMEOUT_S = {3:g} s; '
               'stuck in step "{4}" for {5:.1f} s.  A peer has not reached the same step (died, hung, or cannot be reached over the fabric).'
               '\n').format(self.rank, self.device, self._what or 'unknown', self.seconds, self._where, now - self._since)
        sys.stderr.write(msg)
        sys.stderr.flush()
        # librccl's text may sit behind a lock the blocked call holds: ask on a thread of its own and leave regardless after 5 s
        def ask():
            try:
                if self._last_error is not None:
                    sys.stderr.write('[prisim_amd] rank {0}: last RCCL error: {1}\n'.format(self.rank, self._last_error()))
                    sys.stderr.flush()
            except Exception:
                pass
        t = threading.Thread(target=ask)
        t.daemon = True
        t.start()
        t.join(5.0)
        sys.stderr.write('[prisim_amd] rank {0}: exiting with code {1}\n'.format(self.rank, EXIT_CODE))
        sys.stderr.flush()
        self._exit(EXIT_CODE)


def for_context(rank, device, abi):
    """A CommDeadline that describes the device through the C-ABI (`abi` = prisim_amd._abi)."""
    ctx_cls = abi.Context
    return CommDeadline(rank, device, describe=lambda: ctx_cls.device_pci(device), last_error=getattr(ctx_cls, 'comm_last_error', None))
