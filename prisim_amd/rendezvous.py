"""Single-node process rendezvous without torch: the few host-side collectives a one-process-per-GPU run needs before and
around its GPU work -- hand the 128-byte RCCL unique id to every rank, barrier, max / min reduce of a scalar, gather of small
python objects.  Replaces mpi4py's COMM_WORLD in scripts/run_prisim.py (:864-880 rank / size, :2211 barrier, :2233-2242 gather
at rank 0) for runs launched as `python -m torch.distributed.run --nproc-per-node N ...` (or any launcher that sets RANK /
WORLD_SIZE): the launcher only provides the environment, nothing of torch is imported here, so the RCCL the library loads is
the ROCm one it was compiled against.

Transport: rank 0 listens on an ephemeral TCP port of 127.0.0.1 and publishes "port nonce" in a file that every rank of the
launch can name (same parent process = the launcher's agent, same MASTER_PORT); the others connect, and every collective is a
star through rank 0 (N <= 8 ranks, payloads of bytes: microseconds).  No GPU call is made here, and none must be made by
the caller before `Rendezvous(...)` returns on the ranks that fork nothing afterwards -- sockets only.
"""
import os
import pickle
import socket
import struct
import tempfile
import time

_MAGIC = b'PRSM'


def _send(sock, payload):
    sock.sendall(struct.pack('<I', len(payload)) + payload)


def _recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(n - len(buf))
        if not chunk:
            raise ConnectionError('rendezvous peer closed the connection')
        buf += chunk
    return bytes(buf)


def _recv(sock):
    (n,) = struct.unpack('<I', _recv_exact(sock, 4))
    return _recv_exact(sock, n)


def default_key():
    """Names the launch: every rank of one `torch.distributed.run` has the same parent (the agent) and MASTER_PORT."""
    return '%s_%s_%s' % (os.environ.get('TORCHELASTIC_RUN_ID', 'none'), os.environ.get('MASTER_PORT', '0'), os.getppid())


class Rendezvous(object):
    def __init__(self, rank=None, world=None, key=None, timeout=600.0):
        self.rank = int(os.environ.get('RANK', '0')) if rank is None else int(rank)
        self.world = int(os.environ.get('WORLD_SIZE', '1')) if world is None else int(world)
        self.timeout = float(timeout)
        self._peers = []
        self._sock = None
        self._path = None
        if self.world <= 1:
            return
        path = os.environ.get('PRISIM_RDZV_FILE')
        if not path:
            path = os.path.join(tempfile.gettempdir(), 'prisim_rdzv_%d_%s' % (os.getuid(), key or default_key()))
        self._path = path
        deadline = time.time() + self.timeout
        if self.rank == 0:
            try:
                os.unlink(path)
            except OSError:
                pass
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind(('127.0.0.1', 0))
            srv.listen(self.world)
            nonce = os.urandom(8).hex()
            tmp = path + '.tmp%d' % os.getpid()
            with open(tmp, 'w') as f:
                f.write('%d %s\n' % (srv.getsockname()[1], nonce))
            os.replace(tmp, path)                     # atomic: a reader sees nothing or the whole line
            peers = {}
            srv.settimeout(1.0)
            while len(peers) < self.world - 1:
                if time.time() > deadline:
                    raise TimeoutError('rendezvous: only %d of %d ranks connected' % (len(peers) + 1, self.world))
                try:
                    conn, _ = srv.accept()
                except socket.timeout:
                    continue
                conn.settimeout(self.timeout)
                conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                hello = _recv(conn)
                if hello[:4] != _MAGIC or hello[8:].decode() != nonce:
                    conn.close()                      # something else found the port, or a rank of an older launch
                    continue
                (r,) = struct.unpack('<I', hello[4:8])
                peers[r] = conn
            srv.close()
            self._peers = [peers[r] for r in range(1, self.world)]
            for c in self._peers:
                _send(c, b'ok')
        else:
            while True:
                if time.time() > deadline:
                    raise TimeoutError('rendezvous: rank %d could not reach rank 0 through %s' % (self.rank, path))
                try:
                    with open(path) as f:
                        port_s, nonce = f.read().split()
                    s = socket.create_connection(('127.0.0.1', int(port_s)), timeout=2.0)
                    s.settimeout(self.timeout)
                    s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    _send(s, _MAGIC + struct.pack('<I', self.rank) + nonce.encode())
                    if _recv(s) == b'ok':
                        self._sock = s
                        break
                    s.close()
                except (OSError, ValueError, ConnectionError):
                    time.sleep(0.02)                  # file not there yet, or stale: read it again

    # ---- collectives (star through rank 0) ----
    def _exchange(self, payload, combine):
        """Every rank contributes `payload` (bytes); rank 0 applies combine(list of payloads) -> bytes, everyone gets it."""
        if self.world <= 1:
            return combine([payload])
        if self.rank == 0:
            parts = [payload] + [_recv(c) for c in self._peers]
            result = combine(parts)
            for c in self._peers:
                _send(c, result)
            return result
        _send(self._sock, payload)
        return _recv(self._sock)

    def barrier(self):
        self._exchange(b'', lambda parts: b'')

    def broadcast_bytes(self, data, src=0):
        if src != 0:
            raise ValueError('only rank 0 broadcasts')
        return self._exchange(data if self.rank == 0 else b'', lambda parts: parts[0])

    def allgather(self, obj):
        """List of every rank's (picklable, small) object, in rank order."""
        return pickle.loads(self._exchange(pickle.dumps(obj), lambda parts: pickle.dumps([pickle.loads(p) for p in parts])))

    def allreduce_max(self, x):
        return max(self.allgather(float(x)))

    def allreduce_min(self, x):
        return min(self.allgather(float(x)))

    def close(self):
        for c in self._peers:
            try:
                c.close()
            except OSError:
                pass
        self._peers = []
        if self._sock is not None:
            try:
                self._sock.close()
            except OSError:
                pass
            self._sock = None
        if self.rank == 0 and self._path:
            try:
                os.unlink(self._path)
            except OSError:
                pass
