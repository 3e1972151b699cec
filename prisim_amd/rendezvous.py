"""Single-node process rendezvous without torch: the few host-side collectives a one-process-per-GPU run needs before and
around its GPU work -- hand the 128-byte RCCL unique id to every rank, barrier, max / min reduce of a scalar, gather of small
values.  Replaces mpi4py's COMM_WORLD in scripts/run_prisim.py (:864-880 rank / size, :2211 barrier, :2233-2242 gather
at rank 0).  The ranks are started by prisim_amd.launch (the `mpirun -n N` of README.rst:93-99) or by any launcher that sets
RANK / WORLD_SIZE (torch.distributed.run works: it only provides the environment, nothing of torch is imported here), so the
RCCL the library loads is the ROCm one it was compiled against.

Transport: rank 0 listens on an ephemeral TCP port of 127.0.0.1 and publishes "port nonce" in a file only this user can read or
plant: prisim_amd.launch creates a private directory (mkdtemp, 0700) and passes the path as PRISIM_RDZV_FILE; otherwise the file
lives in a per-user 0700 directory (XDG_RUNTIME_DIR or <tmp>/prisim_rdzv_<uid>, ownership and mode verified) under a name every
rank of the launch derives (same parent process, same MASTER_PORT).  The file is created with O_CREAT|O_EXCL|O_NOFOLLOW, mode 0600;
readers refuse a file that is not a regular file of their own uid.  Every collective is a star through rank 0 (N <= 8 ranks,
payloads of bytes: microseconds).  On the wire: raw bytes or JSON -- nothing is unpickled.  No GPU call is made here, and none
must be made by the caller before `Rendezvous(...)` returns on the ranks that fork nothing afterwards -- sockets only.
"""
import json
import os
import socket
import stat
import struct
import tempfile
import time

_MAGIC = b'PRSM'
_MAX_MSG = 1 << 34          # 16 GiB: test stand-ins move numpy cubes through here; anything larger is a corrupted length word


def _send(sock, payload):
    sock.sendall(struct.pack('<Q', len(payload)) + payload)


def _recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(min(n - len(buf), 1 << 24))
        if not chunk:
            raise ConnectionError('rendezvous peer closed the connection')
        buf += chunk
    return bytes(buf)


def _recv(sock, limit=_MAX_MSG):
    (n,) = struct.unpack('<Q', _recv_exact(sock, 8))
    if n > limit:
        raise ConnectionError('rendezvous message of %d bytes exceeds the limit' % n)
    return _recv_exact(sock, n)


def default_key():
    """Names the launch: every rank of one launcher has the same parent (prisim_amd.launch or torchrun's agent) and MASTER_PORT."""
    return '%s_%s_%s' % (os.environ.get('TORCHELASTIC_RUN_ID', 'none'), os.environ.get('MASTER_PORT', '0'), os.getppid())


def private_dir():
    """A directory only this user can write or list: XDG_RUNTIME_DIR when it is one, else <tmp>/prisim_rdzv_<uid> created 0700.
    Raises when the path exists but belongs to someone else, is a symlink, or is accessible to group / others."""
    uid = os.getuid()
    cand = os.environ.get('XDG_RUNTIME_DIR')
    if cand:
        try:
            st = os.lstat(cand)
            if stat.S_ISDIR(st.st_mode) and st.st_uid == uid and not (st.st_mode & 0o077):
                return cand
        except OSError:
            pass
    path = os.path.join(tempfile.gettempdir(), 'prisim_rdzv_%d' % uid)
    try:
        os.mkdir(path, 0o700)
    except FileExistsError:
        pass
    st = os.lstat(path)
    if not stat.S_ISDIR(st.st_mode) or st.st_uid != uid or (st.st_mode & 0o077):
        raise PermissionError('rendezvous directory %s is not a private directory of uid %d' % (path, uid))
    return path


def _publish(path, text):
    """Create `path` with `text` so that readers see nothing or the whole line: written to a sibling created O_EXCL|O_NOFOLLOW 0600,
    then renamed over the name (a stale file of an earlier launch is replaced; a symlink planted at the name is not followed)."""
    tmp = '%s.tmp%d' % (path, os.getpid())
    try:
        os.unlink(tmp)
    except OSError:
        pass
    fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL | getattr(os, 'O_NOFOLLOW', 0), 0o600)
    try:
        os.write(fd, text.encode())
    finally:
        os.close(fd)
    os.replace(tmp, path)


def _read_published(path):
    """The line rank 0 published -- only from a regular file owned by this user (no symlink, no foreign file)."""
    fd = os.open(path, os.O_RDONLY | getattr(os, 'O_NOFOLLOW', 0))
    try:
        st = os.fstat(fd)
        if not stat.S_ISREG(st.st_mode) or st.st_uid != os.getuid():
            raise PermissionError('rendezvous file %s is not a regular file of this user' % path)
        return os.read(fd, 256).decode()
    finally:
        os.close(fd)


class Rendezvous(object):
    """timeout: for connecting (seconds).  collective_timeout: for every later exchange; None (default) blocks -- a rank that writes a
    large file while the others wait in a barrier must not time them out, and a rank that dies takes its launcher's other ranks down
    (prisim_amd.launch and torchrun both end the job when one rank fails)."""

    def __init__(self, rank=None, world=None, key=None, timeout=600.0, collective_timeout=None):
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
            path = os.path.join(private_dir(), 'rdzv_%s' % (key or default_key()))
        self._path = path
        deadline = time.time() + self.timeout
        if self.rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind(('127.0.0.1', 0))
            srv.listen(max(self.world, 8))
            nonce = os.urandom(16).hex()
            _publish(path, '%d %s\n' % (srv.getsockname()[1], nonce))
            peers = {}
            srv.settimeout(1.0)
            while len(peers) < self.world - 1:
                if time.time() > deadline:
                    raise TimeoutError('rendezvous: only %d of %d ranks connected' % (len(peers) + 1, self.world))
                try:
                    conn, _ = srv.accept()
                except socket.timeout:
                    continue
                # anything may find an open loopback port: a connection that does not say the right thing within 5 s is dropped and the
                # loop goes on; it never takes rank 0 down
                try:
                    conn.settimeout(5.0)
                    hello = _recv(conn, limit=256)
                    if len(hello) < 8 or hello[:4] != _MAGIC or hello[8:].decode('ascii', 'replace') != nonce:
                        raise ValueError('not a rank of this launch')
                    (r,) = struct.unpack('<I', hello[4:8])
                    if not (1 <= r < self.world) or r in peers:
                        raise ValueError('rank %d out of range or already connected' % r)
                    conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    peers[r] = conn
                except (OSError, ValueError, struct.error, ConnectionError):
                    try:
                        conn.close()
                    except OSError:
                        pass
            srv.close()
            self._peers = [peers[r] for r in range(1, self.world)]
            for c in self._peers:
                c.settimeout(collective_timeout)
                _send(c, b'ok')
        else:
            while True:
                if time.time() > deadline:
                    raise TimeoutError('rendezvous: rank %d could not reach rank 0 through %s' % (self.rank, path))
                s = None
                try:
                    port_s, nonce = _read_published(path).split()
                    s = socket.create_connection(('127.0.0.1', int(port_s)), timeout=2.0)
                    s.settimeout(max(5.0, min(self.timeout, deadline - time.time())))
                    s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    _send(s, _MAGIC + struct.pack('<I', self.rank) + nonce.encode())
                    if _recv(s, limit=16) == b'ok':
                        s.settimeout(collective_timeout)
                        self._sock = s
                        break
                    s.close()
                except PermissionError:
                    raise
                except (OSError, ValueError, ConnectionError):
                    if s is not None:
                        try:
                            s.close()
                        except OSError:
                            pass
                    time.sleep(0.02)                  # file not there yet, or the stale one of an earlier launch: read it again

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

    def allgather_bytes(self, data):
        """List of every rank's byte string, in rank order."""
        def combine(parts):
            return struct.pack('<I', len(parts)) + b''.join(struct.pack('<Q', len(p)) for p in parts) + b''.join(parts)
        blob = self._exchange(bytes(data), combine)
        (n,) = struct.unpack_from('<I', blob, 0)
        sizes = struct.unpack_from('<%dQ' % n, blob, 4)
        out, off = [], 4 + 8 * n
        for sz in sizes:
            out.append(blob[off:off + sz])
            off += sz
        return out

    def allgather(self, obj):
        """List of every rank's small value (what JSON carries: numbers, booleans, None, strings, lists, dicts), in rank order."""
        return [json.loads(p.decode()) for p in self.allgather_bytes(json.dumps(obj).encode())]

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
