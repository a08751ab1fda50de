"""Monte-Carlo-axis data parallelism: one process per GPU, one RCCL all-reduce per call.

The reference is single-process; every estimator on the hot path is a mean (or a
weighted sum) over the ``num_mc_samples`` rows, so rank r evaluates the contiguous row block
``shard_rows(N, G, r)`` and the fp64 partial-sum vector ``[scalars | column sums]`` is
sum-all-reduced over RCCL/xGMI on the device before the O(P) epilogue
(``vb_comm_init`` in ``include/viabel_hip.h``).  The variational parameter is replicated.

The data path is RCCL only.  The control path -- handing the 128-byte RCCL unique id from rank 0 to
the other ranks, host barriers, a max over ranks of a host timing -- is a few bytes over plain TCP
sockets (:class:`SocketGroup`, rendezvous on ``MASTER_ADDR`` / ``MASTER_PORT`` as exported by
``torch.distributed.run``): no torch in the runtime path.  ``attach`` also accepts an initialised
``torch.distributed`` process group (any backend) for callers that already have one.
"""
import hashlib
import os
import socket
import struct
import time

import numpy as np

from . import _lib
from .objectives import shard_rows  # noqa: F401  (re-exported)

_MAGIC = b'VBAMD2\0\0'


def combine_partial_sums(partials):
    """Reference semantics of the device all-reduce: elementwise sum of the ranks' vectors."""
    return np.sum(np.stack([np.asarray(p, dtype=np.float64) for p in partials]), axis=0)


def _recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(n - len(buf))
        if not chunk:
            raise ConnectionError('peer closed the control connection')
        buf.extend(chunk)
    return bytes(buf)


def _send_msg(sock, payload):
    sock.sendall(struct.pack('<I', len(payload)) + payload)


def _recv_msg(sock):
    (n,) = struct.unpack('<I', _recv_exact(sock, 4))
    return _recv_exact(sock, n)


class SocketGroup:
    """Star-topology control group over TCP: rank 0 listens, ranks 1..G-1 connect.

    Every operation is collective and is a gather to rank 0 followed by a scatter of the result, so ranks leave
    it together (that is what makes it a barrier).  Payloads are a handful of bytes: this is the control plane
    only -- gradients never travel here.

    The listening port is ``MASTER_PORT + port_offset`` (torchrun's own store owns ``MASTER_PORT`` itself); if
    that port is taken rank 0 walks upwards and the other ranks probe the same sequence.  The handshake carries a
    16-byte **job token** (``from_env``: a hash of MASTER_ADDR, MASTER_PORT and the launcher's run id) that BOTH
    sides check, so two jobs on one host with neighbouring ports and the same world size cannot capture each
    other's ranks while both are starting up: a mismatching hello is dropped by the server, a mismatching reply
    makes the client move on to the next port.
    """

    HELLO_TIMEOUT = 2.0          # a connection that does not send its hello within this is dropped

    def __init__(self, rank, world, addr='127.0.0.1', port=29531, timeout=120.0, tries=16, token=b''):
        self.rank, self.world = int(rank), int(world)
        self._peers = []
        self._sock = None
        if self.world == 1:
            return
        token = hashlib.sha256(bytes(token)).digest()[:16]
        deadline = time.time() + timeout
        hello = _MAGIC + token + struct.pack('<ii', self.world, self.rank)
        reply = _MAGIC + token
        if self.rank == 0:
            srv = None
            for k in range(tries):
                srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                try:
                    srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                    srv.bind((addr, port + k))
                    break
                except OSError:
                    srv.close()
                    srv = None
            if srv is None:
                raise RuntimeError('no free control port in [%d, %d)' % (port, port + tries))
            srv.listen(self.world + 8)
            peers = {}
            try:
                while len(peers) < self.world - 1:
                    left = deadline - time.time()
                    if left <= 0:
                        raise socket.timeout()
                    srv.settimeout(left)
                    conn, _ = srv.accept()
                    conn.settimeout(self.HELLO_TIMEOUT)
                    try:
                        msg = _recv_exact(conn, len(hello))
                    except (ConnectionError, OSError):
                        conn.close()
                        continue
                    w, r = struct.unpack('<ii', msg[len(reply):])
                    if msg[:len(reply)] != reply or w != self.world or not 0 < r < self.world or r in peers:
                        conn.close()
                        continue
                    conn.sendall(reply)
                    conn.settimeout(timeout)
                    conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    peers[r] = conn
            except socket.timeout:
                for c in peers.values():
                    c.close()
                missing = [r for r in range(1, self.world) if r not in peers]
                raise RuntimeError('control rendezvous on %s:%d timed out after %.0f s: ranks %s never arrived'
                                   % (addr, srv.getsockname()[1], timeout, missing)) from None
            finally:
                srv.close()
            self._peers = [peers[r] for r in range(1, self.world)]
        else:
            k = 0
            while True:
                if time.time() > deadline:
                    raise RuntimeError('rank %d: no control server of this job on %s:%d..%d'
                                       % (self.rank, addr, port, port + tries - 1))
                s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                s.settimeout(5.0)
                try:
                    s.connect((addr, port + k % tries))
                    s.sendall(hello)
                    if _recv_exact(s, len(reply)) == reply:
                        s.settimeout(timeout)
                        s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                        self._sock = s
                        break
                except (OSError, ConnectionError):
                    pass
                s.close()
                k += 1
                if k % tries == 0:
                    time.sleep(0.05)

    @staticmethod
    def job_token(environ=None):
        """Bytes that identify THIS job among others on the host: MASTER_ADDR:MASTER_PORT plus the launcher's run id
        (``TORCHELASTIC_RUN_ID`` from torch.distributed.run, ``VIABEL_AMD_JOB_ID`` from bench.py's own launcher)."""
        env = os.environ if environ is None else environ
        return ('%s:%s|%s|%s' % (env.get('MASTER_ADDR', '127.0.0.1'), env.get('MASTER_PORT', '29500'),
                                 env.get('TORCHELASTIC_RUN_ID', ''), env.get('VIABEL_AMD_JOB_ID', ''))).encode()

    @classmethod
    def from_env(cls, port_offset=23, timeout=120.0):
        """Group described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (``torch.distributed.run`` exports them)."""
        world = int(os.environ.get('WORLD_SIZE', '1'))
        rank = int(os.environ.get('RANK', '0'))
        addr = os.environ.get('MASTER_ADDR', '127.0.0.1')
        port = int(os.environ.get('VIABEL_AMD_CONTROL_PORT', int(os.environ.get('MASTER_PORT', '29500')) + port_offset))
        return cls(rank, world, addr, port, timeout=timeout, token=cls.job_token())

    # ---- collectives (gather to rank 0, combine, scatter) ------------------------------------------------------
    def _exchange(self, payload, combine):
        if self.world == 1:
            return combine([payload])
        if self.rank == 0:
            parts = [payload] + [_recv_msg(c) for c in self._peers]
            out = combine(parts)
            for c in self._peers:
                _send_msg(c, out)
            return out
        _send_msg(self._sock, payload)
        return _recv_msg(self._sock)

    def barrier(self):
        self._exchange(b'', lambda parts: b'')

    def broadcast_bytes(self, payload=None):
        """Rank 0's ``payload`` on every rank."""
        return self._exchange(payload if self.rank == 0 else b'', lambda parts: parts[0])

    def allgather_bytes(self, payload):
        """Every rank's equal-length ``payload``, as a list in rank order, on every rank."""
        n = len(payload)
        out = self._exchange(payload, lambda parts: b''.join(parts))
        if len(out) != n * self.world:
            raise RuntimeError('allgather_bytes: ranks sent payloads of different lengths')
        return [out[i * n:(i + 1) * n] for i in range(self.world)]

    def allreduce_max(self, x):
        out = self._exchange(struct.pack('<d', float(x)),
                             lambda parts: struct.pack('<d', max(struct.unpack('<d', p)[0] for p in parts)))
        return struct.unpack('<d', out)[0]

    def allreduce_sum(self, x):
        out = self._exchange(struct.pack('<d', float(x)),
                             lambda parts: struct.pack('<d', sum(struct.unpack('<d', p)[0] for p in parts)))
        return struct.unpack('<d', out)[0]

    def allreduce_array(self, array, op=0):
        """In-place elementwise sum (``op == 0``) or maximum (``op == 1``) of a float64 array over the ranks,
        combined on rank 0 in rank order.  Only the host-staged transport (``attach(..., transport='host')``) sends
        vectors this way."""
        a = np.ascontiguousarray(array, dtype=np.float64)

        def combine(parts):
            acc = np.frombuffer(parts[0], dtype=np.float64).copy()
            for p in parts[1:]:
                other = np.frombuffer(p, dtype=np.float64)
                if other.shape != acc.shape:
                    raise RuntimeError('ranks entered a collective with %d and %d doubles' % (acc.size, other.size))
                acc = np.maximum(acc, other) if op == 1 else acc + other
            return acc.tobytes()
        out = np.frombuffer(self._exchange(a.tobytes(), combine), dtype=np.float64)
        array[...] = out.reshape(np.shape(array))
        return array

    def close(self):
        for c in self._peers:
            c.close()
        if self._sock is not None:
            self._sock.close()
        self._peers, self._sock = [], None


def broadcast_unique_id(rank, make_id, group=None):
    """Rank 0 creates the RCCL unique id; everybody receives it over ``group`` (a :class:`SocketGroup`) or,
    without one, over an initialised ``torch.distributed`` process group."""
    if group is not None:
        return group.broadcast_bytes(make_id() if rank == 0 else None)
    import torch.distributed as dist
    if not dist.is_initialized():
        raise RuntimeError('no control group: pass a SocketGroup or initialise torch.distributed first')
    box = [make_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    return box[0]


def attach(engine=None, group=None, transport=None):
    """Attach a communicator spanning ``group`` (default: the torch.distributed world) to ``engine``.

    ``transport`` (default: ``$VIABEL_AMD_TRANSPORT`` or ``'rccl'``): ``'rccl'`` is the data path of a real job;
    ``'host'`` stages every device collective through pinned host memory and ``group.allreduce_array`` -- for ranks
    RCCL cannot join (two ranks sharing one GPU, as the two-rank tests on a one-GPU box do); ``'ipc'`` is the
    xGMI-native all-reduce over windows the ranks map from each other (``vb_comm_init_ipc``: every element summed by one
    rank in rank order -- the same bits on every rank -- three launches, no ring; works between processes on one GPU
    too)."""
    engine = engine or _lib.default_engine()
    transport = transport or os.environ.get('VIABEL_AMD_TRANSPORT', 'rccl')
    if transport not in ('rccl', 'host', 'ipc'):
        raise ValueError("transport must be 'rccl', 'host' or 'ipc', got %r" % (transport,))
    if group is not None:
        world, rank = group.world, group.rank
    else:
        import torch.distributed as dist
        world, rank = dist.get_world_size(), dist.get_rank()
    if world == 1:
        return engine
    if transport == 'ipc':
        # xGMI-native all-reduce (vb_comm_init_ipc): every rank's window handle to every rank over the control group
        if group is None:
            raise ValueError("transport='ipc' needs a control group (SocketGroup) to exchange the window handles")
        mine = engine.comm_ipc_window(int(os.environ.get('VIABEL_AMD_IPC_DOUBLES', 1 << 21)))
        handles = group.allgather_bytes(mine)
        engine.comm_init_ipc(handles, world, rank)
    elif transport == 'host':
        if group is None:
            import torch
            import torch.distributed as dist

            def collective(array, op):
                t = torch.from_numpy(array)          # shares the staging buffer's memory
                dist.all_reduce(t, op=dist.ReduceOp.MAX if op == 1 else dist.ReduceOp.SUM)
        else:
            collective = group.allreduce_array
        engine.comm_init_host(collective, world, rank)
    else:
        uid = broadcast_unique_id(rank, _lib.Engine.comm_unique_id, group)
        engine.comm_init(uid, world, rank)
    if group is not None:
        # host-side random draws of the objectives (the alpha-divergence seed, the DIS resampling indices: the global
        # numpy RNG in the reference) are taken on rank 0 and handed to the other ranks over this group
        engine.control_group = group
    return engine


def init_from_env(backend=None):
    """One process per GPU as launched by ``torch.distributed.run``: socket control group from RANK / WORLD_SIZE /
    MASTER_*, RCCL communicator on the default engine (device ``LOCAL_RANK``).  ``backend`` (e.g. ``'gloo'``) uses
    a torch.distributed process group for the control path instead.  Returns the engine."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    engine = _lib.default_engine()
    if world == 1:
        return engine
    if backend is not None:
        import torch.distributed as dist
        if not dist.is_initialized():
            dist.init_process_group(backend=backend)
        return attach(engine)
    return attach(engine, SocketGroup.from_env())
